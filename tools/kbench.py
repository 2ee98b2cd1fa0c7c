#!/usr/bin/env python3
"""Per-kernel timing of the PET hot path with HIP events (torch.cuda.Event on the launch stream).
Prints algorithmic GB/s (SURVEY.md section 8d byte counts) for K1 fwd / bwd at config-2 sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vlpet_amd.functional as F
from vlpet_amd import _lib

def timeit(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us

def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 28000
    dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
    r = int(sys.argv[3]) if len(sys.argv) > 3 else 96
    d, dev = 768, "cuda"
    esz = 2 if dt == torch.bfloat16 else 4
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    x1 = torch.randn(M, d, device=dev, generator=g).to(dt); x2 = torch.randn(M, d, device=dev, generator=g).to(dt)
    dy = torch.randn(M, d, device=dev, generator=g).to(dt)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.05
    wd, bd, wu, bu = mk(r, d), mk(r), mk(d, r), mk(d)
    wgd, bgd, wgu, bgu = mk(r, d), mk(r), mk(d, r), mk(d)
    io = F._io_dtype(x2); tiles = F.rank_tiles(r)
    pa = F.pack_pair([wd], [bd], wu, bu, io, tiles); pg = F.pack_pair([wgd], [bgd], wgu, bgu, io, tiles)
    out = torch.empty_like(x2)
    st = torch.cuda.current_stream().cuda_stream
    t_pack = timeit(lambda: F.pack_pair([wd], [bd], wu, bu, io, tiles))
    def fwd():
        rc = lib.vlpet_adapter_gate_fwd(x1.data_ptr(), x2.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(), out.data_ptr(),
                                        M, d, tiles, 1, 1.0, 1.0, 1.0, io, st); assert rc == 0
    t_fwd = timeit(fwd)
    nws = lib.vlpet_bwd_workspace_bytes(M, d, tiles, 1, io)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    G = [torch.empty(r, d, **f32), torch.empty(r, **f32), torch.empty(d, r, **f32), torch.empty(d, **f32),
         torch.empty(r, d, **f32), torch.empty(r, **f32), torch.empty(d, r, **f32), torch.empty(d, **f32)]
    dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
    def bwd():
        rc = lib.vlpet_adapter_gate_bwd(dy.data_ptr(), x1.data_ptr(), x2.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(),
                                        dx1.data_ptr(), dx2.data_ptr(), *[t.data_ptr() for t in G], r, r, ws.data_ptr(), nws,
                                        M, d, tiles, 1, 1.0, 1.0, 1.0, io, st); assert rc == 0
    t_bwd = timeit(bwd)
    def bwd_ph(ph):
        def f():
            rc = lib.vlpet_adapter_gate_bwd_phase(ph, dy.data_ptr(), x1.data_ptr(), x2.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(),
                                                  dx1.data_ptr(), dx2.data_ptr(), *[t.data_ptr() for t in G], r, r, ws.data_ptr(), nws,
                                                  M, d, tiles, 1, 1.0, 1.0, 1.0, io, st); assert rc == 0
        return f
    t_rows, t_wg = timeit(bwd_ph(1)), timeit(bwd_ph(2))
    # training form: forward saves z / gelu'(pre), backward rows kernel skips the recompute phase
    nsv = lib.vlpet_saved_bytes(M, tiles, io)
    sv = torch.empty(nsv, dtype=torch.uint8, device=dev)
    def fwd_save():
        rc = lib.vlpet_adapter_gate_fwd_save(x1.data_ptr(), x2.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(), out.data_ptr(),
                                             sv.data_ptr(), M, d, tiles, 1, 1.0, 1.0, 1.0, io, st); assert rc == 0
    t_fwd_s = timeit(fwd_save)
    def bwd_saved(ph):
        def f():
            rc = lib.vlpet_adapter_gate_bwd_saved(ph, dy.data_ptr(), x1.data_ptr(), x2.data_ptr(), sv.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(),
                                                  dx1.data_ptr(), dx2.data_ptr(), *[t.data_ptr() for t in G], r, r, ws.data_ptr(), nws,
                                                  M, d, tiles, 1, 1.0, 1.0, 1.0, io, st); assert rc == 0
        return f
    fwd_save()
    t_rows_s = timeit(bwd_saved(1))
    t_ph2_s, t_op_s = timeit(bwd_saved(2)), timeit(bwd_saved(3))
    t_old1, t_old2 = timeit(bwd_saved(1 | 4)), timeit(bwd_saved(2 | 4))       # the previous form: rows kernel with side products + wgrad
    def k2f():
        rc = lib.vlpet_parallel_adapter_fwd(x1.data_ptr(), x2.data_ptr(), pa.buf.data_ptr(), out.data_ptr(), M, d, tiles, 1.0, io, st); assert rc == 0
    t_k2 = timeit(k2f)
    fb, bb = 3 * d * M * esz, 5 * d * M * esz
    print(f"M={M} dtype={dt} r={r} ws={nws/1e6:.1f}MB")
    print(f"pack pair      : {t_pack:8.1f} us")
    print(f"K1 fwd         : {t_fwd:8.1f} us   {fb/t_fwd/1e3:8.1f} GB/s algorithmic ({fb/1e6:.1f} MB)  frac(8TB/s)={fb/t_fwd/1e3/8000:.3f}")
    print(f"K1 bwd (3 krn) : {t_bwd:8.1f} us   {bb/t_bwd/1e3:8.1f} GB/s algorithmic ({bb/1e6:.1f} MB)  frac(8TB/s)={bb/t_bwd/1e3/8000:.3f}")
    print(f"   bwd rows krn: {t_rows:8.1f} us   {bb/t_rows/1e3:8.1f} GB/s algorithmic  frac(8TB/s)={bb/t_rows/1e3/8000:.3f}")
    print(f"   bwd wgrad+fin: {t_wg:7.1f} us")
    print(f"K1 fwd + save  : {t_fwd_s:8.1f} us   bwd rows with saved activations: {t_rows_s:8.1f} us   {bb/t_rows_s/1e3:8.1f} GB/s algorithmic  frac(8TB/s)={bb/t_rows_s/1e3/8000:.3f}")
    print(f"K1 bwd, training form (saved activations), two-pass: pass 1 (dz) {t_rows_s:7.1f} us + pass 2 (cols + finalize) {t_ph2_s:7.1f} us; "
          f"whole op {t_op_s:7.1f} us = {bb/t_op_s/1e3:8.1f} GB/s algorithmic, frac(8TB/s)={bb/t_op_s/1e3/8000:.3f}")
    print(f"   previous form (rows kernel with dh / dq side products + wgrad): {t_old1:7.1f} + {t_old2:7.1f} = {t_old1 + t_old2:7.1f} us, "
          f"frac(8TB/s)={bb/(t_old1 + t_old2)/1e3/8000:.3f}")
    print(f"K2 fwd         : {t_k2:8.1f} us   {fb/t_k2/1e3:8.1f} GB/s algorithmic")
    # eager reference chain on the GPU for comparison (what the reference runs today)
    sys.path.insert(0, ROOT)
    from oracle import vlpet_oracle as O   # checker only: timed as the eager-GPU comparison, not shipped
    W = [t.to(dt) for t in (wd, bd, wu, bu, wgd, bgd, wgu, bgu)]
    def eager():
        return O.encoder_adapter_gate(x1[None], x2[None], [W[0]], [W[1]], W[2], W[3],
                                      dict(down_w=W[4], down_b=W[5], up_w=W[6], up_b=W[7]))
    t_e = timeit(eager, iters=20, warm=5)
    print(f"eager torch fwd: {t_e:8.1f} us   ({t_e/t_fwd:.1f}x the fused kernel)")

if __name__ == "__main__":
    main()
