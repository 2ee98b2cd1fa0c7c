#!/usr/bin/env python3
"""K1 forward (training form) and backward rows kernel only, at the sizes given: a quick A/B target for alternative builds
(VLPET_LIB=...).  usage: k1bench.py tag M [M ...]    K1BENCH_R=192: six tiles; K1BENCH_FROM_X2=1: the backward recomputes h from x2
(rounds 2-4) instead of starting from the forward's output (vlpet_adapter_gate_bwd_saved_y, round 5)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vlpet_amd.functional as F
from vlpet_amd import _lib
from kbench import timeit

def run(M, tag):
    dt, r, d, dev = torch.bfloat16, int(os.environ.get("K1BENCH_R", "96")), 768, "cuda"
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
    nws = lib.vlpet_bwd_workspace_bytes(M, d, tiles, 1, io)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    G = [torch.empty(r, d, **f32), torch.empty(r, **f32), torch.empty(d, r, **f32), torch.empty(d, **f32),
         torch.empty(r, d, **f32), torch.empty(r, **f32), torch.empty(d, r, **f32), torch.empty(d, **f32)]
    dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
    nsv = lib.vlpet_saved_bytes(M, tiles, io)
    sv = torch.empty(nsv, dtype=torch.uint8, device=dev)
    def fwd_save():
        rc = lib.vlpet_adapter_gate_fwd_save(x1.data_ptr(), x2.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(), out.data_ptr(),
                                             sv.data_ptr(), M, d, tiles, 1, 1.0, 1.0, 1.0, io, st); assert rc == 0
    yp = None if os.environ.get("K1BENCH_FROM_X2") else out.data_ptr()
    def bwd_saved(ph):
        def f():
            rc = lib.vlpet_adapter_gate_bwd_saved_y(ph, dy.data_ptr(), x1.data_ptr(), x2.data_ptr(), yp, sv.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(),
                                                    None, dx1.data_ptr(), dx2.data_ptr(), *[t.data_ptr() for t in G], r, r, ws.data_ptr(), nws,
                                                    M, d, tiles, 1, 1.0, 1.0, 1.0, io, st); assert rc == 0
        return f
    fwd_save()
    if os.environ.get("K1BENCH_FWD_ONLY"):      # forward only (A/B of forward forms: VLPET_LIB=<debug build> VLPET_FWD2P=0|1|2|3)
        f = min(timeit(fwd_save, iters=60, warm=5) for _ in range(2))
        nb = 3 * d * M * 2
        print(f"k1bench {tag:10s} M={M:6d} r={r:3d}: fwd+save {f:6.1f} us (frac of 8 TB/s on 3*d*b per row: {nb / f / 1e3 / 8000:.3f})", flush=True)
        return
    if os.environ.get("K1BENCH_COLD"):
        # cold mode: the repeated call's inputs are otherwise hits of the 256 MB Infinity Cache left by the previous iteration, which a
        # training step never sees (third session).  Before every timed call a read-modify-write over 1 GiB.
        evict = torch.zeros(1 << 28, dtype=torch.float32, device=dev)
        def cold(fn, iters=15):
            ts = []
            for _ in range(iters):
                evict.add_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            return ts[len(ts) // 2]
        f, b, w, p1, p2, op = cold(fwd_save), cold(bwd_saved(1 | 4)), cold(bwd_saved(2 | 4)), cold(bwd_saved(1)), cold(bwd_saved(2)), cold(bwd_saved(3))
        print(f"k1bench {tag:10s} M={M:6d} COLD: fwd+save {f:6.1f} us | previous split: rows {b:6.1f} + wgrad+fin (dh, dq cold too) {w:6.1f} us | "
              f"default: pass 1 {p1:6.1f}, pass 2 + fin {p2:6.1f}, whole op in one call {op:6.1f} us (op frac {5 * d * M * 2 / op / 1e3 / 8000:.3f})", flush=True)
        return
    res = []
    for rep in range(2):
        res.append((timeit(fwd_save, iters=60, warm=5), timeit(bwd_saved(1 | 4), iters=60, warm=5), timeit(bwd_saved(2 | 4), iters=60, warm=5),
                    timeit(bwd_saved(1), iters=60, warm=5), timeit(bwd_saved(2), iters=60, warm=5), timeit(bwd_saved(3), iters=60, warm=5)))
    f, b, w, p1, p2, op = (min(x[i] for x in res) for i in range(6))
    frac = lambda t: 5 * d * M * 2 / t / 1e3 / 8000
    print(f"k1bench {tag:10s} M={M:6d}: fwd+save {f:6.1f} us | previous split: rows {b:6.1f} + wgrad+fin {w:6.1f} = {b + w:6.1f} us (op frac {frac(b + w):.3f}) | "
          f"default: pass 1 {p1:6.1f} + pass 2 + fin {p2:6.1f} us, whole op {op:6.1f} us (op frac {frac(op):.3f})", flush=True)

if __name__ == "__main__":
    tag = sys.argv[1]
    for M in [int(a) for a in sys.argv[2:]] or [28000]:
        run(M, tag)
