#!/usr/bin/env python3
"""Times only the weight-gradient part of the K1 backward (phase 2 of the previous form: wgrad kernel + finalize) at one M.
usage: wgbench.py M [tag]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vlpet_amd.functional as F
from vlpet_amd import _lib
from kbench import timeit

def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 28000
    tag = sys.argv[2] if len(sys.argv) > 2 else ""
    dt, r, d, dev = torch.bfloat16, 96, 768, "cuda"
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
    rc = lib.vlpet_adapter_gate_fwd_save(x1.data_ptr(), x2.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(), out.data_ptr(),
                                         sv.data_ptr(), M, d, tiles, 1, 1.0, 1.0, 1.0, io, st); assert rc == 0
    def bwd_saved(ph):
        def f():
            rc = lib.vlpet_adapter_gate_bwd_saved(ph, dy.data_ptr(), x1.data_ptr(), x2.data_ptr(), sv.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(),
                                                  dx1.data_ptr(), dx2.data_ptr(), *[t.data_ptr() for t in G], r, r, ws.data_ptr(), nws,
                                                  M, d, tiles, 1, 1.0, 1.0, 1.0, io, st); assert rc == 0
        return f
    bwd_saved(1 | 4)()
    if tag != "modes":
        t = timeit(bwd_saved(2 | 4), iters=100, warm=10)
        print(f"wgbench M={M} {tag}: wgrad+finalize {t:7.1f} us")
        return
    # experiment build (VLPET_LIB=..._exp.so): the mode switches are read at every launch, so one process / one set of buffers
    modes = [("base", 0, 0), ("stream: no stores", 0, 1), ("stream: no products", 0, 2), ("stream: no products, no stores", 0, 3),
             ("stream: X pieces only", 0, 3 + 4), ("stream: P pieces only", 0, 3 + 8)]
    # reference: what plain streaming kernels reach on this box at the same footprint (4 x [M, 768] bf16 read)
    big = torch.randn(4 * M, d, device=dev).to(dt); o2 = torch.empty(M, d, device=dev, dtype=dt)
    t = timeit(lambda: big.sum(dtype=torch.float32), iters=30, warm=3)
    print(f"reference: torch sum over {big.numel() * 2 / 1e6:.0f} MB: {t:7.1f} us = {big.numel() * 2 / t / 1e3:7.0f} GB/s read")
    t = timeit(lambda: torch.add(x1, x2, out=o2), iters=30, warm=3)
    print(f"reference: torch add, 3 x {x1.numel() * 2 / 1e6:.0f} MB: {t:7.1f} us = {3 * x1.numel() * 2 / t / 1e3:7.0f} GB/s read + write")
    del big
    for rep in range(2):
        for name, fm, wm in modes:
            os.environ["VLPET_FIN_MODE"] = str(fm); os.environ["VLPET_WGS_MODE"] = str(wm)
            t = timeit(bwd_saved(2 | 4), iters=60, warm=5)
            print(f"rep {rep} {name:48s} {t:7.1f} us")

if __name__ == "__main__":
    main()
