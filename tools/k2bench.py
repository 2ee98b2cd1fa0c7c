#!/usr/bin/env python3
"""K2 (decoder value-parallel adapter, no gate) through the C ABI: forward (training form) and backward with the saved
activations, HIP events.  usage: k2bench.py [tag] [M...]   (K2BENCH_R = rank, default 96)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vlpet_amd.functional as F
from vlpet_amd import _lib

tag = sys.argv[1] if len(sys.argv) > 1 else "k2"
Ms = [int(a) for a in sys.argv[2:]] or [28000]
lib = _lib.load()
dev, dt, d = "cuda", torch.bfloat16, 768
r = int(os.environ.get("K2BENCH_R", "96"))
st = torch.cuda.current_stream().cuda_stream


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for M in Ms:
    g = torch.Generator(device=dev).manual_seed(0)
    x, y, dy = (torch.randn(M, d, device=dev, generator=g).to(dt) for _ in range(3))
    mk = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.05
    wd, bd, wu, bu = mk(r, d), mk(r), mk(d, r), mk(d)
    io, tiles = 1, F.rank_tiles(r)
    pk = F.pack_pair([wd], [bd], wu, bu, io, tiles)
    out = torch.empty_like(x); dx = torch.empty_like(x)
    sv = torch.empty(lib.vlpet_saved_bytes(M, tiles, io), dtype=torch.uint8, device=dev)
    nws = lib.vlpet_bwd_workspace_bytes(M, d, tiles, 0, io); ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    G = [torch.empty(r, d, device=dev), torch.empty(r, device=dev), torch.empty(d, r, device=dev), torch.empty(d, device=dev)]

    def fwd():
        rc = lib.vlpet_parallel_adapter_fwd_save(x.data_ptr(), y.data_ptr(), pk.buf.data_ptr(), out.data_ptr(), sv.data_ptr(), M, d, tiles, 1.0, io, st)
        assert rc == 0

    def bwd():
        rc = lib.vlpet_parallel_adapter_bwd_saved(dy.data_ptr(), x.data_ptr(), sv.data_ptr(), pk.buf.data_ptr(), dx.data_ptr(),
                                                  *[t.data_ptr() for t in G], r, ws.data_ptr(), nws, M, d, tiles, 1.0, io, st)
        assert rc == 0

    fwd()
    tf, tb = timed(fwd), timed(bwd)
    alg = 3.0 * d * 2 * M
    print(f"k2bench {tag:8s} M={M:6d} r={r}: fwd+save {tf:6.1f} us (frac {alg / tf / 1e6 / 8:.3f})   bwd {tb:6.1f} us (frac {alg / tb / 1e6 / 8:.3f})", flush=True)
