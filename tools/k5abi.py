#!/usr/bin/env python3
"""K5 forward / backward kernels called through the C ABI (no autograd glue), warm (repeated call: Infinity-Cache hits) and cold
(1 GiB read-modify-write before every timed call: the condition inside a training step).  usage: k5abi.py [M ...]
Both forms: fwd3 / bwd_out = the forward writes only its output and the backward recovers xhat from it (3 units forward, 3 (+1 for dy
under dropout) backward); fwd4 / bwd_h = the forward also writes the pre-norm sum.  Fractions = 3 * d * M * 2 B / time / 8 TB/s
(SURVEY 8d's algorithmic count); GB/s = the bytes the call really moves."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vlpet_amd import _lib
import vlpet_amd.functional as F
from kbench import timeit

def cold(fn, evict, iters=15):
    ts = []
    for _ in range(iters):
        evict.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

def run(M, evict, d=768, p=0.1):
    dev, dt = "cuda", torch.bfloat16
    lib = _lib.load()
    y, x1, dout = (torch.randn(M, d, device=dev).to(dt) for _ in range(3))
    out, h, dx1, dy = (torch.empty(M, d, device=dev, dtype=dt) for _ in range(4))
    gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    part = torch.empty(lib.vlpet_sublayer_tail_partials(M), 2, d, device=dev)
    io, st = F._io_dtype(y), torch.cuda.current_stream().cuda_stream
    def fwd(hp):
        def f():
            rc = lib.vlpet_sublayer_tail_fwd(y.data_ptr(), x1.data_ptr(), gam.data_ptr(), bet.data_ptr(), out.data_ptr(), hp,
                                             mean.data_ptr(), rstd.data_ptr(), None, M, d, 1e-5, p, 7, 1, io, st); assert rc == 0, rc
        return f
    def bwd_h():
        rc = lib.vlpet_sublayer_tail_bwd(dout.data_ptr(), h.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gam.data_ptr(), dx1.data_ptr(),
                                         dy.data_ptr(), part.data_ptr(), M, d, p, 7, 1, io, st); assert rc == 0, rc
    def bwd_out():
        rc = lib.vlpet_sublayer_tail_bwd_out(dout.data_ptr(), out.data_ptr(), rstd.data_ptr(), gam.data_ptr(), bet.data_ptr(), dx1.data_ptr(),
                                             dy.data_ptr(), part.data_ptr(), M, d, p, 7, io, st); assert rc == 0, rc
    fwd(h.data_ptr())(); bwd_h(); bwd_out()
    unit = M * d * 2
    cols = [("fwd3", fwd(None), 3), ("fwd4", fwd(h.data_ptr()), 4), ("bwd_out", bwd_out, 4 if p > 0 else 3), ("bwd_h", bwd_h, 4 if p > 0 else 3)]
    line = f"k5abi M={M:6d} p={p}:"
    for name, fn, units in cols:
        tw, tc = timeit(fn), cold(fn, evict)
        line += f"  {name} warm {tw:5.1f} us ({3 * unit / tw / 1e3 / 8000:.3f}; {units * unit / tw / 1e3:5.0f} GB/s) cold {tc:5.1f} us ({3 * unit / tc / 1e3 / 8000:.3f})"
    print(line + f"   blocks {lib.vlpet_sublayer_tail_partials(M)}", flush=True)

if __name__ == "__main__":
    evict = torch.zeros(1 << 28, dtype=torch.float32, device="cuda")
    p = float(os.environ.get("K5ABI_P", "0.1"))
    for M in [int(a) for a in sys.argv[1:]] or [10000, 28000, 46648]:
        run(M, evict, p=p)
