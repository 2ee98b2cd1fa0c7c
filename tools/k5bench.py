#!/usr/bin/env python3
"""K5 (sublayer tail: LN(x1 + dropout(y)) / x1 + dropout(y)) forward and backward timings at one size (HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vlpet_amd import tail as T

def timeit(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 28000
    d, dev, dt = 768, "cuda", torch.bfloat16
    x1 = torch.randn(M, d, device=dev).to(dt).requires_grad_(True)
    y = torch.randn(M, d, device=dev).to(dt).requires_grad_(True)
    norm = torch.nn.LayerNorm(d).to(dev)
    T.SAVE_PRENORM = os.environ.get("K5BENCH_PRENORM", "0") == "1"       # A/B: the forward also writes the pre-norm sum (round-3 form)
    print(f"SAVE_PRENORM = {T.SAVE_PRENORM}")
    for p in (0.0, 0.1):
        out = T.sublayer_tail(x1, y, norm, p, True, seed=1)
        g = torch.randn_like(out)
        tf = timeit(lambda: T.sublayer_tail(x1, y, norm, p, True, seed=1))
        def fb():
            o = T.sublayer_tail(x1, y, norm, p, True, seed=1)
            o.backward(g)
        tfb = timeit(fb)
        b = M * d * 2
        print(f"M={M} p={p}: fwd {tf:6.1f} us ({3*b/tf/1e3:7.1f} GB/s for 3 units)   fwd+bwd {tfb:6.1f} us  -> bwd ~{tfb-tf:6.1f} us ({3*b/(tfb-tf)/1e3:7.1f} GB/s for 3 units)")

if __name__ == "__main__":
    main()
