#!/usr/bin/env python3
"""FFN activation + dropout pass (csrc/actdrop.hip) at the configs[1] shapes: forward and backward, HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vlpet_amd.act import act_dropout

def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for act in ("gelu", "gelu_new"):
    for M in (28000, 46648, 10000):
        x = torch.randn(M, 3072, device="cuda").bfloat16().requires_grad_(True)
        dy = torch.randn(M, 3072, device="cuda").bfloat16()
        f = lambda: act_dropout(x, act, 0.1, True, seed=5)
        t_f = timeit(f)
        o = f()
        t_b = timeit(lambda: torch.autograd.grad(o, x, dy, retain_graph=True))
        unit = M * 3072 * 2 / 1e6
        print(f"{act:8s} M={M:6d}: fwd {t_f:7.1f} us ({2 * unit / t_f * 1e3:6.0f} GB/s)   bwd {t_b:7.1f} us ({3 * unit / t_b * 1e3:6.0f} GB/s)")
