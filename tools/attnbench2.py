#!/usr/bin/env python3
"""Short-sequence attention kernels, forward and backward timed separately (HIP events), at the configs[1] shapes of the
encoder self-attention, the decoder's causal self-attention and the cross-attention."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vlpet_amd.attention import short_attention

def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

H = 12
shapes = [("enc vqa", 500, 56, 56, False), ("enc gqa", 833, 56, 56, False), ("enc nlvr", 166, 92, 92, False), ("enc cap", 416, 76, 76, False),
          ("dec-self vqa", 500, 20, 20, True), ("dec-self gqa", 833, 20, 20, True), ("cross vqa", 500, 20, 56, False),
          ("cross gqa", 833, 20, 56, False), ("cross nlvr", 166, 20, 92, False), ("cross cap", 416, 20, 76, False)]
for name, B, Lq, Lk, causal in shapes:
    q = torch.randn(B, Lq, H * 64, device="cuda").bfloat16().requires_grad_(True)
    k, v = (torch.randn(B, Lk, H * 64, device="cuda").bfloat16().requires_grad_(True) for _ in range(2))
    do = torch.randn(B, Lq, H * 64, device="cuda").bfloat16()
    f = lambda: short_attention(q, k, v, H, causal=causal, p=0.1, training=True, seed=1)
    t_f = timeit(f)
    o = f()
    t_b = timeit(lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True))
    uq, uk = B * Lq * H * 64 * 2 / 1e6, B * Lk * H * 64 * 2 / 1e6
    fb, bb = 2 * uq + 2 * uk, 4 * uq + 4 * uk          # MB: fwd q, o + k, v;  bwd q, o, do, dq + k, v, dk, dv
    print(f"{name:13s} B={B:4d} Lq={Lq:3d} Lk={Lk:3d}: fwd {t_f:7.1f} us ({fb / t_f * 1e3:6.0f} GB/s)   bwd {t_b:7.1f} us ({bb / t_b * 1e3:6.0f} GB/s)")
