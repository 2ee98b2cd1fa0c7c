#!/usr/bin/env python3
"""Short-sequence attention kernels vs torch SDPA at the configs[1] encoder shapes (HIP events on the launch stream)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
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
P = float(os.environ.get("ATTNBENCH_P", "0.1"))
for name, B, S in [("vqa", 500, 56), ("gqa", 833, 56), ("nlvr", 166, 92), ("caption", 416, 76), ("dec-self", 500, 5)]:
    q, k, v, do = (torch.randn(B, S, H * 64, device="cuda").bfloat16().requires_grad_(i < 3) for i in range(4))
    o = short_attention(q, k, v, H, p=P, training=True, seed=1)
    t_f = timeit(lambda: short_attention(q, k, v, H, p=P, training=True, seed=1))
    t_fb = timeit(lambda: torch.autograd.grad(short_attention(q, k, v, H, p=P, training=True, seed=1), (q, k, v), do))
    sh = lambda t: t.view(B, S, H, 64).transpose(1, 2)
    sd = lambda: F.scaled_dot_product_attention(sh(q), sh(k), sh(v), dropout_p=P).transpose(1, 2).reshape(B, S, H * 64)
    t_sf = timeit(sd)
    t_sfb = timeit(lambda: torch.autograd.grad(sd(), (q, k, v), do))
    unit = B * S * H * 64 * 2 / 1e6
    print(f"{name:8s} B={B} S={S}: fwd {t_f:7.1f} us ({4*unit/t_f/1e3*1e3:6.0f} GB/s)  bwd {t_fb - t_f:7.1f} us ({8*unit/(t_fb-t_f)*1e3/1e3:6.0f} GB/s)"
          f"   | SDPA fwd {t_sf:7.1f}  bwd {t_sfb - t_sf:7.1f}")
