#!/usr/bin/env python3
"""T5 attention shapes (configs[2]: B = 300 / 500 / 100 / 250, scale 1, relative bias shared by the batch) on the short-sequence
kernels with an AttnBias, forward and backward timed separately; the library's SDPA with the merged dense mask beside it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from vlpet_amd.attention import AttnBias, short_attention
from attnbench2 import timeit

H = 12
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for name, B, L in (("enc vqa", 300, 56), ("enc gqa", 500, 56), ("enc nlvr", 100, 92), ("enc cap", 250, 76)):
    q, k, v = (torch.randn(B, L, H * 64, device="cuda").bfloat16().mul_(0.3).requires_grad_(True) for _ in range(3))
    do = torch.randn(B, L, H * 64, device="cuda").bfloat16()
    rel = torch.randn(1, H, L, L, device="cuda")
    km = torch.ones(B, L, dtype=torch.bool, device="cuda")
    ab = AttnBias(rel)
    f = lambda: short_attention(q, k, v, H, km, False, 0.1, True, scale=1.0, seed=1, bias=ab)
    t_f = timeit(f); o = f()
    t_b = timeit(lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True))
    dense = (rel + torch.zeros(B, 1, 1, L, device="cuda")).bfloat16()
    sh = lambda t: t.view(B, L, H, 64).transpose(1, 2)
    g = lambda: F.scaled_dot_product_attention(sh(q), sh(k), sh(v), attn_mask=dense, dropout_p=0.1, scale=1.0)
    t_lf = timeit(g); o2 = g()
    t_lb = timeit(lambda: torch.autograd.grad(o2, (q, k, v), sh(do), retain_graph=True))
    print(f"attnbench_t5 {tag} {name:9s} B={B:4d} L={L:3d}: kernels fwd {t_f:6.1f} us bwd {t_b:6.1f} us | library SDPA with a dense mask fwd {t_lf:6.1f} us bwd {t_lb:6.1f} us", flush=True)
