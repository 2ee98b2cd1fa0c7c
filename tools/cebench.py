#!/usr/bin/env python3
"""LM-head cross entropy kernels through the C ABI at the configs[1] sizes (rows = batch x target length, V = 50,465, ld = 50,472)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vlpet_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
V, ld = 50465, 50472
for N in (2500, 4165, 332, 8320):
    lg = torch.randn(N, ld, device="cuda").bfloat16()
    lab = torch.randint(0, V, (N,), device="cuda")
    loss, lse, dl = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.full((N,), 1.0 / N, device="cuda")
    dg = torch.empty_like(lg)
    f = lambda: lib.vlpet_ce_loss_fwd(lg.data_ptr(), lab.data_ptr(), loss.data_ptr(), lse.data_ptr(), N, V, ld, _lib.VLPET_BF16, st)
    b = lambda: lib.vlpet_ce_loss_bwd(lg.data_ptr(), lab.data_ptr(), lse.data_ptr(), dl.data_ptr(), dg.data_ptr(), N, V, ld, _lib.VLPET_BF16, st)
    def timed(fn, it=20):
        for _ in range(3): assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e3
    tf, tb = timed(f), timed(b)
    mb = N * ld * 2 / 1e6
    print(f"cebench N={N}: fwd {tf:6.1f} us ({mb / tf:5.2f} TB/s of {mb:4.0f} MB)   bwd {tb:6.1f} us ({2 * mb / tb:5.2f} TB/s of {2 * mb:4.0f} MB)")
