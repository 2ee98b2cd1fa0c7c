#!/usr/bin/env python3
"""Downsample (AdaptiveMaxPool2d 7x7 -> 6x6 over the CLIP grid, fp32 features -> bf16) through the C ABI at the configs[1] batch sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vlpet_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for B in (500, 833, 332, 416):
    x = torch.randn(B, 49, 2048, device="cuda")
    out = torch.empty(B, 36, 2048, device="cuda", dtype=torch.bfloat16)
    f = lambda: lib.vlpet_downsample_fwd(x.data_ptr(), out.data_ptr(), B, 7, 6, 2048, _lib.VLPET_F32, _lib.VLPET_BF16, st)
    for _ in range(5): assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 30 * 1e3
    mb = (x.numel() * 4 + out.numel() * 2) / 1e6
    print(f"poolbench B={B}: {t:6.1f} us  ({mb / t:5.2f} TB/s of {mb:4.0f} MB)")
