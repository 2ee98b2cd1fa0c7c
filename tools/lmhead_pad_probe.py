"""LM-head GEMM (h [R, 768] x W^T [768, Vp]) and its input gradient at the bench's decoder row counts for several paddings of the
vocabulary (50265 -> Vp): does a rounder Vp let the library pick a better kernel?  TunableOp tunes every shape it meets (in memory)."""
import sys
import torch
import torch.cuda.tunable as tunable

tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_max_tuning_duration(30)
tunable.set_max_tuning_iterations(20)
dev = "cuda"
d = 768


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for R in (2500, 4165, 8320, 332, 313, 521, 1040):
    h = torch.randn(R, d, device=dev).bfloat16()
    for Vp in (50272, 50304, 50432, 50688, 51200):
        W = torch.randn(Vp, d, device=dev).bfloat16()
        dl = torch.randn(R, Vp, device=dev).bfloat16()
        tf = timed(lambda: torch.nn.functional.linear(h, W))
        tb = timed(lambda: dl @ W)
        gf = 2.0 * R * d * Vp / 1e9
        print(f"R={R:5d} Vp={Vp}: fwd {tf:7.1f} us ({gf / tf / 1e3:5.2f} PF/s)   dgrad {tb:7.1f} us ({gf / tb / 1e3:5.2f} PF/s)   logits {R * Vp * 2 / 1e6:6.1f} MB", flush=True)
