#!/usr/bin/env python3
"""Do HIP events recorded INSIDE a captured graph give elapsed times after a replay (torch.cuda.Event.record() under capture = an event-record node)?"""
import torch
x = torch.randn(4096, 4096, device="cuda")
y = x @ x; torch.cuda.synchronize()
s = torch.cuda.Stream()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        evs[0].record()
        y = x @ x
        evs[1].record()
        z = y @ x
        evs[2].record()
    for it in range(3):
        g.replay()
        torch.cuda.synchronize()
        print("replay", it, "elapsed 0->1", evs[0].elapsed_time(evs[1]), "ms; 1->2", evs[1].elapsed_time(evs[2]), "ms", flush=True)
except Exception as e:
    print("FAILED:", type(e).__name__, e)
# external timing for comparison
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); y = x @ x; e1.record(); torch.cuda.synchronize(); print("eager matmul", e0.elapsed_time(e1), "ms")
