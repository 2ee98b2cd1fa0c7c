"""The LM-head GEMMs (logits = h W^T, dh = dlogits W) measured COLD (a 1 GB fill between calls, one call per measurement) -- the train
step runs them once per step behind ~400 other launches -- with the committed TunableOp table, with the library's default choice, and
split into row chunks."""
import os
import sys

import torch
import torch.cuda.tunable as tunable

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import vlpet_amd.train as TR  # noqa: E402

dev = "cuda"
d, Vp = 768, 50472
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)


def cold(fn, n=12):
    ts = []
    for _ in range(n):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def warm(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


W = torch.randn(Vp, d, device=dev).bfloat16()
for mode in ("table", "default"):
    if mode == "table":
        assert TR.use_tuned_gemms()
    else:
        tunable.enable(False)
    for R in (2500, 4165, 8320):
        h = torch.randn(R, d, device=dev).bfloat16()
        dl = torch.randn(R, Vp, device=dev).bfloat16()
        out = {}
        out["fwd"] = (cold(lambda: torch.nn.functional.linear(h, W)), warm(lambda: torch.nn.functional.linear(h, W)))
        out["dgrad"] = (cold(lambda: dl @ W), warm(lambda: dl @ W))
        for nch in (2, 4):
            hs = h.chunk(nch, 0)
            out[f"fwd/{nch} chunks"] = (cold(lambda: [torch.nn.functional.linear(x, W) for x in hs]), warm(lambda: [torch.nn.functional.linear(x, W) for x in hs]))
        print(f"[{mode}] R={R}: " + "   ".join(f"{k} cold {v[0]:6.1f} warm {v[1]:6.1f} us" for k, v in out.items()), flush=True)
