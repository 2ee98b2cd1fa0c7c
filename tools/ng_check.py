#!/usr/bin/env python3
"""Parity of the backward without a gate (csrc/pet_cols_ng.hip through the product path) against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpu_cases as C
bad = 0
for kw in [dict(M=130, gate_mode=0), dict(M=32, gate_mode=0), dict(M=1000, gate_mode=0), dict(M=777, gate_mode=0, r=8, rg=8, nh=4), dict(M=28000, gate_mode=0)]:
    e = C.run_k1(torch.bfloat16, **kw)
    worst = max(e.values()); bad += worst > 1e-2
    print("k1 adapter-only", kw, " ".join(f"{k}={v:.1e}" for k, v in e.items()), "" if worst <= 1e-2 else "  <-- FAIL", flush=True)
for kw in [dict(M=777), dict(M=32), dict(M=5000)]:
    e = C.run_k2(torch.bfloat16, **kw)
    worst = max(e.values()); bad += worst > 1e-2
    print("k2", kw, " ".join(f"{k}={v:.1e}" for k, v in e.items()), "" if worst <= 1e-2 else "  <-- FAIL", flush=True)
print("FAILED" if bad else "ALL OK", bad)
