#!/usr/bin/env python3
"""Parity of the gated K1 backward (bf16 -> the column-parallel pass of csrc/pet_cols.hip) against the oracle at a list of
shapes, then its timing (tools/k1bench.py).  usage: colz_check.py [quick]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpu_cases as C

cases = [dict(M=224), dict(M=32), dict(M=1000), dict(M=3500), dict(M=1000, gate_mode=2), dict(M=777, r=8, rg=8, nh=4),
         dict(M=1000, r=96, rg=32, nh=4), dict(M=2100, gate_scale=0.3, delta_scale=0.5, x2_scale=0.7), dict(M=28000), dict(M=33200)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    cases = cases[:4]
if len(sys.argv) > 1 and sys.argv[1] == "r192":      # the T5 script's rank: six tiles (csrc/pet_cols6.hip)
    T5 = dict(r=192, rg=192, nh=4, delta_scale=4.0, x2_scale=0.5, gate_scale=0.3)
    cases = [dict(M=32, **T5), dict(M=100, **T5), dict(M=1000, **T5), dict(M=2100, **T5), dict(M=999, gate_mode=2, **T5),
             dict(M=1000, r=192, rg=128, nh=4), dict(M=1000, r=128, rg=192, nh=4), dict(M=16800, **T5)]
bad = 0
for kw in cases:
    ce = {}
    e = C.run_k1(torch.bfloat16, col_errs=ce, **kw)
    worst = max(e.values())
    flag = "" if worst <= 1e-2 else "   <-- FAIL"
    bad += worst > 1e-2
    print(kw, " ".join(f"{k}={v:.1e}" for k, v in e.items()), "| cols", " ".join(f"{k}={v:.1e}" for k, v in ce.items()), flag, flush=True)
print("FAILED" if bad else "ALL OK", bad)
