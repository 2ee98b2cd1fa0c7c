#!/usr/bin/env python3
"""Run the parity case matrix on the GPU and print every error (no asserts) -- first stop when a
kernel misbehaves.  Usage on the GPU box:  python tools/gpu_diag.py [quick]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpu_cases as C

def show(tag, fn, *a, **k):
    t0 = time.time()
    try:
        r = fn(*a, **k)
        if isinstance(r, dict):
            worst = max(r.values())
            body = " ".join(f"{n}={v:.2e}" for n, v in r.items())
            print(f"[{tag}] worst={worst:.2e} :: {body}  ({time.time()-t0:.1f}s)", flush=True)
        else:
            print(f"[{tag}] {r}  ({time.time()-t0:.1f}s)", flush=True)
    except Exception:
        print(f"[{tag}] EXCEPTION\n{traceback.format_exc()}", flush=True)

print(torch.__version__, torch.cuda.get_device_name(0), flush=True)
bf, f32 = torch.bfloat16, torch.float32
show("pack bf16 r96", C.run_pack_check, 96, 768, 4, False)
show("pack f32 r96", C.run_pack_check, 96, 768, 4, True)
show("pack bf16 r8 d64", C.run_pack_check, 8, 64, 4, False)
show("k1 f32 tiny", C.run_k1, f32, M=40, d=64, r=8, rg=8, nh=4)
show("k1 bf16 tiny", C.run_k1, bf, M=40, d=64, r=8, rg=8, nh=4)
show("k1 f32 cfg1", C.run_k1, f32, M=224)
show("k1 bf16 cfg1", C.run_k1, bf, M=224)
show("k1 f32 nogate", C.run_k1, f32, M=130, gate_mode=0)
show("k1 f32 add", C.run_k1, f32, M=130, gate_mode=2, gate_scale=0.3)
show("k1 f32 t5 scales", C.run_k1, f32, M=100, d=768, r=192, rg=192, nh=4, delta_scale=4.0, x2_scale=0.5, gate_scale=0.3)
show("k1 bf16 r192", C.run_k1, bf, M=300, d=768, r=192, rg=192, nh=4, gate_scale=0.3)
show("k2 f32", C.run_k2, f32)
show("k2 bf16 scaled", C.run_k2, bf, M=333, scale=4.0)
show("k3 f32 r8", C.run_k3, f32, r=8)
show("k3 bf16 r64", C.run_k3, bf, r=64)
show("k3 f32 r128 drop", C.run_k3, f32, r=128, p=0.1)
if len(sys.argv) < 2:
    show("k1 bf16 M=28000", C.run_k1, bf, M=28000)
    show("k1 f32 M=5000", C.run_k1, f32, M=5000)
