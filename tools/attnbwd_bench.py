#!/usr/bin/env python3
"""The attention kernels through the C ABI alone (no autograd, no host work between launches): HIP events on the launch stream
around N back-to-back calls, at the configs[1] / configs[2] shapes.  tools/attnbench.py's "bwd" column is (fwd + bwd) - fwd of
autograd.grad and bottoms out near 100 us of host time; this is the kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vlpet_amd import _lib
from vlpet_amd.attention import short_attention

H = 12
P = float(os.environ.get("ATTNBENCH_P", "0.1"))
lib = _lib.load()
tag = sys.argv[1] if len(sys.argv) > 1 else ""
SHAPES = [("vqa", 500, 56, 56), ("gqa", 833, 56, 56), ("nlvr", 166, 92, 92), ("caption", 416, 76, 76), ("dec-self", 500, 20, 20),
          ("dec-cross", 500, 20, 56)]
for name, B, Lq, Lk in SHAPES:
    g = torch.Generator().manual_seed(1)
    mk = lambda L: (torch.randn(B, L, H * 64, generator=g) * 0.5).cuda().bfloat16()
    q, k, v, do = mk(Lq), mk(Lk), mk(Lk), mk(Lq)
    o = short_attention(q, k, v, H, p=P, training=True, seed=1)
    # the log-sum-exp of the forward: run the forward entry once more into our own buffers
    o2 = torch.empty_like(o); lse = torch.empty(B, H, Lq, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.vlpet_attn_fwd_bias(q.data_ptr(), k.data_ptr(), v.data_ptr(), None, None, o2.data_ptr(), lse.data_ptr(), None, B, H, Lq, Lk,
                                 H * 64, H * 64, 0, 0.125, P, 1, st)
    assert rc == 0
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    def run_f():
        return lib.vlpet_attn_fwd_bias(q.data_ptr(), k.data_ptr(), v.data_ptr(), None, None, o2.data_ptr(), lse.data_ptr(), None, B, H, Lq, Lk,
                                       H * 64, H * 64, 0, 0.125, P, 1, st)
    def run():
        return lib.vlpet_attn_bwd_bias(q.data_ptr(), k.data_ptr(), v.data_ptr(), o2.data_ptr(), do.data_ptr(), lse.data_ptr(), None, None, None,
                                       dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, Lq, Lk, H * 64, H * 64, 0, 0.125, P, 1, st)
    def timed(fn, N=50):
        for _ in range(5): assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / N * 1e3
    t, tf = timed(run), timed(run_f)
    unit = B * H * 64 * 2 / 1e6                                   # MB per row of tokens
    mb = unit * (3 * Lq + 2 * Lk) + unit * (Lq + 2 * Lk)          # reads q, o, do, k, v; writes dq, dk, dv
    mbf = unit * (2 * Lq + 2 * Lk)                                # forward: reads q, k, v; writes o
    print(f"attnbwd {tag} {name:9s} B={B} Lq={Lq} Lk={Lk}: bwd {t:7.1f} us ({mb / t:5.2f} TB/s of {mb:4.0f} MB)   fwd {tf:6.1f} us ({mbf / tf:5.2f} TB/s of {mbf:4.0f} MB)")
