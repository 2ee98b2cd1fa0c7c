#!/usr/bin/env python3
"""K4 (visual projection: Linear(2048 -> 768) + LayerNorm) at the bench's visual-row count: forward and weight gradient through
the C ABI, HIP events, plus the same GEMM through the library for reference.  usage: k4bench.py [tag] [M...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vlpet_amd.functional as F
from vlpet_amd import _lib
from vlpet_amd.visproj import VisProjPackCache

tag = sys.argv[1] if len(sys.argv) > 1 else "k4"
Ms = [int(a) for a in sys.argv[2:]] or [18700]
lib = _lib.load()
dev, dt, Fd, d = "cuda", torch.bfloat16, 2048, 768
st = torch.cuda.current_stream().cuda_stream


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for M in Ms:
    g = torch.Generator(device=dev).manual_seed(0)
    feats = torch.randn(M, Fd, device=dev, generator=g).to(dt)
    lin = torch.nn.Linear(Fd, d).to(dev)
    gamma = torch.ones(d, device=dev); beta = torch.zeros(d, device=dev)
    packed = VisProjPackCache().get(lin.weight, lin.bias, 1)
    out = torch.empty(M, d, dtype=dt, device=dev); xhat = torch.empty_like(out); rstd = torch.empty(M, device=dev)
    dpre = torch.randn(M, d, device=dev, generator=g).to(dt)
    dw = torch.empty(d, Fd, device=dev); db = torch.empty(d, device=dev)
    nws = lib.vlpet_visproj_wgrad_workspace_bytes(M, Fd, d); ws = torch.empty(nws, dtype=torch.uint8, device=dev)

    def fwd():
        rc = lib.vlpet_visproj_fwd(feats.data_ptr(), packed.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 0, out.data_ptr(),
                                   xhat.data_ptr(), rstd.data_ptr(), M, Fd, d, 1e-5, 0, 1, st); assert rc == 0

    def wgrad():
        rc = lib.vlpet_visproj_wgrad(dpre.data_ptr(), feats.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nws, M, Fd, d, 1, st)
        assert rc == 0

    wb = lin.weight.detach().to(dt)
    bb = lin.bias.detach().to(dt)
    R = torch.zeros(M, d, dtype=dt, device=dev); mean = torch.empty(M, device=dev)

    def fwd_composed():      # the default K4 forward since round 4: library GEMM, then LayerNorm + residual in one pass of the K5 kernel
        pre = torch.addmm(bb, feats, wb.t())
        rc = lib.vlpet_norm_residual_fwd(pre.data_ptr(), R.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), mean.data_ptr(),
                                         rstd.data_ptr(), M, d, 1e-5, 1, st); assert rc == 0
    nwg = lib.vlpet_visproj_gemm_workspace_bytes(M, Fd, d); wsg = torch.zeros(max(nwg, 256), dtype=torch.uint8, device=dev)
    b32 = lin.bias.detach().float()

    def fwd_gemm(form, bm):      # round 5: the hand-written tiled GEMM + statistics exchange (csrc/visproj_gemm.hip), R added like the composed form
        def f():
            rc = lib.vlpet_visproj_fwd_gemm_cfg(feats.data_ptr(), wb.data_ptr(), b32.data_ptr(), gamma.data_ptr(), beta.data_ptr(), R.data_ptr(),
                                                out.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), None, wsg.data_ptr(), nwg, M, Fd, d, 1e-5, 0, 1,
                                                form, bm, st); assert rc == 0, rc
        return f
    gemm_cols = []
    for form, bm in ((0, 0), (4, 256), (4, 192), (4, 128), (1, 192)):
        tg = timed(fwd_gemm(form, bm)); torch.cuda.synchronize()
        refg = torch.nn.functional.layer_norm((feats.float() @ lin.weight.detach().float().t() + lin.bias.detach().float()), (d,), gamma, beta, 1e-5)
        eg = float((out.float() - refg).abs().max() / refg.abs().max())
        stat = int(wsg[:4].view(torch.int32)[0].item())
        st7 = wsg[64:64 + 56].view(torch.int64).tolist()
        if (form, bm) in ((0, 0), (4, 192)):      # workgroup 0's wall-clock stamps (10 ns units -> us)
            d7 = [(st7[k + 1] - st7[k]) / 100.0 for k in range(6)]
            print(f"   stamps form {form} bm {bm}: prologue {d7[0]:.1f}  K loop {d7[1]:.1f}  wave stats {d7[2]:.1f}  exchange {d7[3]:.1f}  norm + out {d7[4]:.1f}  xhat {d7[5]:.1f} us")
        gemm_cols.append(f"form {form} bm {bm:3d}: {tg:6.1f} us ({fl_ / tg / 1e6 / 2500:.3f}, err {eg:.1e}, status {stat})" if False else (form, bm, tg, eg, stat))
    t_fc = timed(fwd_composed)
    t_f, t_w = timed(fwd), timed(wgrad)
    t_lf = timed(lambda: torch.nn.functional.linear(feats, wb))
    t_lw = timed(lambda: dpre.t() @ feats)
    fl = 2.0 * M * Fd * d
    # parity of the two against the library results (bf16 tolerance)
    fwd(); wgrad(); torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm((feats.float() @ lin.weight.detach().float().t() + lin.bias.detach().float()), (d,), gamma, beta, 1e-5)
    e_f = float((out.float() - ref).abs().max() / ref.abs().max())
    refw = dpre.float().t() @ feats.float()
    e_w = float((dw - refw).abs().max() / refw.abs().max())
    print(f"k4bench {tag:8s} M={M:6d}: tiled GEMM + exchange (form, rows/WG: us, frac of 2.5 PF, err, status): "
          + "  ".join(f"({fo},{bm}: {tg:.1f}, {fl / tg / 1e6 / 2500:.3f}, {eg:.0e}, {stt})" for fo, bm, tg, eg, stt in gemm_cols), flush=True)
    print(f"k4bench {tag:8s} M={M:6d}: fwd (library GEMM + norm pass) {t_fc:7.1f} us ({fl / t_fc / 1e6:6.0f} TFLOP/s, frac {fl / t_fc / 1e6 / 2500:.3f})  |  fused kernel: "
          f"fwd {t_f:7.1f} us ({fl / t_f / 1e6:6.0f} TFLOP/s, frac {fl / t_f / 1e6 / 2500:.3f}, err {e_f:.1e})   "
          f"wgrad {t_w:7.1f} us ({fl / t_w / 1e6:6.0f} TFLOP/s, frac {fl / t_w / 1e6 / 2500:.3f}, err {e_w:.1e})   | library GEMMs alone: "
          f"fwd {t_lf:6.1f} us, wgrad {t_lw:6.1f} us", flush=True)
