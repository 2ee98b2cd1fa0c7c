#!/usr/bin/env python3
"""f4 (low-rank visual projector, gated) per launch at the sizes given, through the C ABI, next to the library composition it
replaced (bf16 F.linear / gelu / sigmoid / layer_norm chain with autograd) on the same box.
usage: f4bench.py [M ...]   (default 18700 = the bench's average visual rows per step; feat_dim 2048, d 768, r = r_g = 96)

Algorithmic bytes per row (bf16): forward reads the features once (2*F) and writes fe (2*d) = 5,632 B at F = 2048; the
norm pass reads fe, R and writes out (6*d); the backward reads dfe (2*d) and the features once (2*F) = 5,632 B (weight
gradients are 0.6 MB)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as TF
import vlpet_amd.functional as F
from vlpet_amd import _lib
from vlpet_amd.lowrank import pack_lowrank
from kbench import timeit


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3)))


def run(M, Fd=2048, d=768, r=96, nh=4, dt=torch.bfloat16):
    dev = "cuda"
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.03
    x = torch.randn(M, Fd, device=dev, generator=g).to(dt)
    R = torch.randn(M, d, device=dev, generator=g).to(dt)
    dout = torch.randn(M, d, device=dev, generator=g).to(dt)
    wd, bd = [mk(r // nh, Fd) for _ in range(nh)], [mk(r // nh) for _ in range(nh)]
    wu, bu = mk(d, r), mk(d)
    gwd, gbd, gwu, gbu = mk(r, Fd), mk(r), mk(d, r), mk(d)
    gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    io = F._io_dtype(x)
    pa = pack_lowrank(wd, bd, wu, bu, io, 3)
    pg = pack_lowrank([gwd], [gbd], gwu, gbu, io, 3)
    st = torch.cuda.current_stream().cuda_stream
    fe, out, dfe = torch.empty(M, d, dtype=dt, device=dev), torch.empty(M, d, dtype=dt, device=dev), torch.empty(M, d, dtype=dt, device=dev)
    saved = torch.empty(lib.vlpet_saved_bytes(M, 3, io), dtype=torch.uint8, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    nws = lib.vlpet_lowrank_bwd_workspace_bytes(M, Fd, d, 3, io)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    G = [torch.empty(r, Fd, **f32), torch.empty(r, **f32), torch.empty(d, r, **f32), torch.empty(d, **f32),
         torch.empty(r, Fd, **f32), torch.empty(r, **f32), torch.empty(d, r, **f32), torch.empty(d, **f32)]
    part = torch.empty(lib.vlpet_sublayer_tail_partials(M), 2, d, **f32)
    chk = lambda rc: (_ for _ in ()).throw(RuntimeError(rc)) if rc else None
    fwd = lambda: chk(lib.vlpet_lowrank_gate_fwd(x.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(), fe.data_ptr(), saved.data_ptr(),
                                                 M, Fd, d, 3, 0, io, st))
    nfw = lambda: chk(lib.vlpet_norm_residual_fwd(fe.data_ptr(), R.data_ptr(), gam.data_ptr(), bet.data_ptr(), out.data_ptr(),
                                                  mean.data_ptr(), rstd.data_ptr(), M, d, 1e-5, io, st))
    nbw = lambda: chk(lib.vlpet_sublayer_tail_bwd(dout.data_ptr(), fe.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gam.data_ptr(),
                                                  dfe.data_ptr(), None, part.data_ptr(), M, d, 0.0, 0, 1, io, st))
    bwd = lambda: chk(lib.vlpet_lowrank_gate_bwd(dfe.data_ptr(), x.data_ptr(), saved.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(),
                                                 *[t.data_ptr() for t in G], r, r, ws.data_ptr(), nws, M, Fd, d, 3, 0, io, st))
    fwd(); nfw(); nbw(); bwd()
    t_f, t_nf, t_nb, t_b = timeit(fwd), timeit(nfw), timeit(nbw), timeit(bwd)
    esz = 2
    b_f, b_n, b_b = (Fd + d) * esz * M, 3 * d * esz * M, (Fd + d) * esz * M
    gb = lambda b, t: b / t / 1e3
    print(f"f4bench M={M:6d} F={Fd}: fwd {t_f:6.1f} us ({gb(b_f, t_f):6.0f} GB/s, {gb(b_f, t_f) / 8000:.3f} of 8 TB/s)   "
          f"norm fwd {t_nf:5.1f} us ({gb(b_n, t_nf) / 8000:.3f})   norm bwd {t_nb:5.1f} us   "
          f"bwd rows+wgrad {t_b:6.1f} us ({gb(b_b, t_b):6.0f} GB/s, {gb(b_b, t_b) / 8000:.3f})   total {t_f + t_nf + t_nb + t_b:6.1f} us")

    # the library composition this replaced (what visual.py ran for the gated form before): bf16 ops with autograd
    P = [t.to(dt).requires_grad_(True) for t in wd + bd + [wu, bu, gwd, gbd, gwu, gbu]] + [gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)]
    def lib_step():
        W, B = P[:nh], P[nh:2 * nh]
        z = gelu_new(torch.cat([TF.linear(x, w, b) for w, b in zip(W, B)], -1))
        u = TF.linear(z, P[2 * nh], P[2 * nh + 1])
        gt = torch.sigmoid(TF.linear(gelu_new(TF.linear(x, P[2 * nh + 2], P[2 * nh + 3])), P[2 * nh + 4], P[2 * nh + 5]))
        o = TF.layer_norm((u * gt).float(), (d,), P[-2], P[-1], 1e-5).to(dt) + R
        o.backward(dout)
        for p in P: p.grad = None
    t_lib = timeit(lib_step, iters=20, warm=5)
    print(f"f4bench M={M:6d} F={Fd}: library composition (bf16, fwd + bwd with autograd) {t_lib:7.1f} us  ->  HIP path {t_lib / (t_f + t_nf + t_nb + t_b):.2f}x")


if __name__ == "__main__":
    for M in ([int(a) for a in sys.argv[1:]] or [18700]):
        run(M)
    run(3200, Fd=512)          # the video config's visual rows per step (50 clips x 64 pooled tokens, clip-vit 512)
