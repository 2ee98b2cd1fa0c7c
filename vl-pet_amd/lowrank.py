"""f4 host glue: the feature branch of ``LowRankVisualEmbedding`` (src/modeling_bart.py:278-299, 324-325)

    fe  = up(gelu_new(cat_i down_i(feats)))
    fe  = fe * sigmoid(gup(gelu_new(gdown(feats))))   [+ fe with use_visual_projector_residual_connection]
    out = LayerNorm(fe) + R                           (R = position branch + order embeddings)

on the HIP path: the rectangular K1 kernels (csrc/pet_gate_fwd.hip / pet_gate_bwd2.hip, LR form: feat_dim-wide input,
d_model-wide output, both chains on the same feature rows, no residual, no input gradients), the weight-gradient kernels
(csrc/wgrad.hip) and the K5 kernel with the residual joining after the norm (csrc/tail.hip, POST form).  The features are
data: they get no gradient.  No CPU fallback."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch

from . import _lib
from . import functional as Fn
from .functional import _flat, _grad_dest, _grad_like, _io_dtype, _need_cuda, _param_dtype, _ptr, _stream, _timed


class LowRankPack:
    """One projection pair of the projector in fragment order: [square pair pack at d_out | down-only pack at feat_dim]
    (vlpet_lowrank_pack)."""

    __slots__ = ("buf", "tiles", "r", "feat_dim", "d_out", "io_dtype")

    def __init__(self, buf, tiles, r, feat_dim, d_out, io_dtype):
        self.buf, self.tiles, self.r, self.feat_dim, self.d_out, self.io_dtype = buf, tiles, r, feat_dim, d_out, io_dtype


def lowrank_tiles(r: int, rg: int = 0) -> int:
    """Padded rank / 32 shared by both chains; the rectangular kernels exist for 1 and 3 (r, r_g <= 96)."""
    t = max(Fn.rank_tiles(r), Fn.rank_tiles(rg) if rg else 1)
    if t not in (1, 3):
        raise NotImplementedError(f"vl-pet_amd: low-rank visual projector with bottleneck {max(r, rg)} > 96")
    return t


def pack_lowrank(down_w: Sequence[torch.Tensor], down_b: Sequence[torch.Tensor], up_w: torch.Tensor, up_b: torch.Tensor,
                 io_dtype: int, tiles: int) -> LowRankPack:
    lib = _lib.load()
    _need_cuda(up_w, *down_w)
    n = len(down_w)
    rh, feat_dim = down_w[0].shape
    r = rh * n
    d_out = up_w.shape[0]
    if tuple(up_w.shape) != (d_out, r):
        raise RuntimeError(f"vl-pet_amd: up weight {tuple(up_w.shape)} does not match a rank-{r} down projection")
    ws = [w.detach().contiguous() for w in down_w]
    bs = [b.detach().contiguous() for b in down_b]
    uw, ub = up_w.detach().contiguous(), up_b.detach().contiguous()
    pd = _param_dtype(uw)
    if any(_param_dtype(t) != pd for t in ws + bs + [ub]):
        raise RuntimeError("vl-pet_amd: mixed parameter dtypes in one projection pair")
    nbytes = lib.vlpet_lowrank_packed_bytes(tiles, feat_dim, d_out, io_dtype)
    if nbytes == 0:
        raise RuntimeError(f"vl-pet_amd: low-rank projector geometry not supported (feat_dim {feat_dim}, d_model {d_out}: "
                           "both must be multiples of 64)")
    buf = torch.empty(nbytes, dtype=torch.uint8, device=uw.device)
    arr_w = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
    arr_b = (ctypes.c_void_p * n)(*[b.data_ptr() for b in bs])
    rc = lib.vlpet_lowrank_pack(arr_w, arr_b, n, uw.data_ptr(), ub.data_ptr(), r, feat_dim, d_out, tiles, pd, io_dtype,
                                buf.data_ptr(), _stream())
    _lib.check(rc, "vlpet_lowrank_pack")
    return LowRankPack(buf, tiles, r, feat_dim, d_out, io_dtype)


class LowRankPackCache:
    """Re-pack only when a parameter changed: keyed, like functional.PackCache, on the source parameters' addresses and
    versions plus WEIGHTS_EPOCH (the fused optimizer writes through raw pointers)."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, down_w, down_b, up_w, up_b, io_dtype, tiles) -> LowRankPack:
        ts = list(down_w) + list(down_b) + [up_w, up_b]
        key = (io_dtype, tiles, Fn.WEIGHTS_EPOCH) + tuple((t.data_ptr(), t._version) for t in ts)
        if key != self._key:
            self._val = pack_lowrank(down_w, down_b, up_w, up_b, io_dtype, tiles)
            self._key = key
        return self._val


class _LowRankProjFn(torch.autograd.Function):
    """inputs: feats, R, then N_h down weights, N_h down biases, up w, up b, [gate down w/b, gate up w/b], LN gamma, beta."""

    @staticmethod
    def forward(ctx, feats, R, pk_a, pk_g, n_heads, gate_residual, eps, *params):
        lib = _lib.load()
        _need_cuda(feats, R)
        io = _io_dtype(feats)
        F_, d = pk_a.feat_dim, pk_a.d_out
        ff = _flat(feats, F_)
        Rf = _flat(R.to(feats.dtype), d)
        M = ff.shape[0]
        gamma, beta = params[-2], params[-1]
        dev = ff.device
        need_bwd = any(p.requires_grad for p in params) or R.requires_grad
        fe = torch.empty(M, d, dtype=ff.dtype, device=dev)
        act = None
        if need_bwd:
            act = torch.empty(lib.vlpet_saved_bytes(M, pk_a.tiles, io), dtype=torch.uint8, device=dev)
        rc = _timed("f4_fwd", M, lambda: lib.vlpet_lowrank_gate_fwd(
            ff.data_ptr(), pk_a.buf.data_ptr(), pk_g.buf.data_ptr() if pk_g is not None else None, fe.data_ptr(),
            _ptr(act), M, F_, d, pk_a.tiles, int(gate_residual), io, _stream()))
        _lib.check(rc, "vlpet_lowrank_gate_fwd")
        g32 = gamma.detach().float().contiguous()
        b32 = beta.detach().float().contiguous()
        mean = torch.empty(M, dtype=torch.float32, device=dev)
        rstd = torch.empty(M, dtype=torch.float32, device=dev)
        out = torch.empty_like(fe)
        rc = _timed("f4_norm_fwd", M, lambda: lib.vlpet_norm_residual_fwd(
            fe.data_ptr(), Rf.data_ptr(), g32.data_ptr(), b32.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
            M, d, float(eps), io, _stream()))
        _lib.check(rc, "vlpet_norm_residual_fwd")
        ctx.save_for_backward(ff, fe, mean, rstd, g32, *params)
        ctx.act = act
        ctx.pk = (pk_a, pk_g)
        ctx.cfg = (n_heads, int(gate_residual), feats.shape[:-1] + (d,), R.shape, R.dtype)
        return out.view(feats.shape[:-1] + (d,))

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        ff, fe, mean, rstd, g32, *params = ctx.saved_tensors
        pk_a, pk_g = ctx.pk
        n_heads, gate_residual, oshape, rshape, rdtype = ctx.cfg
        act = ctx.act
        if act is None:
            raise RuntimeError("vl-pet_amd: low-rank projector backward without the forward's saved activations")
        M, d = fe.shape
        F_ = pk_a.feat_dim
        io = _io_dtype(fe)
        dev = fe.device
        gated = pk_g is not None
        df = _flat(dout.to(fe.dtype), d)
        # (i) LayerNorm backward (the K5 backward with h = fe, no dropout): d/dfe and the partial sums of dgamma / dbeta
        dfe = torch.empty_like(fe)
        nb = lib.vlpet_sublayer_tail_partials(M)
        part = torch.empty(nb, 2, d, dtype=torch.float32, device=dev)
        rc = _timed("f4_norm_bwd", M, lambda: lib.vlpet_sublayer_tail_bwd(
            df.data_ptr(), fe.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g32.data_ptr(), dfe.data_ptr(), None,
            part.data_ptr(), M, d, 0.0, 0, 1, io, _stream()))
        _lib.check(rc, "vlpet_sublayer_tail_bwd")
        s = part.sum(0)
        # (ii) rows kernel (dh, dq, dpre from the saved activations) + weight gradients
        r = pk_a.r
        rg = pk_g.r if gated else 0
        nh2 = 2 * n_heads
        (dwd, s_wd), (dbd, s_bd) = _grad_dest(params[0], (r, F_), block=True), _grad_dest(params[n_heads], (r,), block=True)
        (dwu, s_wu), (dbu, s_bu) = _grad_dest(params[nh2], (d, r)), _grad_dest(params[nh2 + 1], (d,))
        dwgd = dbgd = dwgu = dbgu = None
        if gated:
            (dwgd, s_gd), (dbgd, s_gdb) = _grad_dest(params[nh2 + 2], (rg, F_)), _grad_dest(params[nh2 + 3], (rg,))
            (dwgu, s_gu), (dbgu, s_gub) = _grad_dest(params[nh2 + 4], (d, rg)), _grad_dest(params[nh2 + 5], (d,))
        nws = lib.vlpet_lowrank_bwd_workspace_bytes(M, F_, d, pk_a.tiles, io)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        rc = _timed("f4_bwd", M, lambda: lib.vlpet_lowrank_gate_bwd(
            dfe.data_ptr(), ff.data_ptr(), act.data_ptr(), pk_a.buf.data_ptr(), pk_g.buf.data_ptr() if gated else None,
            dwd.data_ptr(), dbd.data_ptr(), dwu.data_ptr(), dbu.data_ptr(), _ptr(dwgd), _ptr(dbgd), _ptr(dwgu), _ptr(dbgu),
            r, rg, ws.data_ptr(), nws, M, F_, d, pk_a.tiles, gate_residual, io, _stream()))
        ctx.act = None
        _lib.check(rc, "vlpet_lowrank_gate_bwd")
        rh = r // n_heads
        grads: List[Optional[torch.Tensor]] = []
        if s_wd is not None:
            s_wd.done()
            grads += [None] * n_heads
        else:
            grads += [_grad_like(dwd[i * rh:(i + 1) * rh], params[i]) for i in range(n_heads)]
        if s_bd is not None:
            s_bd.done()
            grads += [None] * n_heads
        else:
            grads += [_grad_like(dbd[i * rh:(i + 1) * rh], params[n_heads + i]) for i in range(n_heads)]
        grads += Fn._finish([(dwu, s_wu, params[nh2]), (dbu, s_bu, params[nh2 + 1])])
        if gated:
            grads += Fn._finish([(dwgd, s_gd, params[nh2 + 2]), (dbgd, s_gdb, params[nh2 + 3]),
                                 (dwgu, s_gu, params[nh2 + 4]), (dbgu, s_gub, params[nh2 + 5])])
        grads += [_grad_like(s[0], params[-2]), _grad_like(s[1], params[-1])]
        dR = dout.reshape(rshape).to(rdtype) if ctx.needs_input_grad[1] else None      # the residual joins after the norm
        return (None, dR, None, None, None, None, None, *grads)


def lowrank_project(feats: torch.Tensor, R: torch.Tensor, down: Sequence[torch.nn.Linear], up: torch.nn.Linear,
                    norm: torch.nn.LayerNorm, gate_down: Optional[torch.nn.Linear], gate_up: Optional[torch.nn.Linear],
                    gate_residual: bool, cache_a: LowRankPackCache, cache_g: LowRankPackCache) -> torch.Tensor:
    """``LayerNorm(fe) + R`` with fe as in the module docstring.  ``R``: [..., d_model], same leading shape as ``feats``."""
    io = _io_dtype(feats)
    gated = gate_down is not None
    r = sum(m.weight.shape[0] for m in down)
    rg = gate_down.weight.shape[0] if gated else 0
    tiles = lowrank_tiles(r, rg)
    dw, db = [m.weight for m in down], [m.bias for m in down]
    pk_a = cache_a.get(dw, db, up.weight, up.bias, io, tiles)
    params = dw + db + [up.weight, up.bias]
    pk_g = None
    if gated:
        pk_g = cache_g.get([gate_down.weight], [gate_down.bias], gate_up.weight, gate_up.bias, io, tiles)
        params += [gate_down.weight, gate_down.bias, gate_up.weight, gate_up.bias]
    params += [norm.weight, norm.bias]
    if feats.numel() == 0:
        return Fn._empty_result(R.to(feats.dtype), params)
    return _LowRankProjFn.apply(feats, R, pk_a, pk_g, len(down), bool(gate_residual), norm.eps, *params)
