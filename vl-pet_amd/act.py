"""Host glue of the FFN activation + dropout pass (csrc/actdrop.hip): ``dropout(act(x), p)`` between fc1 and fc2 of the
backbone's feed-forward sublayer as one HIP pass forward and one backward.

Reference op chains replaced: my_transformers/modeling_bart.py:1264-1265 (encoder), :1750-1756 (decoder);
my_transformers/modeling_t5.py:262-265 (T5DenseReluDense).  Only the pre-activation is kept for the backward (the
reference keeps x, act(x) and a byte mask); the mask is a function of (seed, element index) and is regenerated.
One 64-bit seed per call from torch's CPU generator, as the sublayer tail does.  No CPU fallback."""
from __future__ import annotations

import torch

from . import _lib
from .functional import _io_dtype, _need_cuda, _ptr, _stream, _timed
from .tail import _draw_seed

ACTS = {"gelu": 0, "gelu_new": 1, "relu": 2}


class _ActDropFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, p, seed, want_mask):
        lib = _lib.load()
        _need_cuda(x)
        xc = x.contiguous()
        n = xc.numel()
        io = _io_dtype(xc)
        out = torch.empty_like(xc)
        mask = torch.empty(xc.shape, dtype=torch.uint8, device=x.device) if want_mask else None
        rc = _timed("ffn_act_fwd", n // xc.shape[-1], lambda: lib.vlpet_act_dropout_fwd(
            xc.data_ptr(), out.data_ptr(), _ptr(mask), n, act, float(p), seed, io, _stream()))
        _lib.check(rc, "vlpet_act_dropout_fwd")
        ctx.save_for_backward(xc)
        ctx.cfg = (act, float(p), seed, io)
        if want_mask:
            ctx.mark_non_differentiable(mask)
            return out, mask
        return out

    @staticmethod
    def backward(ctx, dout, *unused):
        lib = _lib.load()
        (xc,) = ctx.saved_tensors
        act, p, seed, io = ctx.cfg
        dy = dout.contiguous()
        if dy.dtype != xc.dtype:
            dy = dy.to(xc.dtype)
        dx = torch.empty_like(xc)
        n = xc.numel()
        rc = _timed("ffn_act_bwd", n // xc.shape[-1], lambda: lib.vlpet_act_dropout_bwd(
            dy.data_ptr(), xc.data_ptr(), dx.data_ptr(), n, act, p, seed, io, _stream()))
        _lib.check(rc, "vlpet_act_dropout_bwd")
        return dx, None, None, None, None


def act_dropout(x: torch.Tensor, act: str = "gelu", p: float = 0.0, training: bool = False, seed=None,
                return_mask: bool = False):
    """``F.dropout(act(x), p, training)`` in one pass; ``act`` in {"gelu", "gelu_new", "relu"}.  The last dimension
    must be a multiple of 8 (every FFN width is)."""
    if act not in ACTS:
        raise RuntimeError(f"vl-pet_amd: unsupported FFN activation {act!r} (expected one of {sorted(ACTS)})")
    if x.shape[-1] % 8 != 0:
        raise RuntimeError("vl-pet_amd: act_dropout needs a last dimension that is a multiple of 8")
    pe = float(p) if training else 0.0
    if seed is None:
        seed = _draw_seed() if pe > 0.0 else 0
    return _ActDropFn.apply(x, ACTS[act], pe, int(seed), bool(return_mask))


# ------------------------------------------------------------------------------------------------ joint-encoder input assembly
FUSE_CONCAT_DROPOUT = True      # A/B switch (tools/ab_switches.py: VLPET_NO_CONCAT_DROPOUT=1): torch.cat + F.dropout instead


class _ConcatDropFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, v, p, seed):
        lib = _lib.load()
        _need_cuda(a, v)
        ac, vc = a.contiguous(), v.contiguous()
        B, La, d = ac.shape
        Lv = vc.shape[1]
        io = _io_dtype(ac)
        x = torch.empty(B, La + Lv, d, dtype=ac.dtype, device=ac.device)
        rc = _timed("concat_drop_fwd", B * (La + Lv), lambda: lib.vlpet_concat_dropout_fwd(
            ac.data_ptr(), vc.data_ptr(), x.data_ptr(), B, La, Lv, d, float(p), seed, io, _stream()))
        _lib.check(rc, "vlpet_concat_dropout_fwd")
        ctx.cfg = (B, La, Lv, d, float(p), seed, io, ac.dtype)
        return x

    @staticmethod
    def backward(ctx, dx):
        lib = _lib.load()
        B, La, Lv, d, p, seed, io, dtype = ctx.cfg
        dxc = dx.contiguous()
        if dxc.dtype != dtype:
            dxc = dxc.to(dtype)
        da = torch.empty(B, La, d, dtype=dtype, device=dx.device) if ctx.needs_input_grad[0] else None
        dv = torch.empty(B, Lv, d, dtype=dtype, device=dx.device) if ctx.needs_input_grad[1] else None
        if da is None and dv is None:
            return None, None, None, None
        rc = _timed("concat_drop_bwd", B * (La + Lv), lambda: lib.vlpet_concat_dropout_bwd(
            dxc.data_ptr(), _ptr(da), _ptr(dv), B, La, Lv, d, p, seed, io, _stream()))
        _lib.check(rc, "vlpet_concat_dropout_bwd")
        return da, dv, None, None


def concat_dropout(a: torch.Tensor, v: torch.Tensor, p: float = 0.0, training: bool = False, seed=None) -> torch.Tensor:
    """``F.dropout(torch.cat([a, v], dim=1), p, training)`` for ``a [B, La, d]``, ``v [B, Lv, d]`` in one HIP pass each way
    (src/modeling_bart.py:804-820: the joint encoder's text | visual concatenation and the dropout behind it); plain torch ops where
    the kernel does not apply (CPU tensors, other dtypes, widths not a multiple of 8)."""
    pe = float(p) if training else 0.0
    ok = (FUSE_CONCAT_DROPOUT and a.is_cuda and v.is_cuda and a.dim() == 3 and v.dim() == 3 and a.dtype == v.dtype
          and a.dtype in (torch.bfloat16, torch.float32) and a.shape[0] == v.shape[0] and a.shape[2] == v.shape[2] and a.shape[2] % 8 == 0
          and a.shape[0] > 0 and a.shape[1] > 0 and v.shape[1] > 0)
    if not ok:
        return torch.nn.functional.dropout(torch.cat([a, v], dim=1), p=p, training=training)
    if seed is None:
        seed = _draw_seed() if pe > 0.0 else 0
    return _ConcatDropFn.apply(a, v, pe, int(seed))
