"""Host glue of the short-sequence attention kernels (csrc/attn.hip): ``dropout(softmax(scale * q k^T + mask), p) v`` for
the sequence lengths VL-PET trains on (at most 128 keys / queries, head dim 64, bf16), forward and backward on chip, one
workgroup per (batch, head).

Reference op chain replaced: BartAttention.forward (my_transformers/modeling_bart.py:283-566): bmm, mask add, softmax,
F.dropout(p=attention_dropout), bmm, head transposes.  Tensors stay in the ``[B, L, H * 64]`` layout the projections
produce (no head transpose / re-pack in either direction).  ``supported`` tells the caller whether a call fits; anything
else (fp32 parity runs, long video sequences, per-sample additive biases) stays on the library path.  A bias shared by the
batch -- T5's relative position bias (my_transformers/modeling_t5.py:520-560, 640-660) -- is taken as ``AttnBias`` (round 4).
No CPU fallback."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .functional import _need_cuda, _ptr, _stream, _timed
from .tail import _draw_seed

MAX_LEN = 128
HEAD_DIM = 64


class AttnBias:
    """An additive score bias ``[1 or none, H, Lq, Lk]`` shared by every sample, in the layout the kernels read: fp32, both
    sequence axes zero-padded to multiples of 32, plus its transpose for the key-major phase of the backward.  Built once per
    forward (it is the same for every layer of a T5 stack) and passed to ``short_attention(..., bias=...)``.  No gradient."""

    def __init__(self, bias: torch.Tensor):
        b = bias.detach()
        if b.dim() == 4:
            if b.shape[0] != 1:
                raise RuntimeError("vl-pet_amd: AttnBias is shared by the batch ([1, H, Lq, Lk] or [H, Lq, Lk])")
            b = b[0]
        H, Lq, Lk = b.shape
        Lqp, Lkp = (Lq + 31) // 32 * 32, (Lk + 31) // 32 * 32
        self.H, self.Lq, self.Lk = H, Lq, Lk
        self.b = torch.zeros(H, Lqp, Lkp, dtype=torch.float32, device=b.device)
        self.b[:, :Lq, :Lk] = b.float()
        self.bt = self.b.transpose(1, 2).contiguous()


def supported(q: torch.Tensor, k: torch.Tensor, num_heads: int) -> bool:
    return (q.is_cuda and q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16 and q.shape[-1] == num_heads * HEAD_DIM
            and q.shape[1] <= MAX_LEN and k.shape[1] <= MAX_LEN)


class KeyGradSlot:
    """Where the key gradients of n attention calls go when their keys are the column blocks of ONE ``[B, Lk, n * E]`` buffer (the
    decoder layers' fused key projection of the encoder output, functional.cross_key_blocks): every call's backward writes its dk into
    block ``index`` of one gradient buffer of the same geometry, allocated by the first call that gets there, and the projection's
    backward consumes the whole buffer with a single GEMM."""

    def __init__(self, n: int, E: int):
        self.n, self.E, self.buf = n, E, None

    def block(self, k: torch.Tensor, index: int) -> torch.Tensor:
        if self.buf is None:
            self.buf = torch.empty(k.shape[0], k.shape[1], self.n * self.E, dtype=k.dtype, device=k.device)
        return self.buf[..., index * self.E:(index + 1) * self.E]


def _is_row_block(t: torch.Tensor) -> bool:
    """a [B, L, E] column block of a wider contiguous [B, L, ld] buffer that the kernels can read in place"""
    return (t.dim() == 3 and t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1) and t.stride(1) % 8 == 0
            and t.data_ptr() % 16 == 0)


class _AttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, key_mask, H, causal, scale, p, seed, want_mask, bias=None, k_slot=None):
        lib = _lib.load()
        _need_cuda(q, k, v)
        # (k_slot = (KeyGradSlot, index): k is a column block of the fused key projection's output and is read in place, row stride n * E)
        if k_slot is not None and not _is_row_block(k):
            k_slot = None
        q, v = q.contiguous(), v.contiguous()
        if k_slot is None:
            k = k.contiguous()
        ld_k = k.stride(1)
        B, Lq, _ = q.shape
        Lk = k.shape[1]
        if bias is not None and (bias.H, bias.Lq, bias.Lk) != (H, Lq, Lk):
            raise RuntimeError(f"vl-pet_amd: attention bias [{bias.H}, {bias.Lq}, {bias.Lk}] does not match [{H}, {Lq}, {Lk}]")
        o = torch.empty_like(q)
        lse = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device)
        # (the kernel writes the mask only when it drops something: without dropout every element is kept)
        keep = torch.ones(B, H, Lq, Lk, dtype=torch.uint8, device=q.device) if want_mask else None
        rc = _timed("attn_fwd", B * Lq, lambda: lib.vlpet_attn_fwd_kv(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), _ptr(key_mask), bias.b.data_ptr() if bias is not None else None,
            o.data_ptr(), lse.data_ptr(), _ptr(keep), B, H, Lq, Lk, H * HEAD_DIM, ld_k, H * HEAD_DIM, int(causal), float(scale), float(p), seed,
            _stream()))
        _lib.check(rc, "vlpet_attn_fwd")
        ctx.save_for_backward(q, k, v, o, lse, key_mask)
        ctx.bias = bias
        ctx.k_slot = k_slot
        ctx.cfg = (H, int(causal), float(scale), float(p), seed)
        if want_mask:
            ctx.mark_non_differentiable(keep)
            return o, keep
        return o

    @staticmethod
    def backward(ctx, dout, *unused):
        lib = _lib.load()
        q, k, v, o, lse, key_mask = ctx.saved_tensors
        H, causal, scale, p, seed = ctx.cfg
        B, Lq, _ = q.shape
        Lk = k.shape[1]
        do = dout.contiguous()
        if do.dtype != q.dtype:
            do = do.to(q.dtype)
        dq, dv = torch.empty_like(q), torch.empty_like(v)
        k_slot, ctx.k_slot = ctx.k_slot, None
        dk = torch.empty_like(k) if k_slot is None else k_slot[0].block(k, k_slot[1])      # (the slot's block has k's strides)
        assert dk.stride() == k.stride()
        bias = ctx.bias
        rc = _timed("attn_bwd", B * Lq, lambda: lib.vlpet_attn_bwd_kv(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), _ptr(key_mask),
            bias.b.data_ptr() if bias is not None else None, bias.bt.data_ptr() if bias is not None else None,
            dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, Lq, Lk, H * HEAD_DIM, k.stride(1), H * HEAD_DIM, causal, scale, p, seed, _stream()))
        _lib.check(rc, "vlpet_attn_bwd")
        ctx.bias = None
        return dq, dk, dv, None, None, None, None, None, None, None, None, None


class _AttnQkvFn(torch.autograd.Function):
    """Self-attention on a fused projection output ``qkv [B, L, 3*E]`` (columns q | k | v): the kernels read the three column
    blocks in place (row stride 3E) and the backward writes dq | dk | dv into ONE ``[B, L, 3E]`` gradient, which the fused
    projection's dgrad GEMM consumes directly."""

    @staticmethod
    def forward(ctx, qkv, key_mask, H, causal, scale, p, seed, bias=None):
        lib = _lib.load()
        _need_cuda(qkv)
        qkv = qkv.contiguous()
        B, L, E3 = qkv.shape
        E = E3 // 3
        if bias is not None and (bias.H, bias.Lq, bias.Lk) != (H, L, L):
            raise RuntimeError(f"vl-pet_amd: attention bias [{bias.H}, {bias.Lq}, {bias.Lk}] does not match [{H}, {L}, {L}]")
        o = torch.empty(B, L, E, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(B, H, L, dtype=torch.float32, device=qkv.device)
        base, esz = qkv.data_ptr(), qkv.element_size()
        rc = _timed("attn_fwd", B * L, lambda: lib.vlpet_attn_fwd_bias(
            base, base + E * esz, base + 2 * E * esz, _ptr(key_mask), bias.b.data_ptr() if bias is not None else None, o.data_ptr(),
            lse.data_ptr(), None, B, H, L, L, E3, E3, int(causal), float(scale), float(p), seed, _stream()))
        _lib.check(rc, "vlpet_attn_fwd_bias")
        ctx.save_for_backward(qkv, o, lse, key_mask)
        ctx.bias = bias
        ctx.cfg = (H, int(causal), float(scale), float(p), seed)
        return o

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        qkv, o, lse, key_mask = ctx.saved_tensors
        H, causal, scale, p, seed = ctx.cfg
        B, L, E3 = qkv.shape
        E = E3 // 3
        do = dout.contiguous()
        if do.dtype != qkv.dtype:
            do = do.to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        base, dbase, esz = qkv.data_ptr(), dqkv.data_ptr(), qkv.element_size()
        bias = ctx.bias
        rc = _timed("attn_bwd", B * L, lambda: lib.vlpet_attn_bwd_bias(
            base, base + E * esz, base + 2 * E * esz, o.data_ptr(), do.data_ptr(), lse.data_ptr(), _ptr(key_mask),
            bias.b.data_ptr() if bias is not None else None, bias.bt.data_ptr() if bias is not None else None,
            dbase, dbase + E * esz, dbase + 2 * E * esz, B, H, L, L, E3, E3, causal, scale, p, seed, _stream()))
        _lib.check(rc, "vlpet_attn_bwd_bias")
        ctx.bias = None
        return dqkv, None, None, None, None, None, None, None


def supported_qkv(qkv: torch.Tensor, num_heads: int) -> bool:
    return (qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.dim() == 3 and qkv.shape[-1] == 3 * num_heads * HEAD_DIM
            and qkv.shape[1] <= MAX_LEN)


# The kernels take the key mask as [B, Lk] bytes; the models hand every layer the same boolean mask (a fresh view of it per layer), so
# the conversion is kept for the last two sources: same storage, same version counter (an in-place write to the mask bumps it),
# same geometry -> the same bytes.  The entry holds the source view, which keeps its storage from being recycled under the key.
_KM_CACHE: list = []


def _key_mask_u8(key_mask: torch.Tensor, B: int, L: int) -> torch.Tensor:
    if key_mask.dtype == torch.uint8 and key_mask.shape == (B, L) and key_mask.is_contiguous():
        return key_mask
    if key_mask.is_inference():             # tensors made under torch.inference_mode() track no version: nothing to key the cache on
        return key_mask.reshape(B, L).to(torch.uint8).contiguous()
    for src, ver, out in _KM_CACHE:
        if (src.data_ptr() == key_mask.data_ptr() and ver == key_mask._version and src.dtype == key_mask.dtype
                and src.shape == key_mask.shape and src.stride() == key_mask.stride() and out.shape == (B, L)):
            return out
    out = key_mask.reshape(B, L).to(torch.uint8).contiguous()
    _KM_CACHE.insert(0, (key_mask, key_mask._version, out))
    del _KM_CACHE[2:]
    return out


def short_self_attention(qkv: torch.Tensor, num_heads: int, key_mask: Optional[torch.Tensor] = None, causal: bool = False,
                         p: float = 0.0, training: bool = False, scale: Optional[float] = None, seed=None,
                         bias: Optional[AttnBias] = None):
    """qkv [B, L, 3*H*64] (bf16; the output of one fused q|k|v projection) -> [B, L, H*64].  bias: see short_attention."""
    if not supported_qkv(qkv, num_heads):
        raise RuntimeError("vl-pet_amd: short_self_attention needs a bf16 CUDA [B, L, 3*H*64] tensor with L <= 128")
    if key_mask is not None:
        key_mask = _key_mask_u8(key_mask, qkv.shape[0], qkv.shape[1])
    pe = float(p) if training else 0.0
    if seed is None:
        seed = _draw_seed() if pe > 0.0 else 0
    return _AttnQkvFn.apply(qkv, key_mask, num_heads, bool(causal), HEAD_DIM ** -0.5 if scale is None else float(scale), pe, int(seed), bias)


def short_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int, key_mask: Optional[torch.Tensor] = None,
                    causal: bool = False, p: float = 0.0, training: bool = False, scale: Optional[float] = None, seed=None,
                    return_mask: bool = False, bias: Optional[AttnBias] = None, k_slot=None):
    """q [B, Lq, H*64], k / v [B, Lk, H*64] (bf16) -> [B, Lq, H*64].  key_mask: [B, Lk] bool / uint8, True = attend.
    bias: an ``AttnBias`` ([H, Lq, Lk] added to the scaled scores of every sample; no gradient).  k_slot: (KeyGradSlot, index) when
    ``k`` is a column block of a fused key projection (read in place; its gradient goes into the slot's shared buffer)."""
    if not supported(q, k, num_heads):
        raise RuntimeError("vl-pet_amd: short_attention needs bf16 CUDA tensors, head dim 64 and at most 128 keys / queries")
    if key_mask is not None:
        key_mask = _key_mask_u8(key_mask, k.shape[0], k.shape[1])
    pe = float(p) if training else 0.0
    if seed is None:
        seed = _draw_seed() if pe > 0.0 else 0
    return _AttnFn.apply(q, k, v, key_mask, num_heads, bool(causal), HEAD_DIM ** -0.5 if scale is None else float(scale),
                         pe, int(seed), bool(return_mask), bias, k_slot)
