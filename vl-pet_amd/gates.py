"""Host glue for the small / middleX / middleY granularity gates (csrc/rowgate.hip).

Reference inline code: my_transformers/modeling_bart.py:1210-1231 (attention sublayer), :1326-1347 (FFN);
T5: my_transformers/modeling_t5.py:391-403, 807-819.  ``h`` is the adapter output ``s2*x2 + sd*delta`` of
the fused K1 kernel in adapter-only mode; every [M, d] pass below is a HIP row kernel, only O(M) vectors
(one scalar per token / sample) are handled with torch ops.  No CPU fallback."""
from __future__ import annotations

import torch

from . import _lib
from .functional import _empty_result as _empty
from .functional import _flat, _grad_like, _io_dtype, _need_cuda, _ptr, _stream, _timed


def _row_dot(lib, a, c, wa, wc, M, d, io):
    s = torch.empty(M, dtype=torch.float32, device=a.device)
    _lib.check(lib.vlpet_row_dot(a.data_ptr(), _ptr(c), _ptr(wa), _ptr(wc), s.data_ptr(), M, d, io, _stream()),
               "vlpet_row_dot")
    return s


class _RowGateFn(torch.autograd.Function):
    """small (per-sample scalar) and middleX (per-token scalar) gates."""

    @staticmethod
    def forward(ctx, x1, h, w, b, small, gating_add, gs):
        lib = _lib.load()
        _need_cuda(x1, h)
        d = h.shape[-1]
        io = _io_dtype(h)
        if x1.dtype != h.dtype:
            x1 = x1.to(h.dtype)
        x1f, hf = _flat(x1, d), _flat(h, d)
        M = hf.shape[0]
        S = h.shape[-2]
        w32 = w.detach().float().reshape(-1)
        wa = w32[:d].contiguous()
        wc = w32[d:].contiguous() if small else wa
        s = _timed("k1g_dot", M, lambda: _row_dot(lib, x1f, hf, wa, wc, M, d, io))
        g_row = torch.sigmoid(s + b.detach().float().reshape(()))
        g_exp = g_row.view(-1, S).mean(1).repeat_interleave(S) if small else g_row
        if gating_add:
            alpha, gamma = torch.full_like(g_exp, gs), gs * g_exp
        else:
            alpha, gamma = gs * g_exp, None
        y = torch.empty_like(hf)
        rc = _timed("k1g_apply", M, lambda: lib.vlpet_row_affine(hf.data_ptr(), alpha.data_ptr(), _ptr(gamma),
                                                                 y.data_ptr(), M, d, io, _stream()))
        _lib.check(rc, "vlpet_row_affine")
        ctx.save_for_backward(x1f, hf, g_row, alpha, wa, wc, w, b)
        ctx.cfg = (bool(small), bool(gating_add), float(gs), S, h.shape, x1.shape, io)
        return y.view(h.shape)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x1f, hf, g_row, alpha, wa, wc, w, b = ctx.saved_tensors
        small, gating_add, gs, S, hshape, x1shape, io = ctx.cfg
        M, d = hf.shape
        dyf = _flat(dy, d)
        if gating_add:       # y = gs*(h + g):  dL/dg = gs * sum_j dy_j
            t = _row_dot(lib, dyf, None, torch.ones(d, dtype=torch.float32, device=dyf.device), None, M, d, io)
        else:                # y = gs*h*g:      dL/dg = gs * sum_j dy_j h_j
            t = _row_dot(lib, dyf, hf, None, None, M, d, io)
        dg = gs * t
        if small:
            dg = (dg.view(-1, S).sum(1) / S).repeat_interleave(S)
        beta = (dg * g_row * (1.0 - g_row)).contiguous()
        dh, dx1 = torch.empty_like(hf), torch.empty_like(hf)
        nb = lib.vlpet_rowgate_partials(M)
        part = torch.empty(nb, 2, d, dtype=torch.float32, device=dyf.device)
        rc = _timed("k1g_bwd", M, lambda: lib.vlpet_rowgate_bwd(
            dyf.data_ptr(), x1f.data_ptr(), hf.data_ptr(), alpha.data_ptr(), beta.data_ptr(), wa.data_ptr(),
            wc.data_ptr(), dh.data_ptr(), dx1.data_ptr(), part.data_ptr(), M, d, io, _stream()))
        _lib.check(rc, "vlpet_rowgate_bwd")
        ps = part.sum(0)
        dw = torch.cat([ps[0], ps[1]]) if small else ps[0] + ps[1]
        dw = _grad_like(dw.view(w.shape), w)
        db = _grad_like(beta.sum().view(b.shape), b)
        return dx1.view(x1shape), dh.view(hshape), dw, db, None, None, None


class _VecGateFn(torch.autograd.Function):
    """middleY: y = (h + h*z) * gs   or, with gating_add, (h + 1 + z) * gs."""

    @staticmethod
    def forward(ctx, h, z, gating_add, gs):
        lib = _lib.load()
        _need_cuda(h)
        d = h.shape[-1]
        io = _io_dtype(h)
        hf = _flat(h, d)
        M = hf.shape[0]
        z32 = z.detach().float()
        if gating_add:
            v, u = torch.full_like(z32, gs), (gs * (1.0 + z32)).contiguous()
        else:
            v, u = (gs * (1.0 + z32)).contiguous(), None
        y = torch.empty_like(hf)
        rc = _timed("k1g_apply", M, lambda: lib.vlpet_vecgate_fwd(hf.data_ptr(), v.data_ptr(), _ptr(u), y.data_ptr(),
                                                                  M, d, io, _stream()))
        _lib.check(rc, "vlpet_vecgate_fwd")
        ctx.save_for_backward(hf, v, z)
        ctx.cfg = (bool(gating_add), float(gs), h.shape, io)
        return y.view(h.shape)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        hf, v, z = ctx.saved_tensors
        gating_add, gs, hshape, io = ctx.cfg
        M, d = hf.shape
        dyf = _flat(dy, d)
        dh = torch.empty_like(hf)
        nb = lib.vlpet_rowgate_partials(M)
        part = torch.empty(nb, 2, d, dtype=torch.float32, device=dyf.device)
        rc = _timed("k1g_bwd", M, lambda: lib.vlpet_vecgate_bwd(dyf.data_ptr(), hf.data_ptr(), v.data_ptr(),
                                                                dh.data_ptr(), part.data_ptr(), M, d, io, _stream()))
        _lib.check(rc, "vlpet_vecgate_bwd")
        ps = part.sum(0)
        dz = gs * (ps[1] if gating_add else ps[0])
        return dh.view(hshape), _grad_like(dz.view(z.shape), z), None, None


def small_gate(x1, h, linear, gating_add=False, gate_scale=1.0):
    """VL-PET-small: ``linear`` = Linear(2d, 1) on cat(x1, h); sigmoid; mean over the sequence."""
    if h.dim() != 3:
        raise ValueError("the small gate averages over the sequence axis: expected [B, S, d]")
    if h.numel() == 0:
        return _empty(h, [linear.weight, linear.bias])
    return _RowGateFn.apply(x1, h, linear.weight, linear.bias, True, gating_add, gate_scale)


def middle_x_gate(x1, h, linear, gating_add=False, gate_scale=1.0):
    """VL-PET-middleX: ``linear`` = Linear(d, 1) on x1 + h; sigmoid; one scalar per token."""
    if h.numel() == 0:
        return _empty(h, [linear.weight, linear.bias])
    return _RowGateFn.apply(x1, h, linear.weight, linear.bias, False, gating_add, gate_scale)


def middle_y_gate(h, z, gating_add=False, gate_scale=1.0):
    """VL-PET-middleY: one learnable vector z in R^d (IA3-style)."""
    if h.numel() == 0:
        return _empty(h, [z])
    return _VecGateFn.apply(h, z, gating_add, gate_scale)
