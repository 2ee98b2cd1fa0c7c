"""K5 host glue: the sublayer tail ``LayerNorm(residual + dropout(y))`` (BART) / ``residual + dropout(y)``
(T5) as one HIP pass forward and one backward (csrc/tail.hip).

Reference op chains replaced: my_transformers/modeling_bart.py:1259-1261, 1375-1377 (encoder),
:1489-1491, 1513-1515, 1527-1529 (decoder); my_transformers/modeling_t5.py:408, 824.  The dropout mask
comes from the kernel's counter-based generator: one 64-bit seed per call, drawn from torch's CPU
generator (so ``torch.manual_seed`` makes a run repeatable); the backward regenerates the mask.
No CPU fallback."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .functional import _finish, _flat, _grad_dest, _grad_like, _io_dtype, _need_cuda, _ptr, _stream, _timed


# K5 with a LayerNorm: the backward needs the normalised rows.  Two forms: (a) the forward saves nothing but its own output (which
# the next sublayer reads anyway) + rstd, and the backward recovers xhat = (out - beta) / gamma -- 3 row tensors of traffic instead
# of 4; (b) the forward also writes the pre-norm sum h for the backward (the round-1..3 form, exact).  (a) divides the rounding of the
# IO-dtype output by gamma: the recovered xhat is off by about (|xhat| + |beta / gamma|) * 2^-9 in bf16, i.e. it loses
# log2(1 + |beta / gamma|) bits -- invisible for gamma ~ 1, beta ~ 0.1 (random init), NOT for a checkpoint whose LayerNorm has outlier
# channels (small gamma, large beta) or a trainable LayerNorm that drifts there.
# SAVE_PRENORM = None (default, round 5): chosen per LayerNorm from max |beta / gamma| -- form (a) up to PRENORM_RATIO, (b) above it or
# where a gamma is 0; frozen parameters are looked at once (keyed on storage + version), trainable ones every PRENORM_RECHECK
# optimizer steps: a trainer does that for ALL of them in one device reduction + one host read (recheck_trainable_norms; round 6) and,
# when it replays captured steps, drops its graphs if a decision flipped -- a replay runs no Python forward, so the form baked into a
# graph would otherwise never be revisited; without a trainer each LayerNorm re-reads its own ratio when its stamp is that old (never
# inside a capture, where the last decision -- or the exact form -- is used).
# True / False force a form (tests, tools/k5bench.py).
SAVE_PRENORM = None
PRENORM_RATIO = 8.0          # measured (tests/test_gpu_tail.py::test_tail_backward_from_output_error_grows_with_beta_over_gamma, profiles/r05_k5abi_ab.txt): worst element of dgamma 0.019 / 0.027 / 0.075 / 0.27 of the 0.1 bound at max |beta / gamma| = 1 / 4 / 16 / 64
PRENORM_RECHECK = 256
ALIAS_RESIDUAL_GRAD = True   # the plain residual tail's d/dx1 = dout handed on without a copy (False: the kernel writes a copy, as before round 5)


def _frozen_epoch():
    from . import functional as _VF
    return _VF.FROZEN_EPOCH


def _draw_seed() -> int:
    return int(torch.empty((), dtype=torch.int64).random_().item())


def _f32_frozen(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """fp32 contiguous copy of a LayerNorm parameter for the kernels.  A frozen parameter kept in the IO dtype (the decoder's
    LayerNorms of a bf16 run) is converted once and the copy cached on the parameter, keyed on its storage and version --
    not once per forward (two conversion launches per sublayer tail)."""
    if t is None:
        return None
    if t.dtype == torch.float32:
        return t.detach().contiguous()
    if t.requires_grad:
        return t.detach().float().contiguous()
    key = (t.data_ptr(), t._version, t.dtype, _frozen_epoch())
    c = getattr(t, "_vlpet_f32", None)
    if c is None or c[0] != key:
        c = (key, t.detach().float().contiguous())
        t._vlpet_f32 = c
    return c[1]


def needs_prenorm(gamma: torch.Tensor, beta: Optional[torch.Tensor]) -> bool:
    """True: this LayerNorm's K5 backward wants the saved pre-norm rows (form (b) above)."""
    if SAVE_PRENORM is not None:
        return bool(SAVE_PRENORM)
    from . import functional as _VF
    trainable = gamma.requires_grad or (beta is not None and beta.requires_grad)
    bkey = (beta.data_ptr(), beta._version) if beta is not None else None
    key = (gamma.data_ptr(), bkey is None) if trainable else (gamma.data_ptr(), gamma._version, bkey, _VF.FROZEN_EPOCH)
    ent = getattr(gamma, "_vlpet_prenorm", None)
    if ent is not None and ent[0] == key and (not trainable or _VF.WEIGHTS_EPOCH - ent[1] < PRENORM_RECHECK):
        return ent[2]
    if torch.cuda.is_current_stream_capturing():          # no host read inside a capture
        return ent[2] if ent is not None else True
    g = gamma.detach().float().abs()
    b = beta.detach().float().abs() if beta is not None else torch.zeros_like(g)
    ratio = float(torch.where(g > 0, b / g, torch.full_like(g, float("inf"))).max())
    dec = not (ratio <= PRENORM_RATIO)
    gamma._vlpet_prenorm = (key, _VF.WEIGHTS_EPOCH, dec)
    return dec


def recheck_trainable_norms(model: torch.nn.Module) -> bool:
    """All trainable LayerNorms of ``model`` that have a cached form decision, re-examined in ONE device reduction and ONE host read
    (needs_prenorm did a blocking read per LayerNorm, ~40 per recheck, and under graph replay never ran at all -- a replay
    runs no Python forward, so the form baked into a captured graph was never revisited).  Refreshes every cache entry's stamp; returns
    True if a decision FLIPPED (a trainer that replays captured steps must then drop its graphs: train.Trainer._finish_step)."""
    if SAVE_PRENORM is not None:
        return False
    from . import functional as _VF
    ents = []
    for m in model.modules():
        g = getattr(m, "weight", None)
        if not isinstance(g, torch.Tensor) or getattr(g, "_vlpet_prenorm", None) is None or g.dim() != 1:
            continue
        b = getattr(m, "bias", None)
        if not (g.requires_grad or (isinstance(b, torch.Tensor) and b.requires_grad)):
            continue
        ents.append((g, b if isinstance(b, torch.Tensor) else None))
    if not ents:
        return False
    ratios = []
    for g, b in ents:
        ga = g.detach().float().abs()
        ba = b.detach().float().abs() if b is not None else torch.zeros_like(ga)
        ratios.append(torch.where(ga > 0, ba / ga, torch.full_like(ga, float("inf"))).max())
    vals = torch.stack(ratios).tolist()             # the one host read
    flipped = False
    for (g, b), ratio in zip(ents, vals):
        key, _, old = g._vlpet_prenorm
        dec = not (ratio <= PRENORM_RATIO)
        flipped |= dec != old
        g._vlpet_prenorm = (key, _VF.WEIGHTS_EPOCH, dec)
    return flipped


class _TailFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, x1, gamma, beta, eps, p, seed, norm, want_mask, link=None):
        lib = _lib.load()
        _need_cuda(y, x1)
        d = y.shape[-1]
        io = _io_dtype(y)
        if x1.dtype != y.dtype:
            x1 = x1.to(y.dtype)
        yf, xf = _flat(y, d), _flat(x1, d)
        M = yf.shape[0]
        out = torch.empty_like(yf)
        need_bwd = any(t is not None and t.requires_grad for t in (y, x1, gamma, beta))
        f32 = dict(dtype=torch.float32, device=y.device)
        h = mean = rstd = g32 = b32 = None
        if norm:
            g32, b32 = _f32_frozen(gamma), _f32_frozen(beta)
            mean, rstd = torch.empty(M, **f32), torch.empty(M, **f32)
            h = torch.empty_like(yf) if (need_bwd and needs_prenorm(gamma, beta)) else None
        mask = torch.empty(M, d, dtype=torch.uint8, device=y.device) if want_mask else None
        rc = _timed("k5_fwd", M, lambda: lib.vlpet_sublayer_tail_fwd(
            yf.data_ptr(), xf.data_ptr(), _ptr(g32), _ptr(b32), out.data_ptr(), _ptr(h), _ptr(mean), _ptr(rstd),
            _ptr(mask), M, d, float(eps), float(p), seed, int(norm), io, _stream()))
        _lib.check(rc, "vlpet_sublayer_tail_fwd")
        from_out = bool(norm) and need_bwd and h is None
        ctx.save_for_backward(out if from_out else h, mean, rstd, g32, gamma, beta, b32 if from_out else None)
        ctx.cfg = (float(p), seed, int(norm), y.shape, io, from_out)
        ctx.link = link if (link is not None and link.armed) else None     # the op that armed it (K1, or a linear_acc GEMM) sums our dx1 into its own
        out = out.view(y.shape)
        if want_mask:
            ctx.mark_non_differentiable(mask)
            return out, mask.view(y.shape)
        return out

    @staticmethod
    def backward(ctx, dout, *unused):
        lib = _lib.load()
        h, mean, rstd, g32, gamma, beta, b32 = ctx.saved_tensors
        p, seed, norm, shape, io, from_out = ctx.cfg
        d = shape[-1]
        df = _flat(dout, d)
        M = df.shape[0]
        # plain residual tail (T5): d/dx1 IS dout -- handed on as it is (no copy: a third of the pass's traffic), marked shared for a
        # consumer that would accumulate in place; without dropout the whole backward is the identity
        alias = (not norm) and ALIAS_RESIDUAL_GRAD
        if alias and p == 0:
            gx1 = df.view(shape)
            if ctx.link is not None:
                ctx.link.dx1, ctx.link.shared, gx1 = df.view(shape), True, None
                ctx.link = None
            return df.view(shape), gx1, None, None, None, None, None, None, None, None
        dx1 = df if alias else torch.empty_like(df)
        dy = torch.empty_like(df) if p > 0 else None
        train_ln = norm and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        part = None
        if train_ln:
            nb = lib.vlpet_sublayer_tail_partials(M)
            part = torch.empty(nb, 2, d, dtype=torch.float32, device=df.device)
        if from_out:       # h holds the forward's OUTPUT rows: xhat = (out - beta) / gamma inside the kernel
            rc = _timed("k5_bwd", M, lambda: lib.vlpet_sublayer_tail_bwd_out(
                df.data_ptr(), h.data_ptr(), rstd.data_ptr(), g32.data_ptr(), _ptr(b32), dx1.data_ptr(), _ptr(dy), _ptr(part), M, d,
                p, seed, io, _stream()))
        else:
            rc = _timed("k5_bwd", M, lambda: lib.vlpet_sublayer_tail_bwd(
                df.data_ptr(), _ptr(h), _ptr(mean), _ptr(rstd), _ptr(g32), None if alias else dx1.data_ptr(), _ptr(dy), _ptr(part), M, d,
                p, seed, norm, io, _stream()))
        _lib.check(rc, "vlpet_sublayer_tail_bwd")
        dgamma = dbeta = None
        if train_ln:
            # one launch sums the partials, straight into the parameters' slots of the trainer's flat gradient buffer when it
            # offers them (functional._grad_dest: then autograd gets None and runs no accumulate kernel)
            want_g = bool(ctx.needs_input_grad[2])
            want_b = beta is not None and bool(ctx.needs_input_grad[3])
            (tg, sg) = _grad_dest(gamma, (d,)) if want_g else (None, None)
            (tb, sb) = _grad_dest(beta, (d,)) if want_b else (None, None)
            from .functional import reduce_partials
            reduce_partials(part, part.shape[0], d, tg, tb, deferrable=(tg is None or sg is not None) and (tb is None or sb is not None))
            if want_g:
                dgamma = _finish([(tg, sg, gamma)])[0]
            if want_b:
                dbeta = _finish([(tb, sb, beta)])[0]
        dx1 = dx1.view(shape)
        dyv = dy.view(shape) if dy is not None else dx1
        gx1 = dx1
        if ctx.link is not None:        # functional.ResidualLink: K1's backward / the sublayer's first dgrad GEMM returns the sum for x1
            ctx.link.dx1 = dx1
            ctx.link.shared = dy is None or alias   # without dropout dy IS dx1 / the aliased dout is autograd's tensor: a consumer must not accumulate into it in place
            gx1 = None
            ctx.link = None
        return dyv, gx1, dgamma, dbeta, None, None, None, None, None, None


def sublayer_tail(x1: torch.Tensor, y: torch.Tensor, norm: Optional[torch.nn.Module], p: float = 0.0,
                  training: bool = False, seed: Optional[int] = None, return_mask: bool = False, link=None):
    """``norm(x1 + dropout(y, p))``; ``norm`` is an ``nn.LayerNorm`` (BART) or None (T5: plain residual add)."""
    p_eff = float(p) if training else 0.0
    if y.numel() == 0:
        from .functional import _empty_result
        out = _empty_result(y + x1.to(y.dtype), [] if norm is None else [norm.weight, norm.bias])
        return (out, torch.empty(y.shape, dtype=torch.uint8, device=y.device)) if return_mask else out
    if seed is None:
        seed = _draw_seed() if p_eff > 0 else 0
    if norm is None:
        return _TailFn.apply(y, x1, None, None, 0.0, p_eff, seed, 0, return_mask, link)
    return _TailFn.apply(y, x1, norm.weight, norm.bias, norm.eps, p_eff, seed, 1, return_mask, link)


class _RmsNormFn(torch.autograd.Function):
    """T5LayerNorm on the tail kernels (vlpet_rmsnorm_{fwd,bwd}): ``x * rsqrt(mean(x^2) + eps) * weight``, statistics in fp32,
    result in x's dtype; one pass each way instead of the eight / ten elementwise and reduction passes of the eager form."""

    @staticmethod
    def forward(ctx, x, weight, eps, link=None):
        lib = _lib.load()
        _need_cuda(x)
        ctx.link = None
        if link is not None and ctx.needs_input_grad[0]:
            link.armed = True               # the other reader of x (the tail / K1, downstream) parks its d/dx here
            ctx.link = link
        d = x.shape[-1]
        io = _io_dtype(x)
        xf = _flat(x, d)
        M = xf.shape[0]
        out = torch.empty_like(xf)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        g32 = _f32_frozen(weight)
        rc = _timed("rms_fwd", M, lambda: lib.vlpet_rmsnorm_fwd(xf.data_ptr(), g32.data_ptr(), out.data_ptr(), rstd.data_ptr(),
                                                                M, d, float(eps), io, _stream()))
        _lib.check(rc, "vlpet_rmsnorm_fwd")
        ctx.save_for_backward(xf, rstd, g32, weight)
        ctx.cfg = (x.shape, io)
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        xf, rstd, g32, weight = ctx.saved_tensors
        shape, io = ctx.cfg
        M, d = xf.shape
        df = _flat(dout, d)
        if df.dtype != xf.dtype:
            df = df.to(xf.dtype)
        dx = torch.empty_like(xf)
        dx_in = None
        if ctx.link is not None:
            dx_in, ctx.link.dx1 = ctx.link.dx1, None
            ctx.link.shared = False
            ctx.link = None
            if dx_in is not None and (dx_in.dtype != xf.dtype or dx_in.numel() != xf.numel() or not dx_in.is_contiguous()):
                dx_in = dx_in.reshape(xf.shape).to(xf.dtype).contiguous()
        train = bool(ctx.needs_input_grad[1])
        part = torch.empty(lib.vlpet_sublayer_tail_partials(M), 2, d, dtype=torch.float32, device=xf.device) if train else None
        rc = _timed("rms_bwd", M, lambda: lib.vlpet_rmsnorm_bwd(df.data_ptr(), xf.data_ptr(), rstd.data_ptr(), g32.data_ptr(),
                                                                _ptr(dx_in), dx.data_ptr(), _ptr(part), M, d, io, _stream()))
        _lib.check(rc, "vlpet_rmsnorm_bwd")
        dgamma = None
        if train:
            (tg, sg) = _grad_dest(weight, (d,))
            from .functional import reduce_partials
            reduce_partials(part, part.shape[0], d, tg, None, deferrable=sg is not None)      # (queued by a trainer: flush_reduces)
            dgamma = _finish([(tg, sg, weight)])[0]
        return dx.view(shape), dgamma, None, None


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float, link=None) -> torch.Tensor:
    """T5's RMS norm of ``x [..., d]`` (d % 8 == 0, bf16 or fp32 CUDA tensor) on the HIP path.  ``link``
    (functional.ResidualLink): armed here; the op that also reads ``x`` further down parks its gradient and this backward adds it
    inside the kernel."""
    if x.numel() == 0:
        return x * weight.to(x.dtype)
    return _RmsNormFn.apply(x, weight, eps, link)


class FusedNorm:
    """What a fused tail leaves on the sum it returns (``sum._vlpet_norm``): the sum already normalised by ONE T5LayerNorm (identified by
    its weight tensor) and the link through which the sum's later readers hand their gradients to the fused backward."""
    __slots__ = ("weight", "normed", "link")

    def __init__(self, weight, normed, link):
        self.weight, self.normed, self.link = weight, normed, link


class _TailRmsFn(torch.autograd.Function):
    """``sum = x1 + dropout(y)`` and ``normed = T5LayerNorm_next(sum)`` in one pass each way (vlpet_sublayer_tail_rms_fwd /
    vlpet_rmsnorm_tail_bwd): the reference's modeling_t5.py:408 (824) followed by :366 (782) of the next sublayer."""

    @staticmethod
    def forward(ctx, y, x1, weight, eps, p, seed, link, norm_link):
        lib = _lib.load()
        _need_cuda(y, x1)
        ctx.set_materialize_grads(False)
        d = y.shape[-1]
        io = _io_dtype(y)
        if x1.dtype != y.dtype:
            x1 = x1.to(y.dtype)
        yf, xf = _flat(y, d), _flat(x1, d)
        M = yf.shape[0]
        out, normed = torch.empty_like(yf), torch.empty_like(yf)
        rstd = torch.empty(M, dtype=torch.float32, device=y.device)
        g32 = _f32_frozen(weight)
        rc = _timed("k5_fwd", M, lambda: lib.vlpet_sublayer_tail_rms_fwd(
            yf.data_ptr(), xf.data_ptr(), g32.data_ptr(), out.data_ptr(), normed.data_ptr(), rstd.data_ptr(), M, d, float(eps), float(p), seed,
            io, _stream()))
        _lib.check(rc, "vlpet_sublayer_tail_rms_fwd")
        ctx.save_for_backward(out, rstd, g32, weight)
        ctx.cfg = (float(p), seed, y.shape, io)
        ctx.link = link if (link is not None and link.armed) else None       # K1 (or a linear_acc GEMM) takes our d/dx1
        ctx.norm_link = norm_link                                            # the sum's later readers park their gradients here
        if norm_link is not None:
            norm_link.armed = True
        return out.view(y.shape), normed.view(y.shape)

    @staticmethod
    def backward(ctx, d_sum, d_normed):
        lib = _lib.load()
        out, rstd, g32, weight = ctx.saved_tensors
        p, seed, shape, io = ctx.cfg
        M, d = out.shape
        dx_in = None
        if ctx.norm_link is not None:
            dx_in, ctx.norm_link.dx1 = ctx.norm_link.dx1, None
            ctx.norm_link.shared = False
            ctx.norm_link = None
        for extra in (d_sum,):              # a reader of the sum that did not use the link: autograd hands its gradient here
            if extra is not None:
                e = extra.reshape(out.shape)
                dx_in = e if dx_in is None else dx_in.reshape(out.shape) + e
        if dx_in is not None and (dx_in.dtype != out.dtype or dx_in.numel() != out.numel() or not dx_in.is_contiguous()):
            dx_in = dx_in.reshape(out.shape).to(out.dtype).contiguous()
        if d_normed is None:                # the normalised rows were never used: the plain tail's backward on d_sum
            d_normed = torch.zeros_like(out)
        df = _flat(d_normed, d)
        if df.dtype != out.dtype:
            df = df.to(out.dtype)
        dx1 = torch.empty_like(out)
        dy = torch.empty_like(out) if p > 0 else None
        train = bool(ctx.needs_input_grad[2])
        part = torch.empty(lib.vlpet_sublayer_tail_partials(M), 2, d, dtype=torch.float32, device=out.device) if train else None
        rc = _timed("k5_bwd", M, lambda: lib.vlpet_rmsnorm_tail_bwd(
            df.data_ptr(), out.data_ptr(), rstd.data_ptr(), g32.data_ptr(), _ptr(dx_in), dx1.data_ptr(), _ptr(dy), _ptr(part), M, d, p, seed,
            io, _stream()))
        _lib.check(rc, "vlpet_rmsnorm_tail_bwd")
        dgamma = None
        if train:
            (tg, sg) = _grad_dest(weight, (d,))
            from .functional import reduce_partials
            reduce_partials(part, part.shape[0], d, tg, None, deferrable=sg is not None)
            dgamma = _finish([(tg, sg, weight)])[0]
        dx1 = dx1.view(shape)
        dyv = dy.view(shape) if dy is not None else dx1
        gx1 = dx1
        if ctx.link is not None:
            ctx.link.dx1 = dx1
            ctx.link.shared = dy is None
            gx1 = None
            ctx.link = None
        return dyv, gx1, dgamma, None, None, None, None, None


def sublayer_tail_rms(x1: torch.Tensor, y: torch.Tensor, next_norm: torch.nn.Module, p: float = 0.0, training: bool = False,
                      seed: Optional[int] = None, link=None):
    """T5 form of K5 fused with the NEXT sublayer's RMS norm: returns ``sum = x1 + dropout(y, p)``, which carries the already normalised
    rows for ``next_norm`` (``sum._vlpet_norm``: visual.T5LayerNorm.forward returns them instead of launching, host.t5._new_norm_link returns
    the link through which the sum's later readers -- the next tail's add, K1's gate -- hand over their gradients)."""
    from .functional import ResidualLink
    p_eff = float(p) if training else 0.0
    if seed is None:
        seed = _draw_seed() if p_eff > 0 else 0
    eps = getattr(next_norm, "variance_epsilon", None)
    if eps is None:
        eps = next_norm.eps
    need_grad = torch.is_grad_enabled() and (x1.requires_grad or y.requires_grad or next_norm.weight.requires_grad)
    norm_link = ResidualLink() if need_grad else None
    out, normed = _TailRmsFn.apply(y, x1, next_norm.weight, eps, p_eff, seed, link, norm_link)
    out._vlpet_norm = FusedNorm(next_norm.weight, normed, norm_link)
    return out
