"""K4 host glue: visual-feature projection Linear(feat_dim -> d_model) + LayerNorm (+ residual add) on the
HIP kernel (csrc/visproj.hip), weight gradient on the column-parallel MFMA kernel (csrc/wgrad.hip).

The LayerNorm backward (an elementwise pass + two row reductions over [M, d_model]) is expressed with
torch ops on the saved normalised activations; the two heavy contractions (K = feat_dim forward,
K = M weight gradient) are the hand-written kernels.  No CPU fallback."""
from __future__ import annotations

import torch

from . import _lib
from .functional import _finish, _flat, _grad_dest, _grad_like, _io_dtype, _need_cuda, _param_dtype, _ptr, _stream, _timed


# K4 forward, bf16:
#   K4_FORM = "gemm" (default, round 5): the hand-written tiled GEMM of csrc/visproj_gemm.hip -- [rows x 256 features] tiles, the three
#       column tiles of a row block exchange the LayerNorm / RMS statistics through L2 and finish the norm + residual add themselves
#       (one launch, d_model a multiple of 256: both backbones' 768); saves xhat + rstd for the backward like the fused kernel.
#   "library" (the round-4 default, kept for same-box A/Bs): hipBLASLt GEMM + ONE pass of the K5 kernel (vlpet_norm_residual_fwd);
#       LayerNorm form only.
#   "fused" (round 2): csrc/visproj.hip, a workgroup owns 128 rows x all features (what fp32 IO and other widths always run).
K4_FORM = "gemm"
GEMM_THEN_NORM = True          # (honoured when K4_FORM == "library")


# Workspace of the tiled-GEMM form (exchange area + per-tile statistics + give-up flags): zeroed ONCE per device; every call leaves the
# exchange area zeroed (each consumer clears the granules it has read; a call in which a workgroup gave up on a partner repairs its
# tiles and re-zeroes the area itself, csrc/visproj_gemm.hip), so no memset runs between launches.  One area per device: K4 launches
# of one process are issued on one stream at a time.
_GEMM_WS = {}
# The area is sized ONCE, for GEMM_WS_ROWS rows, and never reallocated: a captured train step bakes its address into the graph, and since
# round 6 the size depends on the row count (the per-tile statistics) -- an area that grew with the second task shape would leave the first
# shape's graph writing into freed memory (found as a memory fault of the replayed LoRA bench).  More rows than that take the library form.
GEMM_WS_ROWS = 1 << 19


def _gemm_workspace(device: torch.device, feat_dim: int, d_out: int) -> torch.Tensor:
    key = (device.type, device.index, int(d_out))
    ws = _GEMM_WS.get(key)
    if ws is None:
        nbytes = _lib.load().vlpet_visproj_gemm_workspace_bytes(GEMM_WS_ROWS, 64, int(d_out))      # (the size does not depend on feat_dim)
        with torch.cuda.device(device):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("vl-pet_amd: the K4 exchange area must exist before a step is captured (run one eager step first)")
            ws = _GEMM_WS[key] = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    return ws


def gemm_exchange_status(device=None) -> int:
    """Tiles of K4 launches on ``device`` (any, if None) whose workgroup gave up waiting for a partner workgroup's LayerNorm statistics
    since the last call (the GPU was not this process's alone for about a second).  INFORMATIONAL: the same call re-normalised those tiles
    from the complete statistics before anything downstream could read them (csrc/visproj_gemm.hip, repair kernel), so no step was
    computed on wrong rows; a nonzero count says the K4 launch cost a second instead of 80 us.  One 16-byte device read; clears the
    sticky status word (train.Trainer.check_labels calls it every LABEL_CHECK_EVERY steps and warns)."""
    tiles = 0
    for key, ws in _GEMM_WS.items():
        if device is not None and key[:2] != (torch.device(device).type, torch.device(device).index):
            continue
        hdr = ws[:16].view(torch.int32).tolist()        # status, tiles given up so far, ... at the last repair, repair workgroups done
        if hdr[0]:
            seen = getattr(gemm_exchange_status, "_seen", {})
            tiles += max(1, hdr[1] - seen.get(key, 0))
            seen[key] = hdr[1]
            gemm_exchange_status._seen = seen
            ws[:4].zero_()
    return tiles


class VisProjPackCache:
    def __init__(self):
        self._key = None
        self._val = None
        self._ckey = None
        self._cval = None

    def get_cast(self, w: torch.Tensor, b: torch.Tensor, dtype: torch.dtype):
        """the weight and bias in the IO dtype for the library-GEMM form, refreshed when the parameters change"""
        from . import functional as _VF
        key = (dtype, _VF.WEIGHTS_EPOCH, w.data_ptr(), w._version, b.data_ptr(), b._version)
        if key != self._ckey:
            old = self._cval
            if (old is not None and old[0].dtype == dtype and old[0].shape == w.shape and old[0].device == w.device
                    and old[1].shape == b.shape):
                old[0].copy_(w.detach()); old[1].copy_(b.detach())      # in place: a captured graph may hold these addresses
            else:
                self._cval = (w.detach().to(dtype).contiguous(), b.detach().to(dtype).contiguous())
            self._ckey = key
        return self._cval

    def get(self, w: torch.Tensor, b: torch.Tensor, io_dtype: int) -> torch.Tensor:
        from . import functional as _VF
        key = (io_dtype, _VF.WEIGHTS_EPOCH, w.data_ptr(), w._version, b.data_ptr(), b._version)
        if key != self._key:
            lib = _lib.load()
            d_out, F = w.shape
            nb = lib.vlpet_visproj_packed_bytes(d_out, F, io_dtype)
            buf = self._val if (self._val is not None and self._val.numel() == nb and self._val.device == w.device) \
                else torch.empty(nb, dtype=torch.uint8, device=w.device)      # (refreshed in place, as get_cast)
            wc, bc = w.detach().contiguous(), b.detach().contiguous()
            rc = lib.vlpet_visproj_pack(wc.data_ptr(), bc.data_ptr(), d_out, F, _param_dtype(wc), io_dtype,
                                        buf.data_ptr(), _stream())
            _lib.check(rc, "vlpet_visproj_pack")
            self._key, self._val = key, buf
        return self._val


class _VisProjFn(torch.autograd.Function):
    last_status = None

    @staticmethod
    def forward(ctx, feats, R, w, b, gamma, beta, packed, eps, rms, cast=None, gemm=False):
        lib = _lib.load()
        _need_cuda(feats, w)
        F = feats.shape[-1]
        d_out = w.shape[0]
        io = _io_dtype(feats)
        ff = _flat(feats, F)
        M = ff.shape[0]
        Rf = None
        if R is not None:
            Rf = _flat(R.to(feats.dtype), d_out)
        out = torch.empty(M, d_out, dtype=feats.dtype, device=feats.device)
        if gemm:                        # tiled GEMM + statistics exchange (csrc/visproj_gemm.hip)
            from .tail import _f32_frozen
            wc, _ = cast
            g32, be32, b32 = _f32_frozen(gamma), _f32_frozen(beta), _f32_frozen(b)
            xhat = torch.empty_like(out)
            rstd = torch.empty(M, dtype=torch.float32, device=feats.device)
            ws = _gemm_workspace(feats.device, F, d_out)
            nws = ws.numel()
            rc = _timed("k4_fwd", M, lambda: lib.vlpet_visproj_fwd_gemm(
                ff.data_ptr(), wc.data_ptr(), _ptr(b32), g32.data_ptr(), _ptr(be32), _ptr(Rf), out.data_ptr(), xhat.data_ptr(),
                rstd.data_ptr(), None, ws.data_ptr(), nws, M, F, d_out, float(eps), int(bool(rms)), io, _stream()))
            _lib.check(rc, "vlpet_visproj_fwd_gemm")
            _VisProjFn.last_status = ws[:4]         # (tests: view of the launch's status word)
            ctx.composed = False
            ctx.save_for_backward(ff, xhat, rstd, w, b, gamma, beta if beta is not None else gamma, None)
            ctx.cfg = (bool(rms), beta is not None, feats.shape[:-1], R is not None, R.dtype if R is not None else None)
            return out.view(*feats.shape[:-1], d_out)
        if cast is not None:            # library GEMM, then LayerNorm + residual in one pass of the K5 kernel
            from .tail import _f32_frozen
            wc, bc = cast
            g32, b32 = _f32_frozen(gamma), _f32_frozen(beta)
            mean = torch.empty(M, dtype=torch.float32, device=feats.device)
            rstd = torch.empty(M, dtype=torch.float32, device=feats.device)
            if Rf is None:
                Rf = torch.zeros(M, d_out, dtype=feats.dtype, device=feats.device)
            box = {}

            def run():
                box["pre"] = torch.addmm(bc, ff, wc.t())
                return lib.vlpet_norm_residual_fwd(box["pre"].data_ptr(), Rf.data_ptr(), g32.data_ptr(), _ptr(b32), out.data_ptr(),
                                                   mean.data_ptr(), rstd.data_ptr(), M, d_out, float(eps), io, _stream())
            _lib.check(_timed("k4_fwd", M, run), "vlpet_norm_residual_fwd")
            ctx.save_for_backward(ff, box["pre"], rstd, w, b, gamma, beta if beta is not None else gamma, mean)
            ctx.cfg = (False, beta is not None, feats.shape[:-1], R is not None, R.dtype if R is not None else None)
            ctx.composed = True
            return out.view(*feats.shape[:-1], d_out)
        ctx.composed = False
        xhat = torch.empty_like(out)
        rstd = torch.empty(M, dtype=torch.float32, device=feats.device)
        from .tail import _f32_frozen
        g32, b32 = _f32_frozen(gamma), _f32_frozen(beta)
        rc = _timed("k4_fwd", M, lambda: lib.vlpet_visproj_fwd(
            ff.data_ptr(), packed.data_ptr(), g32.data_ptr(), _ptr(b32), _ptr(Rf), out.data_ptr(), xhat.data_ptr(),
            rstd.data_ptr(), M, F, d_out, float(eps), int(bool(rms)), io, _stream()))
        _lib.check(rc, "vlpet_visproj_fwd")
        ctx.save_for_backward(ff, xhat, rstd, w, b, gamma, beta if beta is not None else gamma, None)
        ctx.cfg = (bool(rms), beta is not None, feats.shape[:-1], R is not None, R.dtype if R is not None else None)
        return out.view(*feats.shape[:-1], d_out)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        ff, xhat, rstd, w, b, gamma, beta, mean = ctx.saved_tensors          # (composed form: `xhat` holds the pre-norm rows)
        rms, has_beta, lead, has_r, r_dtype = ctx.cfg
        M, F = ff.shape
        d_out = w.shape[0]
        io = _io_dtype(ff)
        s_g = s_be = None
        if rms:         # T5's RMS norm: library ops (one call per step on [M_v, d])
            dyf = _flat(dy, d_out).float()
            xh = xhat.float()
            g = dyf * gamma.float()
            c2 = (g * xh).mean(dim=1, keepdim=True)
            dpre = (g - xh * c2) * rstd[:, None]
            dgamma = (dyf * xh).sum(0)
            dbeta = None
            dpre_io = dpre.to(ff.dtype).contiguous()
        else:           # LayerNorm: K5's backward kernel on the saved xhat, its partial sums reduced into the gradient slots
            from .tail import _f32_frozen
            dyc = _flat(dy, d_out)
            if dyc.dtype != ff.dtype:
                dyc = dyc.to(ff.dtype)
            dpre_io = torch.empty_like(xhat)
            train_ln = ctx.needs_input_grad[4] or (has_beta and ctx.needs_input_grad[5])
            part = torch.empty(lib.vlpet_sublayer_tail_partials(M), 2, d_out, dtype=torch.float32, device=ff.device) if train_ln else None
            if ctx.composed:
                rc = _timed("k4_ln_bwd", M, lambda: lib.vlpet_sublayer_tail_bwd(
                    dyc.data_ptr(), xhat.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _f32_frozen(gamma).data_ptr(), dpre_io.data_ptr(),
                    None, _ptr(part), M, d_out, 0.0, 0, 1, io, _stream()))
            else:
                rc = _timed("k4_ln_bwd", M, lambda: lib.vlpet_layernorm_bwd_xhat(
                    dyc.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), _f32_frozen(gamma).data_ptr(), dpre_io.data_ptr(), _ptr(part),
                    M, d_out, io, _stream()))
            _lib.check(rc, "vlpet_layernorm_bwd_xhat")
            dgamma = dbeta = None
            if train_ln:
                (dgamma, s_g) = _grad_dest(gamma, (d_out,))
                (dbeta, s_be) = _grad_dest(beta, (d_out,)) if has_beta else (None, None)
                from .functional import reduce_partials
                reduce_partials(part, part.shape[0], d_out, dgamma, dbeta, deferrable=s_g is not None and (dbeta is None or s_be is not None))
        (dw, s_w), (db, s_b) = _grad_dest(w, (d_out, F)), _grad_dest(b, (d_out,))
        nws = lib.vlpet_visproj_wgrad_workspace_bytes_io(M, F, d_out, io)
        ws = torch.empty(nws, dtype=torch.uint8, device=ff.device)
        rc = _timed("k4_wgrad", M, lambda: lib.vlpet_visproj_wgrad(
            dpre_io.data_ptr(), ff.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nws, M, F, d_out, io,
            _stream()))
        _lib.check(rc, "vlpet_visproj_wgrad")
        dR = dy.to(r_dtype) if has_r else None
        gw, gb = _finish([(dw, s_w, w), (db, s_b, b)])
        # the CLIP features of K4 carry no gradient; the low-rank projector's bottleneck activations do (library GEMM)
        dfeats = (dpre_io @ w.detach().to(dpre_io.dtype)).view(*lead, F) if ctx.needs_input_grad[0] else None
        if rms or dgamma is None:
            gg = _grad_like(dgamma, gamma) if dgamma is not None else None
            gbe = None
        else:
            gg = _finish([(dgamma, s_g, gamma)])[0]
            gbe = _finish([(dbeta, s_be, beta)])[0] if has_beta else None
        return (dfeats, dR, gw, gb, gg, gbe, None, None, None, None, None)


def visproj(feats, R, linear: torch.nn.Linear, norm: torch.nn.Module, cache: VisProjPackCache, rms: bool):
    """LN(Linear(feats)) (+ R) through the HIP path; `norm` is nn.LayerNorm or the T5 RMS norm."""
    if feats.numel() == 0:
        from .functional import _empty_result
        base = feats.new_zeros(*feats.shape[:-1], linear.weight.shape[0]) if R is None else R.to(feats.dtype)
        return _empty_result(base, [linear.weight, linear.bias, norm.weight, getattr(norm, "bias", None)])
    io = _io_dtype(feats)
    eps = getattr(norm, "eps", None)
    if eps is None:
        eps = norm.variance_epsilon
    beta = getattr(norm, "bias", None)
    if K4_FORM == "gemm" and feats.dtype == torch.bfloat16 and feats.is_cuda:
        lib = _lib.load()
        M = feats.numel() // feats.shape[-1]
        if M <= GEMM_WS_ROWS and lib.vlpet_visproj_gemm_workspace_bytes(M, feats.shape[-1], linear.weight.shape[0]) > 0:
            cast = cache.get_cast(linear.weight, linear.bias, feats.dtype)
            return _VisProjFn.apply(feats, R, linear.weight, linear.bias, norm.weight, beta, None, eps, rms, cast, True)
    if K4_FORM != "fused" and GEMM_THEN_NORM and not rms and feats.dtype == torch.bfloat16:
        cast = cache.get_cast(linear.weight, linear.bias, feats.dtype)
        return _VisProjFn.apply(feats, R, linear.weight, linear.bias, norm.weight, beta, None, eps, rms, cast)
    packed = cache.get(linear.weight, linear.bias, io)
    return _VisProjFn.apply(feats, R, linear.weight, linear.bias, norm.weight, beta, packed, eps, rms)


# ------------------------------------------------------------------------------------------------ position / order branch
# R = norm_p(Linear(5 -> d)([box, area])) + img_order_embedding[.] + obj_order_embedding[V - 1 - .] as ONE launch each way
# (csrc/vispos.hip, round 6; src/modeling_bart.py:129-141,162-183).  A/B switch (tools/ab_switches.py: VLPET_NO_POS_KERNEL=1).
FUSE_POS_BRANCH = True


def _f32c(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    return (t if t.dtype == torch.float32 else t.float()).contiguous()


def _ids_arg(ids, B: int, N: int):
    """(tensor kept alive, pointer, batch stride) of an id tensor shaped [N], [1, N] or [B, N]; None = the reference's default ids"""
    if ids is None:
        return None, None, 0
    t = ids.detach()
    if t.dim() == 1:
        t = t.unsqueeze(0)
    if t.dtype != torch.long:
        t = t.long()
    t = t.contiguous()
    return t, t.data_ptr(), (0 if t.shape[0] == 1 else N)


class _VisPosFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, w, b, gamma, beta, img_table, obj_table, img_ids, obj_ids, eps, rms, out_dtype):
        lib = _lib.load()
        B, N, _ = pos.shape
        d = w.shape[0]
        M = B * N
        pf = _f32c(pos).view(M, 4)
        wf, bf, gf = _f32c(w), _f32c(b), _f32c(gamma)
        bef = _f32c(beta) if beta is not None else None
        io = _lib.VLPET_F32 if out_dtype == torch.float32 else _lib.VLPET_BF16
        out = torch.empty(B, N, d, dtype=out_dtype, device=pos.device)
        it = ot = None
        n_img = obj_rows = 0
        if img_table is not None:
            it, ot = img_table.detach().contiguous(), obj_table.detach().contiguous()
            n_img, obj_rows = it.shape[0], ot.shape[0]
        ik, ip, ibs = _ids_arg(img_ids, B, N)
        ok, op, obs = _ids_arg(obj_ids, B, N)
        rc = _timed("k4_pos_fwd", M, lambda: lib.vlpet_vispos_fwd(
            pf.data_ptr(), wf.data_ptr(), bf.data_ptr(), gf.data_ptr(), _ptr(bef),
            _ptr(it), _param_dtype(it) if it is not None else 0, n_img, ip, ibs,
            _ptr(ot), _param_dtype(ot) if ot is not None else 0, obj_rows, op, obs,
            out.data_ptr(), M, N, d, float(eps), int(bool(rms)), io, _stream()))
        _lib.check(rc, "vlpet_vispos_fwd")
        ctx.save_for_backward(pf, w, b, gamma, beta if beta is not None else gamma, img_table if img_table is not None else gamma, ik if ik is not None else gamma)
        ctx.cfg = (B, N, d, float(eps), bool(rms), beta is not None, img_table is not None, ik is not None, ibs, io)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        pf, w, b, gamma, beta, img_table, ik = ctx.saved_tensors
        B, N, d, eps, rms, has_beta, has_img, has_ids, ibs, io = ctx.cfg
        M = B * N
        dy = dout.contiguous()
        want = torch.float32 if io == _lib.VLPET_F32 else torch.bfloat16
        if dy.dtype != want:
            dy = dy.to(want)
        n_img = img_table.shape[0] if (has_img and ctx.needs_input_grad[5]) else 0
        (dw, s_w), (db, s_b), (dg, s_g) = _grad_dest(w, (d, 5)), _grad_dest(b, (d,)), _grad_dest(gamma, (d,))
        (dbe, s_be) = _grad_dest(beta, (d,)) if has_beta else (None, None)
        (di, s_i) = _grad_dest(img_table, (n_img, d)) if n_img else (None, None)
        nws = lib.vlpet_vispos_bwd_workspace_bytes(M, d, n_img)
        ws = torch.empty(nws, dtype=torch.uint8, device=dy.device)
        wf, bf, gf = _f32c(w), _f32c(b), _f32c(gamma)
        rc = _timed("k4_pos_bwd", M, lambda: lib.vlpet_vispos_bwd(
            dy.data_ptr(), pf.data_ptr(), wf.data_ptr(), bf.data_ptr(), gf.data_ptr(),
            n_img, ik.data_ptr() if has_ids else None, ibs,
            dw.data_ptr(), db.data_ptr(), dg.data_ptr(), _ptr(dbe), _ptr(di),
            ws.data_ptr(), nws, M, N, d, eps, int(rms), io, _stream()))
        _lib.check(rc, "vlpet_vispos_bwd")
        gw, gb, gg = _finish([(dw, s_w, w), (db, s_b, b), (dg, s_g, gamma)])
        gbe = _finish([(dbe, s_be, beta)])[0] if has_beta else None
        gi = _finish([(di, s_i, img_table)])[0] if n_img else None
        return (None, gw, gb, gg, gbe, gi, None, None, None, None, None, None)


def position_terms(pos, linear: torch.nn.Linear, norm: torch.nn.Module, img_embedding, obj_embedding, img_ids, obj_ids,
                   rms: bool, out_dtype: torch.dtype):
    """R [B, N, d] in ``out_dtype`` through csrc/vispos.hip, or None where the kernel does not apply (the caller keeps its torch ops):
    d not a multiple of 256 / above 1024, more than 4 image-order rows, a trainable object-order table, id tensors of another shape."""
    if not FUSE_POS_BRANCH or not pos.is_cuda or out_dtype not in (torch.float32, torch.bfloat16):
        return None
    B, N, _ = pos.shape
    d = linear.weight.shape[0]
    if B * N == 0 or linear.weight.shape[1] != 5 or linear.bias is None:
        return None
    n_img = img_embedding.num_embeddings if img_embedding is not None else 0
    if not _lib.load().vlpet_vispos_applies(d, n_img):
        return None
    tabs = (None, None)
    if img_embedding is not None:
        if obj_embedding is None or (obj_embedding.weight.requires_grad and torch.is_grad_enabled()):
            return None
        for t in (img_embedding.weight, obj_embedding.weight):
            if t.dtype not in (torch.float32, torch.bfloat16) or t.shape[1] != d:
                return None
        for ids in (img_ids, obj_ids):
            if ids is not None and not (ids.dim() in (1, 2) and ids.shape[-1] == N and (ids.dim() == 1 or ids.shape[0] in (1, B))):
                return None
        tabs = (img_embedding.weight, obj_embedding.weight)
    beta = None if rms else getattr(norm, "bias", None)
    if not rms and beta is None:
        return None
    eps = norm.variance_epsilon if rms else norm.eps
    return _VisPosFn.apply(pos, linear.weight, linear.bias, norm.weight, beta, tabs[0], tabs[1], img_ids, obj_ids, eps, rms, out_dtype)
