"""Host glue of the LM-head loss (csrc/celoss.hip): ``CrossEntropyLoss(ignore_index=-100, reduction='none')`` on the
LM-head logits as one read forward and one read + one write backward.

Reference op chain replaced: src/modeling_bart.py:1574-1586 (``lm_head(h) + final_logits_bias`` then the loss on
``lm_logits.view(-1, V)``) and src/modeling_t5.py:680-694.  The head GEMM stays a library GEMM (frozen, tied to the token
table); its weight is used through a cached copy in the activation dtype whose row count is padded to a multiple of 8,
so that every logits row starts on a 16-byte boundary (V = 50,465 for BART: odd).  Padding columns never enter the
softmax and get zero gradient.  No CPU fallback."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib
from .functional import _io_dtype, _need_cuda, _stream, _timed


def _frozen_epoch():
    from . import functional as _VF
    return _VF.FROZEN_EPOCH


class _CeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, V):
        lib = _lib.load()
        _need_cuda(logits, labels)
        N, ld = logits.shape
        io = _io_dtype(logits)
        lab = labels.contiguous()
        if lab.dtype != torch.int64:
            lab = lab.long()
        loss = torch.empty(N, dtype=torch.float32, device=logits.device)
        lse = torch.empty(N, dtype=torch.float32, device=logits.device)
        bad = _bad_counter(logits.device)
        rc = _timed("ce_fwd", N, lambda: lib.vlpet_ce_loss_fwd_checked(logits.data_ptr(), lab.data_ptr(), loss.data_ptr(), lse.data_ptr(),
                                                                       bad.data_ptr(), N, V, ld, io, _stream()))
        _lib.check(rc, "vlpet_ce_loss_fwd_checked")
        ctx.save_for_backward(logits, lab, lse)
        ctx.cfg = (V, io)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        lib = _lib.load()
        logits, lab, lse = ctx.saved_tensors
        V, io = ctx.cfg
        N, ld = logits.shape
        g = dloss.contiguous().float()
        dlogits = torch.empty_like(logits)
        rc = _timed("ce_bwd", N, lambda: lib.vlpet_ce_loss_bwd(logits.data_ptr(), lab.data_ptr(), lse.data_ptr(), g.data_ptr(),
                                                               dlogits.data_ptr(), N, V, ld, io, _stream()))
        _lib.check(rc, "vlpet_ce_loss_bwd")
        return dlogits, None, None


def cross_entropy_rows(logits: torch.Tensor, labels: torch.Tensor, V: Optional[int] = None) -> torch.Tensor:
    """Per-token loss [N] (fp32) of logits [N, ld] (columns [0, V) valid, ld % 8 == 0, contiguous rows)."""
    if logits.dim() != 2 or not logits.is_contiguous():
        raise RuntimeError("vl-pet_amd: cross_entropy_rows needs contiguous [N, ld] logits")
    V = logits.shape[1] if V is None else int(V)
    if logits.shape[1] % 8 != 0:
        raise RuntimeError("vl-pet_amd: logits row stride must be a multiple of 8 (use lm_head_loss, which pads the head)")
    labels = labels.reshape(-1)
    if logits.shape[0] == 0:                         # an empty batch has an empty loss (F.cross_entropy, reduction 'none')
        return logits.new_zeros(0, dtype=torch.float32) + 0.0 * logits.float().sum()
    _check_labels_once(labels, V)
    return _CeFn.apply(logits, labels, V)


# The kernel treats a label outside [0, V) like ignore_index (loss 0, no gradient); F.cross_entropy raises a device assert for
# it.  A tokenizer / vocabulary mismatch would otherwise train silently on fewer tokens, so the first batch of every (V, label
# dtype) is validated on the host (one synchronisation per process and vocabulary); CHECK_LABELS = "always" checks every call.
CHECK_LABELS = "once"
_CHECKED = set()
# ... and every later batch is tallied on the device: the forward kernel counts the labels outside [0, V) other than -100 into a
# 32-bit word per device (vlpet_ce_loss_fwd_checked; no synchronisation, works inside a captured graph -- the word is allocated at
# the first, eager, call).  bad_label_count() reads it (one 4-byte copy); train.Trainer checks it every LABEL_CHECK_EVERY steps.
_BAD = {}


def _bad_counter(device: torch.device) -> torch.Tensor:
    key = (device.type, device.index)
    t = _BAD.get(key)
    if t is None:
        t = _BAD[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def bad_label_count(device=None) -> int:
    """Labels outside the vocabulary (and not ignore_index) seen by the loss so far on ``device`` (all devices if None)."""
    ts = [t for k, t in _BAD.items() if device is None or k == (torch.device(device).type, torch.device(device).index)]
    return int(sum(int(t.item()) for t in ts))


def _check_labels_once(labels: torch.Tensor, V: int):
    if CHECK_LABELS == "never":
        return
    key = (V, labels.dtype, labels.device.type)
    if CHECK_LABELS == "once" and key in _CHECKED:
        return
    bad = (labels >= V) | ((labels < 0) & (labels != -100))
    if bool(bad.any()):
        raise IndexError(f"vl-pet_amd: label {int(labels[bad][0])} outside [0, {V}) (and not ignore_index -100)")
    _CHECKED.add(key)


def _padded_head(weight: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[ceil8(V), d] copy of the head weight in the activation dtype, zero rows appended.  Frozen weights are cached (keyed on
    the tensor's version counter, so load_state_dict / in-place edits refresh it); a trainable head is rebuilt
    differentiably on every call."""
    V, d = weight.shape
    Vp = (V + 7) // 8 * 8
    if weight.requires_grad:
        w = weight.to(dtype)
        return w if Vp == V else torch.cat([w, w.new_zeros(Vp - V, d)], 0)
    key = (weight.data_ptr(), weight._version, dtype, tuple(weight.shape), _frozen_epoch())
    hit = getattr(weight, "_vlpet_padded_head", None)              # lives and dies with the parameter object
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        wp = torch.zeros(Vp, d, dtype=dtype, device=weight.device)
        wp[:V].copy_(weight)
    weight._vlpet_padded_head = (key, wp)
    return wp


def lm_head_loss(h: torch.Tensor, weight: torch.Tensor, labels: torch.Tensor,
                 bias: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """``loss_fct(lm_head(h) + bias, labels)`` with reduction 'none': returns (loss shaped like labels, fp32; logits
    [..., V] in h's dtype).  ``bias`` is BART's ``final_logits_bias`` buffer: all zeros unless a checkpoint says otherwise,
    in which case it is added with a torch pass before the loss."""
    V = weight.shape[0]
    wp = _padded_head(weight, h.dtype)
    logits = F.linear(h, wp)                                       # [..., Vp]
    if bias is not None and bool(bias.any()):
        logits = logits + F.pad(bias.to(h.dtype).reshape(-1), (0, wp.shape[0] - V))
    flat = logits.reshape(-1, wp.shape[0])
    loss = cross_entropy_rows(flat, labels, V)
    return loss.view(labels.shape), logits[..., :V]
