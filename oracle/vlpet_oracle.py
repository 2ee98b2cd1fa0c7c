"""CPU oracle for the VL-PET PET hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32 or fp64) restatement of the reference
algorithm for the hot path named in BASELINE.json:north_star.  It is the checker
the HIP kernels are compared against.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product package
(``vl-pet_amd``) never does, and fails loudly when its HIP library is missing.

Pinning: the reference holds no golden vectors of its own, so
the oracle is pinned against outputs of the reference's own classes imported in the
build container (``tests/golden/make_goldens.py`` -> ``tests/golden/*.npz``) and
``tests/test_oracle_golden.py`` replays every fixture through this file.

Every function cites the reference file:line (relative to /root/reference/src)
whose op order it follows.  Backward passes are obtained with torch.autograd on
the restated forward, i.e. exactly the mechanism the reference itself uses.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

# gate modes of the encoder granularity-controlled adapter
GATE_NONE = 0      # adapter only (VL-Adapter style, no gate)
GATE_LARGE = 1     # VL-PET-large : low-rank sigmoid gate on x1, [M,d]
GATE_SMALL = 2     # VL-PET-small : Linear(2d->1) on cat(x1,h), sigmoid, mean over S
GATE_MIDDLE_X = 3  # VL-PET-middleX: Linear(d->1) on x1+h, sigmoid, per token
GATE_MIDDLE_Y = 4  # VL-PET-middleY: h + h*z, z in R^d


def gelu_new(x: torch.Tensor) -> torch.Tensor:
    """HF ``NewGELUActivation`` (tanh form), selected by ``get_activation('gelu_new')``
    at my_transformers/modeling_bart.py:1000,1044 and adapters/config.py:10."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


# --------------------------------------------------------------------------- K1
def encoder_adapter_gate(
    x1: torch.Tensor,                  # sublayer input  ("residual")  [B,S,d]
    x2: torch.Tensor,                  # frozen attention / FFN output [B,S,d]
    down_w: Sequence[torch.Tensor],    # N_h tensors [r/N_h, d]  (attn_adapter_multihead_down.{i}.weight)
    down_b: Sequence[torch.Tensor],    # N_h tensors [r/N_h]
    up_w: torch.Tensor,                # [d, r]   attn_adapter_multihead_up.weight
    up_b: torch.Tensor,                # [d]
    gate: Optional[Dict[str, torch.Tensor]] = None,
    gate_mode: int = GATE_LARGE,
    gating_add: bool = False,          # config.use_encoder_adapter_gating_add
    delta_scale: float = 1.0,          # T5 encoder_adapter_scaling_factor
    x2_scale: float = 1.0,             # T5 encoder_x2_scaling_factor
    gate_scale: float = 1.0,           # encoder_gating_scaling_factor
) -> torch.Tensor:
    """Encoder granularity-controlled adapter, the value that is fed to
    ``residual + dropout(.)``.

    BART: my_transformers/modeling_bart.py:1147-1155 (multi-head down, cat, gelu_new,
    up, +x2), :1195-1209 (large gate), :1210-1231 (small / middleX / middleY gates),
    :1256-1257 (gating scale).  FFN sublayer :1270-1278,1317-1347 is identical.
    T5:   my_transformers/modeling_t5.py:366-379,385-406 (FF) and :782-822 (self-attn),
    which add the delta / x2 scaling factors and only have the multiplicative gates.
    """
    # h = x2 + up(gelu_new(cat_i down_i(x2)))
    heads = [F.linear(x2, w, b) for w, b in zip(down_w, down_b)]
    z = torch.cat(heads, dim=-1)
    z = gelu_new(z)
    delta = F.linear(z, up_w, up_b)
    if delta_scale != 1.0:
        delta = delta * delta_scale
    h = x2
    if x2_scale != 1.0:
        h = h * x2_scale
    h = h + delta

    if gate_mode == GATE_NONE:
        y = h
    elif gate_mode == GATE_LARGE:
        g = F.linear(x1, gate["down_w"], gate["down_b"])
        g = gelu_new(g)
        g = F.linear(g, gate["up_w"], gate["up_b"])
        g = torch.sigmoid(g)
        y = h + g if gating_add else h * g
    elif gate_mode == GATE_SMALL:
        gi = torch.cat([x1, h], dim=2)
        g = torch.sigmoid(F.linear(gi, gate["w"], gate["b"]))       # [B,S,1]
        g = torch.mean(g, dim=1).unsqueeze(-1)                      # [B,1,1]
        y = h + g if gating_add else h * g
    elif gate_mode == GATE_MIDDLE_X:
        g = torch.sigmoid(F.linear(x1 + h, gate["w"], gate["b"]))   # [B,S,1]
        y = h + g if gating_add else h * g
    elif gate_mode == GATE_MIDDLE_Y:
        if gating_add:
            y = h + torch.ones_like(h) + gate["z"]
        else:
            y = h + h * gate["z"]
    else:
        raise ValueError(gate_mode)
    if gate_scale != 1.0:
        y = y * gate_scale
    return y


# --------------------------------------------------------------------------- K2
def parallel_adapter(
    x: torch.Tensor, y: Optional[torch.Tensor],
    down_w: torch.Tensor, down_b: torch.Tensor, up_w: torch.Tensor, up_b: torch.Tensor,
    scaling: Optional[float] = None, parallel: bool = True,
) -> torch.Tensor:
    """``AdapterController.forward(inputs, task, y)`` with a plain ``Adapter``.

    adapters/adapter_modeling.py:55-61 (down, gelu_new, up) and
    adapters/adapter_controller.py:149-162 (scaling, ``+ y`` when parallel else ``+ inputs``),
    as called for the decoder cross-attention value at
    my_transformers/modeling_bart.py:427-430 / modeling_t5.py:600-603.
    """
    z = gelu_new(F.linear(x, down_w, down_b))
    out = F.linear(z, up_w, up_b)
    if scaling is not None:
        out = scaling * out
    return out + (y if parallel else x)


# --------------------------------------------------------------------------- K3
def lora_linear(
    x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
    lora_a: torch.Tensor, lora_b: torch.Tensor, scaling: float,
    keep_mask: Optional[torch.Tensor] = None, p: float = 0.0,
) -> torch.Tensor:
    """``LoRALinearController.forward`` (lora/controller.py:56-70).

    ``keep_mask`` (same shape as x, 1 = keep) restates ``nn.Dropout(p)`` on the LoRA branch
    (train mode, :66) with an explicit mask: dropout(x) = x * mask / (1-p)."""
    result = F.linear(x, weight, bias)
    xd = x
    if keep_mask is not None:
        xd = x * keep_mask.to(x.dtype) / (1.0 - p)
    result = result + (xd @ lora_a.T @ lora_b.T) * scaling
    return result


# --------------------------------------------------------------------------- K4
def visual_embedding(
    feats: torch.Tensor, pos: torch.Tensor,
    feat_w: torch.Tensor, feat_b: torch.Tensor,
    feat_ln_w: Optional[torch.Tensor], feat_ln_b: Optional[torch.Tensor],
    pos_w: torch.Tensor, pos_b: torch.Tensor,
    pos_ln_w: Optional[torch.Tensor], pos_ln_b: Optional[torch.Tensor],
    img_order_table: Optional[torch.Tensor], obj_order_table: Optional[torch.Tensor],
    img_order_ids: Optional[torch.Tensor] = None, obj_order_ids: Optional[torch.Tensor] = None,
    final_ln_w: Optional[torch.Tensor] = None, final_ln_b: Optional[torch.Tensor] = None,
    eps: float = 1e-5, rms: bool = False,
) -> torch.Tensor:
    """``VisualEmbedding.forward`` (src/modeling_bart.py:143-192; T5 src/modeling_t5.py:110-174).

    feat_embedding = [LN](Linear(feats)); pos -> cat(pos, area) -> [LN](Linear);
    + img_order_embedding[ids] + obj_order_embedding[V-1-ids] (:170-183; the obj ids index
    the shared token table from the end); optional single LN when not individual (:187-190).
    ``rms=True`` selects T5's ``T5LayerNorm`` (src/modeling_t5.py:56-73).
    """
    B, N, _ = feats.shape
    d = feat_w.shape[0]

    def norm(v, w, b):
        if rms:  # T5LayerNorm (my_transformers/modeling_t5.py:244-252): no mean, no bias
            var = v.to(torch.float32).pow(2).mean(-1, keepdim=True)
            return w * (v * torch.rsqrt(var + eps))
        return F.layer_norm(v, (d,), w, b, eps)

    fe = F.linear(feats, feat_w, feat_b)
    if feat_ln_w is not None:
        fe = norm(fe, feat_ln_w, feat_ln_b)
    height = pos[:, :, 3] - pos[:, :, 2]
    width = pos[:, :, 1] - pos[:, :, 0]
    area = (height * width).unsqueeze(2)
    p5 = torch.cat([pos, area], dim=2)
    pe = F.linear(p5, pos_w, pos_b)
    if pos_ln_w is not None:
        pe = norm(pe, pos_ln_w, pos_ln_b)
    out = fe + pe
    if img_order_table is not None:
        if img_order_ids is None:
            img_order_ids = torch.zeros(N, dtype=torch.long).unsqueeze(0)
        if obj_order_ids is None:
            obj_order_ids = torch.arange(N, dtype=torch.long).unsqueeze(0)
        obj_ids = obj_order_table.shape[0] - obj_order_ids - 1
        out = out + F.embedding(img_order_ids, img_order_table) + F.embedding(obj_ids, obj_order_table)
    if final_ln_w is not None:
        out = norm(out, final_ln_w, final_ln_b)
    return out


# ----------------------------------------------------------------- sublayer tails
def bart_sublayer_tail(x1, y, ln_w, ln_b, eps=1e-5):
    """``LayerNorm(residual + dropout(y))`` with dropout off
    (my_transformers/modeling_bart.py:1259-1261,1375-1377)."""
    return F.layer_norm(x1 + y, (x1.shape[-1],), ln_w, ln_b, eps)


def t5_sublayer_tail(x1, y):
    """``hidden + dropout(y)`` with dropout off (my_transformers/modeling_t5.py:408,824)."""
    return x1 + y


def downsample(x, out_hw=(6, 6)):
    """CLIP-grid down-sampling: ``AdaptiveMaxPool2d(out_hw)`` over the sqrt(L) x sqrt(L) token grid of
    ``x [B, L, dim]`` -> ``[B, out_h*out_w, dim]`` (src/modeling_bart.py:565-581)."""
    B, Ltok, dim = x.shape
    s = int(Ltok ** 0.5)
    g = x.permute(0, 2, 1).reshape(B, dim, s, s)
    g = F.adaptive_max_pool2d(g, out_hw).reshape(B, dim, -1)
    return g.permute(0, 2, 1)


def downsample_nlvr(x, boxes, img_ids, obj_ids, out_hw=(6, 6)):
    """Two-image (NLVR) form: each half of the token axis is pooled on its own; boxes / ids keep the first
    out_h*out_w entries of each half (src/modeling_bart.py:586-604)."""
    def halves(t):
        return torch.cat(torch.chunk(t, 2, 1), 0)

    def back(t):
        return torch.cat(torch.chunk(t, 2, 0), 1)
    y = back(downsample(halves(x), out_hw))
    n = y.shape[1] // 2
    return y, back(halves(boxes)[:, :n]), back(halves(img_ids)[:, :n]), back(halves(obj_ids)[:, :n])


# ------------------------------------------------------------ autograd helpers
def with_grads(fn, tensors: Dict[str, torch.Tensor], dy: torch.Tensor, wrt: Sequence[str]):
    """Run ``fn(**tensors)`` and backprop ``dy``; returns (out, {name: grad})."""
    leaves = {}
    for k, v in tensors.items():
        if isinstance(v, torch.Tensor) and v.is_floating_point() and k in wrt:
            leaves[k] = v.detach().clone().requires_grad_(True)
        else:
            leaves[k] = v
    out = fn(**leaves)
    out.backward(dy)
    return out.detach(), {k: leaves[k].grad for k in wrt}


def k1_fwd_bwd(x1, x2, wd, bd, wu, bu, wgd, bgd, wgu, bgu, dy, *, n_heads=1, gating_add=False,
               delta_scale=1.0, x2_scale=1.0, gate_scale=1.0, has_gate=True):
    """Convenience wrapper: stacked [r,d] down weight is split into N_h heads exactly as the
    reference ModuleList holds it, then forward + autograd backward.  Returns
    (y, dict of grads for x1,x2 and the eight parameter tensors)."""
    t = dict(x1=x1, x2=x2, wd=wd, bd=bd, wu=wu, bu=bu)
    if has_gate:
        t.update(wgd=wgd, bgd=bgd, wgu=wgu, bgu=bgu)
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in t.items()}
    rh = wd.shape[0] // n_heads
    dws = [leaves["wd"][i * rh:(i + 1) * rh] for i in range(n_heads)]
    dbs = [leaves["bd"][i * rh:(i + 1) * rh] for i in range(n_heads)]
    gate = None
    if has_gate:
        gate = dict(down_w=leaves["wgd"], down_b=leaves["bgd"], up_w=leaves["wgu"], up_b=leaves["bgu"])
    a, b = leaves["x1"], leaves["x2"]
    if a.dim() == 2:
        a, b = a.unsqueeze(0), b.unsqueeze(0)
    y = encoder_adapter_gate(a, b, dws, dbs, leaves["wu"], leaves["bu"], gate,
                             GATE_LARGE if has_gate else GATE_NONE, gating_add,
                             delta_scale, x2_scale, gate_scale)
    y = y.reshape(x2.shape)
    y.backward(dy)
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return y.detach(), grads


def lowrank_visual_embedding(
    feats: torch.Tensor, pos: torch.Tensor,
    down_w: Sequence[torch.Tensor], down_b: Sequence[torch.Tensor], up_w: torch.Tensor, up_b: torch.Tensor,
    ln_w: torch.Tensor, ln_b: torch.Tensor,
    pos_w: torch.Tensor, pos_b: torch.Tensor, pos_ln_w: torch.Tensor, pos_ln_b: torch.Tensor,
    img_order_table: torch.Tensor, obj_order_table: torch.Tensor,
    img_order_ids: Optional[torch.Tensor] = None, obj_order_ids: Optional[torch.Tensor] = None,
    gate: Optional[Dict[str, torch.Tensor]] = None, gate_residual: bool = False, eps: float = 1e-5,
) -> torch.Tensor:
    """``LowRankVisualEmbedding.forward`` (src/modeling_bart.py:251-334): multi-head down projection of the CLIP
    features, gelu_new, up projection, optional low-rank sigmoid gate on the features (``proj * gate`` or, with
    ``use_visual_projector_residual_connection``, ``proj + proj * gate``), LayerNorm, then the same position /
    order-embedding terms as ``VisualEmbedding``."""
    B, N, _ = feats.shape
    d = up_w.shape[0]
    z = gelu_new(torch.cat([F.linear(feats, w, b) for w, b in zip(down_w, down_b)], dim=-1))
    fe = F.linear(z, up_w, up_b)
    if gate is not None:
        g = torch.sigmoid(F.linear(gelu_new(F.linear(feats, gate["down_w"], gate["down_b"])), gate["up_w"], gate["up_b"]))
        fe = fe + fe * g if gate_residual else fe * g
    fe = F.layer_norm(fe, (d,), ln_w, ln_b, eps)
    area = ((pos[:, :, 3] - pos[:, :, 2]) * (pos[:, :, 1] - pos[:, :, 0])).unsqueeze(2)
    pe = F.layer_norm(F.linear(torch.cat([pos, area], dim=2), pos_w, pos_b), (d,), pos_ln_w, pos_ln_b, eps)
    if img_order_ids is None:
        img_order_ids = torch.zeros(N, dtype=torch.long).unsqueeze(0)
    if obj_order_ids is None:
        obj_order_ids = torch.arange(N, dtype=torch.long).unsqueeze(0)
    obj_ids = obj_order_table.shape[0] - obj_order_ids - 1
    return fe + pe + F.embedding(img_order_ids, img_order_table) + F.embedding(obj_ids, obj_order_table)


# ------------------------------------------------------------ optimizer step
def clip_grad_norm(grads: Sequence[torch.Tensor], max_norm: float) -> torch.Tensor:
    """``torch.nn.utils.clip_grad_norm_`` as called at multitask.py:279-300: one global 2-norm over every
    gradient, ``coef = max_norm / (norm + 1e-6)``, gradients scaled in place when ``coef < 1``."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = max_norm / (total + 1e-6)
    if coef < 1:
        for g in grads:
            g.mul_(coef)
    return total


def hf_adamw_step(p, g, m, v, step: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-6, weight_decay: float = 0.0):
    """One update of ``transformers.optimization.AdamW`` (transformers 4.2.1, ``correct_bias=True``), the
    optimizer the reference builds at trainer_base.py:690-701: moments, bias-corrected step size with eps
    added to sqrt(v), then decoupled weight decay ``p -= lr * wd * p``.  In place; ``step`` is 1-based."""
    b1, b2 = betas
    m.mul_(b1).add_(g, alpha=1.0 - b1)
    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    denom = v.sqrt().add_(eps)
    step_size = lr * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    p.addcdiv_(m, denom, value=-step_size)
    if weight_decay > 0.0:
        p.add_(p, alpha=-lr * weight_decay)


def linear_warmup_lr(step: int, base_lr: float, warmup_steps: int, total_steps: int) -> float:
    """``get_linear_schedule_with_warmup`` factor for the update with 0-based index ``step``."""
    if step < warmup_steps:
        return base_lr * float(step) / float(max(1, warmup_steps))
    return base_lr * max(0.0, float(total_steps - step) / float(max(1, total_steps - warmup_steps)))
