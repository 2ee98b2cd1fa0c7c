"""Plain-torch fallbacks for the configurations the fused HIP path does not cover ("fused path when ..., else eager
fallback").  They run on whatever device the tensors are on, through torch's own kernels and autograd -- in-package code that imports
nothing from the test infrastructure.  Every call site keeps the fused path as the default and comes
here only for a variant no launch script of the reference uses: other adapter non-linearities, ``track_z``, low-rank adapters,
non-square / transposed LoRA layers, the shared-LayerNorm visual embedding, a wide bottleneck with an odd number of heads.
Each function restates the reference's op order for its case (file:line in the docstring) so the numerics are the reference's."""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn.functional as F


def adapter(x: torch.Tensor, down_w, down_b, up_w, up_b, act, keep_z=None) -> torch.Tensor:
    """``up(act(down(x)))`` -- adapters/adapter_modeling.py:55-61.  ``keep_z``: a callable that receives the bottleneck (track_z)."""
    z = act(F.linear(x, down_w.to(x.dtype), None if down_b is None else down_b.to(x.dtype)))
    if keep_z is not None:
        keep_z(z)
    return F.linear(z, up_w.to(x.dtype), None if up_b is None else up_b.to(x.dtype))


def adapter_gate(x1: Optional[torch.Tensor], x2: torch.Tensor, down_ws: Sequence[torch.Tensor], down_bs: Sequence[torch.Tensor],
                 up_w: torch.Tensor, up_b: torch.Tensor, gate_params, act, gate_act, mode: str,
                 delta_scale: float = 1.0, x2_scale: float = 1.0, gate_scale: float = 1.0) -> torch.Tensor:
    """The encoder granularity-controlled adapter with the "large" gate, my_transformers/modeling_bart.py:1147-1155, 1195-1209 (T5:
    my_transformers/modeling_t5.py:366-390): y = ((s2 x2 + sd up(act(cat_i down_i(x2)))) (* | +) sigmoid(up_g(act_g(down_g(x1))))) gs.
    mode: "mul" | "add" | "none"."""
    dt = x2.dtype
    z = torch.cat([F.linear(x2, w.to(dt), b.to(dt)) for w, b in zip(down_ws, down_bs)], dim=-1)
    h = x2_scale * x2 + delta_scale * F.linear(act(z), up_w.to(dt), up_b.to(dt))
    if mode == "none":
        return h
    gdw, gdb, guw, gub = gate_params
    g = torch.sigmoid(F.linear(gate_act(F.linear(x1, gdw.to(dt), gdb.to(dt))), guw.to(dt), gub.to(dt)))
    return (h + g if mode == "add" else h * g) * gate_scale


def lora_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], A: torch.Tensor, B: torch.Tensor, scaling: float,
                p: float, training: bool, fan_in_fan_out: bool = False) -> torch.Tensor:
    """``F.linear(x, T(W), b) + (dropout(x) A^T B^T) alpha / r`` -- lora/controller.py:56-70 (T = transpose under fan_in_fan_out)."""
    w = weight.t() if fan_in_fan_out else weight
    out = F.linear(x, w.to(x.dtype), None if bias is None else bias.to(x.dtype))
    xd = F.dropout(x, p) if (training and p > 0.0) else x
    return out + (xd @ A.to(x.dtype).t() @ B.to(x.dtype).t()) * scaling


def visual_embedding(feats: torch.Tensor, R: torch.Tensor, feat_embedding, shared_norm) -> torch.Tensor:
    """VisualEmbedding.forward for the configurations WITHOUT a per-branch LayerNorm (src/modeling_bart.py:157-190): the feature branch
    (a bare Linear there) plus the position / order term R, then the shared LayerNorm if the config has one (:186-188)."""
    v = feat_embedding(feats.to(feat_embedding[0].weight.dtype)).to(R.dtype) + R
    return shared_norm(v) if shared_norm is not None else v


def lowrank_visual_features(feats: torch.Tensor, down, up, act, gate_down, gate_up, gate_act, gate_residual: bool, norm) -> torch.Tensor:
    """The feature branch of LowRankVisualEmbedding (src/modeling_bart.py:276-298): up(act(cat_i down_i(feats))), optionally times (or
    plus itself times) sigmoid(up_g(act_g(down_g(feats)))), then the branch's LayerNorm when the config has one (``norm`` may be None)."""
    dt = down[0].weight.dtype
    f = feats.to(dt)
    fe = up(act(torch.cat([m(f) for m in down], dim=-1)))
    if gate_down is not None:
        g = torch.sigmoid(gate_up(gate_act(gate_down(f))))
        fe = fe + fe * g if gate_residual else fe * g
    return norm(fe) if norm is not None else fe
