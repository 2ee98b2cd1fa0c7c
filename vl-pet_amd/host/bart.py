"""Minimal BART encoder-decoder host with the reference's PET hook points.

This is the frozen backbone the hot path plugs into: plain PyTorch-ROCm (library GEMMs, SDPA,
LayerNorm), no HIP code of ours -- the PET arithmetic is delegated to ``encoder_pet.apply_pet``
(K1), ``adapters.AdapterController`` (K2), ``lora.LoRALinearController`` (K3) and
``visual.VisualEmbedding`` (K4).  Module / parameter names follow the reference so that its
checkpoints and its name-substring freeze rules apply:

  model.shared, model.encoder.{embed_tokens,embed_positions,layernorm_embedding,visual_embedding},
  model.encoder.layers.{l}.{self_attn.{q,k,v,out}_proj,self_attn_layer_norm,fc1,fc2,final_layer_norm,
      attn_adapter_multihead_down.{h},attn_adapter_multihead_up,ff_adapter_multihead_*,
      encoder_{attn,ff}_adapter_gating_large_x_{down,up}},
  model.decoder.layers.{l}.{self_attn,encoder_attn.{...,attn_value_parallel_adapter.adapters.{task}},...}

Structure restated from: src/modeling_bart.py:696-900 (JointEncoder: text LN before concat
[text ; visual]), my_transformers/modeling_bart.py:1122-1380 (encoder layer, post-LN),
:1611-1760 (decoder layer), :397-566 (cross-attention with value-parallel adapter),
src/modeling_bart.py:1522-1602 (LM head + CE with reduction='none').
"""
from __future__ import annotations

import copy
import math
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as VF
from ..adapters import AdapterConfig, AdapterController
from ..encoder_pet import apply_pet, build_pet, has_pet
from ..lora import LoRALinearController, LoraConfig
from ..visual import Downsample, LowRankVisualEmbedding, VisualEmbedding

TASKS = ["vqa", "gqa", "nlvr", "caption"]


def vlpet_config(**over) -> SimpleNamespace:
    """Config namespace with the attribute names the reference copies from its argparse flags onto the
    HF config (trainer_base.py:86-87).  Defaults = scripts/image-text/VL-PET-large.sh with r=96."""
    c = SimpleNamespace(
        # backbone (facebook/bart-base)
        d_model=768, encoder_layers=6, decoder_layers=6, encoder_attention_heads=12, decoder_attention_heads=12,
        encoder_ffn_dim=3072, decoder_ffn_dim=3072, vocab_size=50265 + 200, max_position_embeddings=1024,
        dropout=0.1, attention_dropout=0.1, activation_dropout=0.1, pad_token_id=1, decoder_start_token_id=2,
        scale_embedding=False, init_std=0.02,
        # visual
        feat_dim=2048, pos_dim=4, n_images=2, n_boxes=36, downsample=True, use_vis_order_embedding=True,
        use_vis_layer_norm=True, individual_vis_layer_norm=True, share_vis_lang_layer_norm=False,
        use_lowrank_visual_projector=False, visual_projector_down_dim=96, visual_projector_multihead_num_head=1,
        use_visual_projector_gating_large_x_lowrank=False, visual_projector_gating_down_dim=96,
        use_visual_projector_residual_connection=False,
        # PET flags
        tasks=",".join(TASKS), use_adapter=True, use_single_adapter=True, no_encoder_adapter=True,
        no_decoder_adapter=True, use_adapter_down_dim=True, adapter_down_dim=96,
        use_encoder_adapter_down_multihead=True, encoder_adapter_multihead_num_head=4,
        use_encoder_adapter_gating_large_x_lowrank=True, adapter_gating_down_dim=96,
        use_encoder_adapter_gating_add=False, use_encoder_adapter_gating_small_xy_cat=False,
        use_encoder_adapter_gating_middle_xy_add=False, use_encoder_adapter_gating_middle_ia3_add=False,
        use_encoder_gating_scaling=False, encoder_gating_scaling_factor=1.0,
        use_encoder_adapter_scaling=False, encoder_adapter_scaling_factor=1.0,
        use_encoder_x2_scaling=False, encoder_x2_scaling_factor=1.0,
        unfreeze_encoder_layer_norms=True,
        use_decoder_enc_attn_value_parallel_adapter_down_dim=True, decoder_enc_attn_value_parallel_adapter_down_dim=96,
        use_decoder_enc_attn_value_parallel_adapter_scaling=False,
        decoder_enc_attn_value_parallel_adapter_scaling_factor=1.0,
        use_lora=False, lora_dim=4, lora_alpha=32, lora_dropout=0.1, use_single_lora=False, reduction_factor=8,
        use_encoder_multihead_up_zero_init=False, use_encoder_gating_large_x_lowrank_up_zero_init=False,
        use_decoder_enc_vpa_up_zero_init=False, freeze_vis_emb=False,
    )
    for k, v in over.items():
        if not hasattr(c, k):
            raise AttributeError(f"unknown config field {k}")
        setattr(c, k, v)
    task_list = [t for t in c.tasks.replace(" ", ",").split(",") if t]
    c.task_list = task_list
    if c.use_adapter or c.use_decoder_enc_attn_value_parallel_adapter_down_dim:
        c.adapter_config = AdapterConfig(
            tasks=task_list, input_dim=c.d_model, d_model=c.d_model, use_single_adapter=c.use_single_adapter,
            reduction_factor=c.reduction_factor, use_adapter_down_dim=bool(c.use_adapter_down_dim),
            adapter_down_dim=c.adapter_down_dim)
    else:
        c.adapter_config = None
    c.lora_config = LoraConfig(lora_dim=c.lora_dim, lora_alpha=c.lora_alpha, lora_dropout=c.lora_dropout,
                               tasks=task_list, use_single_lora=c.use_single_lora) if c.use_lora else None
    return c


class HostLayerNorm(nn.LayerNorm):
    """LayerNorm whose (possibly fp32, trainable) affine parameters follow the activation dtype."""

    def forward(self, x):
        w, b = self.weight, self.bias       # (LoRA runs train every bias but no LayerNorm weight: the two may differ in dtype)
        if w.dtype != x.dtype:
            w = w.to(x.dtype)
        if b is not None and b.dtype != x.dtype:
            b = b.to(x.dtype)
        return F.layer_norm(x, self.normalized_shape, w, b, self.eps)


# K1's gate and the sublayer tail both read the sublayer input; with a link the tail's backward hands its d/dx1 to K1's
# backward kernel instead of leaving the sum to an elementwise pass of autograd (functional.ResidualLink).  The CPU parity
# harness of the test suite switches it off: its ops are plain autograd.
FUSE_RESIDUAL_GRAD = True      # (A/B switch, tools/ab_switches.py: False = plain autograd sums)


def _pet_then_tail(layer, which, residual, h, norm, p, training, config, gemm_link=None):
    """``norm(residual + dropout(apply_pet(residual, h)))`` -- K1 followed by K5.  ``gemm_link``: the link the sublayer's first
    projection armed (_first_linear): K1's d/dresidual (which already holds the tail's) goes out through it."""
    if not FUSE_RESIDUAL_GRAD:
        return sublayer_tail(residual, apply_pet(layer, which, residual, h, config), norm, p, training)
    from ..functional import ResidualLink
    link = ResidualLink()
    y = apply_pet(layer, which, residual, h, config, link=link, out_link=gemm_link)
    return sublayer_tail(residual, y, norm, p, training, link=link)


# The same hand-over one op further: the input of a sublayer is also the input of its first frozen projection (q|k|v, q_proj,
# fc1), whose dgrad GEMM accumulates onto the gradient K1 / the tail parked (functional.linear_acc: beta = 1 in the GEMM
# epilogue) instead of autograd adding two [M, d] tensors; in the cross-attention, k_proj / v_proj and the value-parallel
# adapter (K2) all read the encoder output: one gradient per decoder layer instead of three.
FUSE_GEMM_GRAD = True       # (A/B switch, tools/ab_switches.py: False = autograd's adds)


def _frozen(*mods) -> bool:
    return all(isinstance(m, nn.Linear) for m in mods) and not any(
        t is not None and t.requires_grad for m in mods for t in (m.weight, m.bias))


def _new_gemm_link(x):
    """A link for the first projection of a sublayer whose input is ``x``, or None when nothing would use it."""
    if not (FUSE_RESIDUAL_GRAD and FUSE_GEMM_GRAD) or not x.requires_grad or not torch.is_grad_enabled():
        return None
    from ..functional import ResidualLink
    return ResidualLink()


def _first_linear(mod, x, link):
    """``mod(x)`` for the first projection of a sublayer; with a link and a frozen ``nn.Linear``: functional.linear_acc."""
    if link is not None and _frozen(mod):
        from ..functional import linear_acc
        return linear_acc(x, link, mod)
    if (link is not None and FUSE_BIAS_GRAD and isinstance(mod, nn.Linear) and not mod.weight.requires_grad and mod.bias is not None
            and mod.bias.requires_grad):
        # frozen weight, trainable bias (the LoRA runs): the same hand-over through the bias-gradient form of the projection
        from ..functional import linear_train_bias, linear_train_bias_ok
        w = mod.weight if mod.weight.dtype == x.dtype else mod.weight.to(x.dtype)
        if linear_train_bias_ok(x, w, mod.bias):
            return linear_train_bias(x, w, mod.bias, link)
    return _linear(mod, x)


def sublayer_tail(residual, h, norm, p, training, link=None):
    """K5: ``norm(residual + dropout(h))`` -- one fused HIP pass (vlpet_amd.tail); the parity / CPU-baseline harnesses
    swap this module attribute for an eager restatement."""
    from ..tail import sublayer_tail as _hip_tail
    return _hip_tail(residual, h, norm, p, training, link=link)


def _tail_linked(residual, h, norm, p, training, link):
    """K5 whose d/dresidual goes to the sublayer's first dgrad GEMM when that armed ``link`` (the swapped-in eager tail of
    the parity harness takes no link: it is only ever called without one)."""
    if link is None or not link.armed:
        return sublayer_tail(residual, h, norm, p, training)
    return sublayer_tail(residual, h, norm, p, training, link=link)


def ffn_activation(x, act, p, training):
    """``dropout(act(fc1 output), p)`` of the feed-forward sublayer: one fused HIP pass (vlpet_amd.act); the parity /
    CPU-baseline harnesses swap this module attribute for the eager pair."""
    if EAGER_FFN_ACT:
        return F.dropout(F.gelu(x) if act == "gelu" else F.relu(x), p=p, training=training)
    from ..act import act_dropout
    return act_dropout(x, act, p, training)


EAGER_FFN_ACT = False     # A/B switch (tools/ab_switches.py): the two elementwise torch passes instead


def lm_loss(h, weight, labels, bias=None):
    """LM head + per-token cross entropy (vlpet_amd.lmloss: library GEMM, then one fused HIP pass each way over the
    logits); the parity / CPU-baseline harnesses swap this module attribute for the eager chain."""
    if EAGER_LM_LOSS:
        logits = F.linear(h, weight.to(h.dtype))
        if bias is not None:
            logits = logits + bias.to(h.dtype)
        loss = F.cross_entropy(logits.float().view(-1, logits.shape[-1]), labels.view(-1), ignore_index=-100, reduction="none")
        return loss.view(labels.shape), logits
    from ..lmloss import lm_head_loss
    return lm_head_loss(h, weight, labels, bias)


EAGER_LM_LOSS = False       # A/B switch (tools/ab_switches.py): torch's cast + log_softmax + nll chain


def _linear(mod: nn.Linear, x):
    w, b = mod.weight, mod.bias             # (a trainable fp32 bias next to a frozen bf16 weight in LoRA runs)
    if w.dtype != x.dtype:
        w = w.to(x.dtype)
    if FUSE_BIAS_GRAD and b is not None and b.requires_grad:
        from ..functional import linear_train_bias, linear_train_bias_ok
        if linear_train_bias_ok(x, w, b):
            return linear_train_bias(x, w, b)             # bias gradient = column sums of dy on the HIP path
    if b is not None and b.dtype != x.dtype:
        b = b.to(x.dtype)
    return F.linear(x, w, b)


def _sdpa_ctx():
    """Frozen-backbone attention = torch SDPA with its default backend choice; ``SDPA_BACKEND`` = flash | efficient | math pins
    one (an experiment switch for the host model, not part of the PET path)."""
    import contextlib
    which = SDPA_BACKEND
    if not which:
        return contextlib.nullcontext()
    from torch.nn.attention import SDPBackend, sdpa_kernel
    return sdpa_kernel({"flash": SDPBackend.FLASH_ATTENTION, "efficient": SDPBackend.EFFICIENT_ATTENTION,
                        "math": SDPBackend.MATH}[which])


def attention_core(q, k, v, num_heads, attn_mask, causal, p, training, k_slot=None):
    """``dropout(softmax(q k^T / sqrt(d) + mask), p) v`` on the projection outputs ``[B, L, H * d]``.  Sequences of at most
    128 tokens in bf16 with head dim 64 (every image-text shape) run on vlpet_amd.attention's on-chip kernels; anything
    else (fp32 parity runs, the 664-token video encoder, non-boolean masks) on torch's SDPA.  The parity / CPU-baseline
    harnesses swap this module attribute for the eager chain."""
    from .. import attention as A
    B, Lq, E = q.shape
    boolean_key_mask = attn_mask is None or (attn_mask.dtype == torch.bool and attn_mask.dim() == 4
                                             and attn_mask.shape[1] == 1 and attn_mask.shape[2] == 1)
    if not EAGER_ATTENTION and boolean_key_mask and A.supported(q, k, num_heads):
        km = None if attn_mask is None else attn_mask[:, 0, 0, :]
        return A.short_attention(q, k, v, num_heads, km, causal and attn_mask is None, p, training, k_slot=k_slot)
    sh = lambda t: t.reshape(B, -1, num_heads, E // num_heads).transpose(1, 2)
    with _sdpa_ctx():
        out = F.scaled_dot_product_attention(sh(q), sh(k), sh(v), attn_mask=attn_mask, is_causal=causal and attn_mask is None,
                                             dropout_p=p if training else 0.0)
    return out.transpose(1, 2).reshape(B, Lq, E)


FUSE_BIAS_GRAD = True    # A/B switch (tools/ab_switches.py): False = autograd's sum(0) for trainable biases
EAGER_ATTENTION = False   # A/B switch: the library (SDPA) path for every shape
FUSE_QKV = True              # A/B switch: False = separate q / k / v projections in self-attention
FUSE_CROSS_KEYS = True       # A/B switch: False = every decoder layer projects its own cross-attention keys (round 5)
SDPA_BACKEND = None          # A/B switch: "flash" | "efficient" | "math" pins torch SDPA's backend on the library path


class BartAttention(nn.Module):
    """Multi-head attention; optional LoRA on q/v (my_transformers/modeling_bart.py:738-879) and optional
    value-parallel adapter on the cross-attention value (:283-566, use at :427-430)."""

    def __init__(self, config, embed_dim, num_heads, dropout, is_cross=False, value_adapter=False):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.use_lora = bool(config.use_lora)
        if self.use_lora:
            self.q_proj = LoRALinearController(embed_dim, embed_dim, config=config.lora_config, bias=True)
            self.v_proj = LoRALinearController(embed_dim, embed_dim, config=config.lora_config, bias=True)
        else:
            self.q_proj = nn.Linear(embed_dim, embed_dim)
            self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self.attn_value_parallel_adapter = None
        if value_adapter:
            ac = copy.deepcopy(config.adapter_config)
            ac.use_adapter_down_dim = True
            ac.adapter_down_dim = config.decoder_enc_attn_value_parallel_adapter_down_dim
            ac.use_parallel_adapter = True
            if config.use_decoder_enc_attn_value_parallel_adapter_scaling:
                ac.use_scaling_factor = True
                ac.scaling_factor = config.decoder_enc_attn_value_parallel_adapter_scaling_factor
            self.attn_value_parallel_adapter = AdapterController(ac)

    def _shape(self, t, B):
        return t.view(B, -1, self.num_heads, self.head_dim).transpose(1, 2)

    def _fused_qkv(self, dtype):
        """The frozen q | k | v projections of a self-attention as one [3E, E] weight (a derived cache keyed on the three
        modules' tensors: rebuilt when any of them changes; never a parameter, the state dict keeps q_proj / k_proj / v_proj)."""
        mods = (self.q_proj, self.k_proj, self.v_proj)
        key = tuple((m.weight.data_ptr(), m.weight._version, m.bias.data_ptr(), m.bias._version) for m in mods) + (dtype, VF.FROZEN_EPOCH)
        c = getattr(self, "_qkv_cache", None)
        if c is None or c[0] != key:
            with torch.no_grad():
                w = torch.cat([m.weight.to(dtype) for m in mods], 0).contiguous()
                b = torch.cat([m.bias.to(dtype) for m in mods], 0).contiguous()
            c = (key, w, b)
            self._qkv_cache = c
        return c[1], c[2]

    def forward(self, hidden, kv=None, attn_mask=None, causal=False, task=None, in_link=None, k_pre=None):
        """``in_link``: see _first_linear -- armed by the projection of ``hidden`` (q|k|v or q_proj) when that is frozen.
        ``k_pre`` = (k, k_slot): this layer's keys already projected (cross-attention: a column block of the decoder's fused key
        projection of the encoder output, BartDecoder._cross_keys)."""
        B, L, _ = hidden.shape
        src = hidden if kv is None else kv
        if kv is None and FUSE_QKV and not self.use_lora and not EAGER_ATTENTION:
            # self-attention with frozen projections: one [E -> 3E] GEMM each way, the attention kernels read / write the
            # q | k | v column blocks in place (one input gradient instead of three that autograd would have to sum)
            from .. import attention as A
            frozen = not any(t.requires_grad for m in (self.q_proj, self.k_proj, self.v_proj) for t in (m.weight, m.bias))
            boolean_key_mask = attn_mask is None or (attn_mask.dtype == torch.bool and attn_mask.dim() == 4
                                                     and attn_mask.shape[1] == 1 and attn_mask.shape[2] == 1)
            if (frozen and boolean_key_mask and hidden.is_cuda and hidden.dtype == torch.bfloat16 and L <= A.MAX_LEN
                    and self.head_dim == A.HEAD_DIM):
                w, b = self._fused_qkv(hidden.dtype)
                if in_link is not None:
                    from ..functional import linear_acc
                    qkv = linear_acc(hidden, in_link, (w, b))
                else:
                    qkv = F.linear(hidden, w, b)
                km = None if attn_mask is None else attn_mask[:, 0, 0, :]
                out = A.short_self_attention(qkv, self.num_heads, km, causal and attn_mask is None, self.dropout, self.training)
                return _linear(self.out_proj, out)
        if self.use_lora:
            q = self.q_proj(hidden, task)
            v = self.v_proj(src, task)
            k = _linear(self.k_proj, src)
        else:
            # (self-attention off the fused path: hidden feeds three projections -- only q_proj takes the sublayer's link)
            q = _first_linear(self.q_proj, hidden, in_link)
            kv_link = _new_gemm_link(src) if (kv is not None and _frozen(self.k_proj, self.v_proj)) else None
            if kv_link is not None:
                from ..functional import linear_acc
                k_slot = None
                if k_pre is not None:
                    (k, k_slot), v = k_pre, linear_acc(src, kv_link, self.v_proj)
                else:
                    k, v = linear_acc(src, kv_link, self.k_proj, self.v_proj)      # one gradient for the encoder output, K2's included
                if self.attn_value_parallel_adapter is not None:
                    v = self.attn_value_parallel_adapter(src, task, y=v, link=kv_link)
                out = attention_core(q, k, v, self.num_heads, attn_mask, causal, self.dropout, self.training, k_slot=k_slot)
                return _linear(self.out_proj, out)
            v = _linear(self.v_proj, src)
            k = _linear(self.k_proj, src)
        if kv is not None and self.attn_value_parallel_adapter is not None:
            v = self.attn_value_parallel_adapter(src, task, y=v)
        out = attention_core(q, k, v, self.num_heads, attn_mask, causal, self.dropout, self.training)
        return _linear(self.out_proj, out)


class BartEncoderLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        d = config.d_model
        self.embed_dim = d
        self.self_attn = BartAttention(config, d, config.encoder_attention_heads, config.attention_dropout)
        self.self_attn_layer_norm = HostLayerNorm(d)
        self.dropout, self.activation_dropout = config.dropout, config.activation_dropout
        self.fc1 = nn.Linear(d, config.encoder_ffn_dim)
        self.fc2 = nn.Linear(config.encoder_ffn_dim, d)
        self.final_layer_norm = HostLayerNorm(d)
        build_pet(self, config, d, ("attn", "ff"))
        # the reference's BART encoder layer has no adapter / x2 scaling (my_transformers/modeling_bart.py:1147-1155 is a plain
        # h + up(...)); only the T5 layers read those flags (my_transformers/modeling_t5.py:373-377, 789-793)
        self.pet_config = copy.copy(config)
        self.pet_config.use_encoder_adapter_scaling = False
        self.pet_config.use_encoder_x2_scaling = False

    def forward(self, hidden, attn_mask=None, task=None):
        residual = hidden
        gl = _new_gemm_link(hidden)
        h = self.self_attn(hidden, attn_mask=attn_mask, task=task, in_link=gl)
        if has_pet(self, "attn"):                                                   # K1 + K5
            hidden = _pet_then_tail(self, "attn", residual, h, self.self_attn_layer_norm, self.dropout, self.training,
                                    self.pet_config, gemm_link=gl)
        else:
            hidden = _tail_linked(residual, h, self.self_attn_layer_norm, self.dropout, self.training, gl)   # K5
        residual = hidden
        gl = _new_gemm_link(hidden)
        h = ffn_activation(_first_linear(self.fc1, hidden, gl), "gelu", self.activation_dropout, self.training)
        h = _linear(self.fc2, h)
        if has_pet(self, "ff"):                                                     # K1 + K5
            return _pet_then_tail(self, "ff", residual, h, self.final_layer_norm, self.dropout, self.training, self.pet_config,
                                  gemm_link=gl)
        return _tail_linked(residual, h, self.final_layer_norm, self.dropout, self.training, gl)             # K5


class BartDecoderLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        d = config.d_model
        self.self_attn = BartAttention(config, d, config.decoder_attention_heads, config.attention_dropout)
        self.self_attn_layer_norm = HostLayerNorm(d)
        vpa = bool(config.use_decoder_enc_attn_value_parallel_adapter_down_dim) and not config.use_lora
        self.encoder_attn = BartAttention(config, d, config.decoder_attention_heads, config.attention_dropout,
                                          is_cross=True, value_adapter=vpa)
        self.encoder_attn_layer_norm = HostLayerNorm(d)
        self.dropout, self.activation_dropout = config.dropout, config.activation_dropout
        self.fc1 = nn.Linear(d, config.decoder_ffn_dim)
        self.fc2 = nn.Linear(config.decoder_ffn_dim, d)
        self.final_layer_norm = HostLayerNorm(d)

    def forward(self, hidden, enc, enc_mask=None, task=None, k_pre=None):
        residual = hidden
        gl = _new_gemm_link(hidden)
        h = self.self_attn(hidden, causal=True, task=task, in_link=gl)
        hidden = _tail_linked(residual, h, self.self_attn_layer_norm, self.dropout, self.training, gl)   # K5
        residual = hidden
        gl = _new_gemm_link(hidden)
        h = self.encoder_attn(hidden, kv=enc, attn_mask=enc_mask, task=task, in_link=gl, k_pre=k_pre)      # K2 inside
        hidden = _tail_linked(residual, h, self.encoder_attn_layer_norm, self.dropout, self.training, gl)  # K5
        residual = hidden
        gl = _new_gemm_link(hidden)
        h = ffn_activation(_first_linear(self.fc1, hidden, gl), "gelu", self.activation_dropout, self.training)
        h = _linear(self.fc2, h)
        return _tail_linked(residual, h, self.final_layer_norm, self.dropout, self.training, gl)             # K5


class LearnedPositionalEmbedding(nn.Embedding):
    """BART's learned positions with the historical offset of 2."""

    def __init__(self, num, dim):
        super().__init__(num + 2, dim)

    def forward(self, L: int, device):
        return super().forward(torch.arange(2, L + 2, device=device))


class JointEncoder(nn.Module):
    def __init__(self, config, embed_tokens):
        super().__init__()
        self.config = config
        d = config.d_model
        self.dropout = config.dropout
        self.embed_scale = math.sqrt(d) if config.scale_embedding else 1.0
        self.embed_tokens = embed_tokens
        self.embed_positions = LearnedPositionalEmbedding(config.max_position_embeddings, d)
        self.layers = nn.ModuleList([BartEncoderLayer(config) for _ in range(config.encoder_layers)])
        self.layernorm_embedding = HostLayerNorm(d)
        # src/modeling_bart.py:704-709: the low-rank projector replaces the full feat_dim x d_model Linear when asked for
        vis_cls = LowRankVisualEmbedding if getattr(config, "use_lowrank_visual_projector", False) else VisualEmbedding
        self.visual_embedding = vis_cls(config, self.embed_tokens)
        self.downsample = None
        if config.downsample:
            s = int(config.n_boxes ** 0.5)
            self.downsample = Downsample((s, s))

    def forward(self, input_ids, vis_inputs, attention_mask=None, task=None, no_padding=False):
        B, L = input_ids.shape
        if attention_mask is None and not no_padding:
            # src/modeling_bart.py:817-818: the text mask defaults to input_ids != pad; it also becomes the decoder's
            # cross-attention mask (:995-996).  ``no_padding`` = the loader's promise that no row is padded (every
            # row has the full length -- true for the synthetic batches), which keeps attention on the unmasked path
            attention_mask = input_ids.ne(self.config.pad_token_id)
        x = self.embed_tokens(input_ids) * self.embed_scale + self.embed_positions(L, input_ids.device)
        if self.downsample is not None:
            # fp32 CLIP features -> compute dtype inside the pooling kernel (rounding is monotone:
            # pool(round(f)) == round(pool(f)))
            vis_inputs = self.downsample(vis_inputs, out_dtype=x.dtype)
        elif vis_inputs[0].dtype != x.dtype:
            vis_inputs = (vis_inputs[0].to(x.dtype),) + tuple(vis_inputs[1:])
        feats, boxes = vis_inputs[0], vis_inputs[1]
        img_ids = vis_inputs[2] if len(vis_inputs) >= 3 else None
        obj_ids = vis_inputs[3] if len(vis_inputs) == 4 else None
        vis = self.visual_embedding(feats, boxes, img_ids, obj_ids).to(x.dtype)   # K4
        if self.config.share_vis_lang_layer_norm:
            x = self.layernorm_embedding(torch.cat([x, vis], dim=1))
            x = F.dropout(x, p=self.dropout, training=self.training)
        else:
            from ..act import concat_dropout
            x = concat_dropout(self.layernorm_embedding(x), vis, self.dropout, self.training)      # cat + dropout: one pass each way
        mask = None
        if attention_mask is not None:      # [B, L] text padding mask; visual tokens always attended
            full = torch.cat([attention_mask.bool(), torch.ones(B, vis.shape[1], dtype=torch.bool,
                                                                device=x.device)], dim=1)
            mask = full[:, None, None, :]
        for layer in self.layers:
            x = layer(x, mask, task)
        return x, mask


class BartDecoder(nn.Module):
    def __init__(self, config, embed_tokens):
        super().__init__()
        d = config.d_model
        self.dropout = config.dropout
        self.embed_scale = math.sqrt(d) if config.scale_embedding else 1.0
        self.embed_tokens = embed_tokens
        self.embed_positions = LearnedPositionalEmbedding(config.max_position_embeddings, d)
        self.layers = nn.ModuleList([BartDecoderLayer(config) for _ in range(config.decoder_layers)])
        self.layernorm_embedding = HostLayerNorm(d)

    def forward(self, input_ids, enc, enc_mask=None, task=None):
        B, L = input_ids.shape
        x = self.embed_tokens(input_ids) * self.embed_scale + self.embed_positions(L, input_ids.device)
        x = F.dropout(self.layernorm_embedding(x), p=self.dropout, training=self.training)
        from ..functional import fanout
        n = len(self.layers)
        fused_keys = self._cross_keys_ok(enc, enc_mask)
        encs = fanout(enc, n + (1 if fused_keys else 0))        # one gradient sum for the encoder output instead of autograd's pairwise adds
        ks = self._cross_keys(encs[n]) if fused_keys else None
        for i, (layer, e) in enumerate(zip(self.layers, encs)):
            x = layer(x, e, enc_mask, task, k_pre=None if ks is None else (ks[0][i], None if ks[1] is None else (ks[1], i)))
        return x

    def _cross_keys_ok(self, enc, enc_mask) -> bool:
        """The layers' cross-attention key projections as ONE GEMM (functional.cross_key_blocks): frozen plain projections, bf16 on the
        GPU, a shape the short-sequence attention kernels take (they read a layer's keys as a column block in place)."""
        from .. import attention as A
        if not FUSE_CROSS_KEYS or EAGER_ATTENTION or len(self.layers) < 2 or not enc.is_cuda or enc.dtype != torch.bfloat16:
            return False
        a0 = self.layers[0].encoder_attn
        if enc.shape[1] > A.MAX_LEN or a0.head_dim != A.HEAD_DIM:
            return False
        if enc_mask is not None and not (enc_mask.dtype == torch.bool and enc_mask.dim() == 4 and enc_mask.shape[1] == 1 and enc_mask.shape[2] == 1):
            return False
        return all((not l.encoder_attn.use_lora) and _frozen(l.encoder_attn.k_proj, l.encoder_attn.v_proj) for l in self.layers)

    def _cross_keys(self, enc):
        from .. import functional as VF
        mods = [l.encoder_attn.k_proj for l in self.layers]
        key = (enc.dtype, VF.FROZEN_EPOCH) + tuple((m.weight.data_ptr(), m.weight._version, m.bias.data_ptr(), m.bias._version) for m in mods)
        c = getattr(self, "_ck_cache", None)
        if c is None or c[0] != key:
            with torch.no_grad():
                w = torch.cat([m.weight.to(enc.dtype) for m in mods], 0).contiguous()
                b = torch.cat([m.bias.to(enc.dtype) for m in mods], 0).contiguous()
            c = self._ck_cache = (key, w, b)
        return VF.cross_key_blocks(enc, c[1], c[2], len(mods))


class VLBartModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.shared = nn.Embedding(config.vocab_size, config.d_model, padding_idx=config.pad_token_id)
        self.encoder = JointEncoder(config, self.shared)
        self.decoder = BartDecoder(config, self.shared)


def shift_tokens_right(labels, pad_id, start_id):
    out = labels.new_zeros(labels.shape)
    out[:, 1:] = labels[:, :-1]
    out[:, 0] = start_id
    return out.masked_fill(out == -100, pad_id)


class VLBart(nn.Module):
    """Encoder-decoder + tied LM head; ``forward`` returns the per-token loss [B, L] like the
    reference's ``reduce_loss=False`` path (src/modeling_bart.py:1574-1586)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.model = VLBartModel(config)
        self.register_buffer("final_logits_bias", torch.zeros(1, config.vocab_size))
        self.apply(self._init_weights)
        # derived copies of frozen tensors (fused q|k|v, padded LM head, fp32 norms) must not outlive a checkpoint load
        self.register_load_state_dict_post_hook(lambda module, incompatible: VF.invalidate_caches())

    def _init_weights(self, m):
        # my_transformers/modeling_bart.py:1819-1828: every Linear (adapters included) ~ N(0, 0.02), zero bias
        std = self.config.init_std
        if isinstance(m, nn.Linear):
            m.weight.data.normal_(0.0, std)
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.Embedding):
            m.weight.data.normal_(0.0, std)
            if m.padding_idx is not None:
                m.weight.data[m.padding_idx].zero_()
        if isinstance(m, LoRALinearController):
            for t in m.tasks:
                nn.init.zeros_(m.lora_Bs[t])

    def _logits_bias(self):
        """final_logits_bias (src/modeling_bart.py:1470, 1574) is a zero buffer unless a checkpoint carries one: checked once
        per buffer version (one host sync), so the usual all-zero case adds no pass over the logits."""
        b = self.final_logits_bias
        key = (b.data_ptr(), b._version, VF.FROZEN_EPOCH)
        if getattr(self, "_bias_key", None) != key:
            self._bias_key, self._bias_nonzero = key, bool(b.any())
        return b if self._bias_nonzero else None

    def forward(self, input_ids, vis_inputs, labels, task, attention_mask=None, no_padding=False):
        cfg = self.config
        enc, mask = self.model.encoder(input_ids, vis_inputs, attention_mask, task, no_padding)
        dec_in = shift_tokens_right(labels, cfg.pad_token_id, cfg.decoder_start_token_id)
        h = self.model.decoder(dec_in, enc, mask, task)
        return lm_loss(h, self.model.shared.weight, labels, self._logits_bias())
