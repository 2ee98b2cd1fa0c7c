"""Minimal VL-T5 host (frozen T5 encoder-decoder in plain PyTorch-ROCm: relative-position attention, RMS
norms, ReLU feed-forward), no HIP code of ours -- the PET arithmetic is delegated to ``encoder_pet.apply_pet``
(K1, with the T5-only delta / x2 / gate scaling factors), ``adapters.AdapterController`` (K2 on the
cross-attention value), ``visual.VisualEmbedding`` (K4, RMS-norm form) and ``tail.sublayer_tail`` (K5 without a
norm: T5 is pre-LN, the tail is ``hidden + dropout(y)``).

Mirrors, with the reference's parameter names (state-dict compatible):
  my_transformers/modeling_t5.py:288-410 (T5LayerFF + inline adapter/gate), :412-678 (T5Attention incl.
  ``project_vpa``), :679-827 (T5LayerSelfAttention + inline adapter/gate), :829-894 (T5LayerCrossAttention),
  :896-1000 (T5Block), :1090-1458 (T5Stack);  src/modeling_t5.py:176-405 (JointEncoder: [text ; visual] order,
  relative-position bias only inside the text block), :407-700 (VLT5: tied head rescaled by d_model**-0.5).
"""
from __future__ import annotations

import copy
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as VF
from ..adapters import AdapterConfig, AdapterController
from ..encoder_pet import apply_pet, build_pet, has_pet
from ..visual import Downsample, T5LayerNorm, VisualEmbedding
from .bart import TASKS, _linear


# K1's gate and the sublayer tail both read the sublayer input; with a link the tail's backward hands its d/dx1 to K1's
# backward kernel instead of leaving the sum to an elementwise pass of autograd (functional.ResidualLink).  The CPU parity
# harness of the test suite switches it off: its ops are plain autograd.
FUSE_RESIDUAL_GRAD = True      # (A/B switch, tools/ab_switches.py: False = plain autograd sums)


def _pet_then_tail(layer, which, residual, h, norm, p, training, config, norm_link=None):
    """``norm(residual + dropout(apply_pet(residual, h)))`` -- K1 followed by K5.  ``norm_link``: armed by the sublayer's RMS norm
    (which read ``residual`` first): K1's d/dresidual -- the tail's included -- goes out through it."""
    if not FUSE_RESIDUAL_GRAD:
        return sublayer_tail(residual, apply_pet(layer, which, residual, h, config), norm, p, training)
    from ..functional import ResidualLink
    link = ResidualLink()
    y = apply_pet(layer, which, residual, h, config, link=link, out_link=norm_link)
    nxt = _fused_tail_ok(layer, y) if norm is None else None
    if nxt is not None:
        from ..tail import sublayer_tail_rms
        return sublayer_tail_rms(residual, y, nxt, p, training, link=link)
    return sublayer_tail(residual, y, norm, p, training, link=link)


# The pre-norm residual stream of a T5 sublayer is read by the RMS norm in front of it and by the tail's add behind it (and by
# K1's gate in the encoder): the later reader parks its gradient and the norm's backward kernel adds it (tail.rms_norm's link) --
# one gradient for the stream, no elementwise add by autograd (my_transformers/modeling_t5.py:366, 408; :782, 824).
FUSE_NORM_GRAD = True     # (A/B switch, tools/ab_switches.py: False = autograd's adds)


def _new_norm_link(x):
    fused = getattr(x, "_vlpet_norm", None)     # x came out of a fused tail + norm: its later readers park their gradients in THAT op's link
    if fused is not None:
        return fused.link
    if not (FUSE_RESIDUAL_GRAD and FUSE_NORM_GRAD) or not x.is_cuda or not x.requires_grad or not torch.is_grad_enabled():
        return None
    from ..functional import ResidualLink
    return ResidualLink()


def _normed(norm, hidden, link):
    return norm(hidden) if link is None else norm(hidden, link=link)


# The sum a T5 tail produces is read next by the following sublayer's RMS norm: the tail kernel applies that norm to the row it has just
# formed and writes both (tail.sublayer_tail_rms: one pass instead of two forward, one instead of two backward).  Each sublayer module
# knows the norm that follows it (``_next_norm``, wired by ``_wire_next_norms`` when the stack is built; not a registered submodule).
FUSE_TAIL_NORM = True     # (A/B switch: False = the tail and the next norm as two launches each way)


def _fused_tail_ok(layer, y):
    nn_ = getattr(layer, "_next_norm", None)
    return (nn_[0] if (FUSE_TAIL_NORM and FUSE_RESIDUAL_GRAD and FUSE_NORM_GRAD and nn_ is not None and y.is_cuda
                       and sublayer_tail is _HIP_SUBLAYER_TAIL and y.shape[-1] % 8 == 0) else None)


def _tail_linked(hidden, y, p, training, link, layer=None):
    nxt = _fused_tail_ok(layer, y) if layer is not None else None
    if nxt is not None:
        from ..tail import sublayer_tail_rms
        return sublayer_tail_rms(hidden, y, nxt, p, training, link=link if (link is not None and link.armed) else None)
    if link is None or not link.armed:
        return sublayer_tail(hidden, y, None, p, training)
    return sublayer_tail(hidden, y, None, p, training, link=link)


def ffn_activation(x, act, p, training):
    """``dropout(relu(wi output))`` of T5DenseReluDense as one fused HIP pass (vlpet_amd.act); harnesses swap this
    attribute for the eager pair."""
    from ..act import act_dropout
    return act_dropout(x, act, p, training)


def lm_loss(h, weight, labels, bias=None):
    """LM head + per-token cross entropy (vlpet_amd.lmloss); harnesses swap this attribute for the eager chain."""
    from ..lmloss import lm_head_loss
    return lm_head_loss(h, weight, labels, bias)


def sublayer_tail(residual, h, norm, p, training, link=None):
    """K5 (T5 form): ``residual + dropout(h)`` -- one fused HIP pass; harnesses swap this attribute for an eager
    restatement, exactly as for host.bart."""
    from ..tail import sublayer_tail as _hip_tail
    return _hip_tail(residual, h, norm, p, training, link=link)


_HIP_SUBLAYER_TAIL = sublayer_tail      # (a parity harness that swaps ``sublayer_tail`` for an eager restatement also switches the fusion off)


def _wire_next_norms(blocks, final_norm):
    """Every sublayer module of a stack learns which T5LayerNorm reads its output: the next sublayer's, the next block's first, or
    the stack's final norm (a 1-tuple attribute, so that the norm is not registered a second time as a submodule)."""
    subs = [m for b in blocks for m in b.layer]
    for i, m in enumerate(subs):
        object.__setattr__(m, "_next_norm", ((subs[i + 1].layer_norm if i + 1 < len(subs) else final_norm),))


def vlt5_config(**over) -> SimpleNamespace:
    """t5-base + scripts/image-text/T5-VL-PET-large.sh (r = r_g = 192, decoder value adapter 96, gate scale 0.3)."""
    c = SimpleNamespace(
        d_model=768, d_kv=64, d_ff=3072, num_layers=12, num_decoder_layers=12, num_heads=12,
        relative_attention_num_buckets=32, dropout_rate=0.1, layer_norm_epsilon=1e-6, vocab_size=32100 + 100,
        pad_token_id=0, decoder_start_token_id=0, initializer_factor=1.0,
        feat_dim=2048, pos_dim=4, n_images=2, n_boxes=36, downsample=True, use_vis_order_embedding=True,
        use_vis_layer_norm=True, individual_vis_layer_norm=True, share_vis_lang_layer_norm=False,
        tasks=",".join(TASKS), use_adapter=True, use_single_adapter=True, no_encoder_adapter=True,
        no_decoder_adapter=True, use_adapter_down_dim=True, adapter_down_dim=192,
        use_encoder_adapter_down_multihead=True, encoder_adapter_multihead_num_head=4,
        use_encoder_adapter_gating_large_x_lowrank=True, adapter_gating_down_dim=192,
        use_encoder_adapter_gating_add=False, use_encoder_adapter_gating_small_xy_cat=False,
        use_encoder_adapter_gating_middle_xy_add=False, use_encoder_adapter_gating_middle_ia3_add=False,
        use_encoder_gating_scaling=True, encoder_gating_scaling_factor=0.3,
        use_encoder_adapter_scaling=False, encoder_adapter_scaling_factor=1.0,
        use_encoder_x2_scaling=False, encoder_x2_scaling_factor=1.0,
        unfreeze_encoder_layer_norms=True,
        use_decoder_enc_attn_value_parallel_adapter_down_dim=True, decoder_enc_attn_value_parallel_adapter_down_dim=96,
        use_decoder_enc_attn_value_parallel_adapter_scaling=False,
        decoder_enc_attn_value_parallel_adapter_scaling_factor=1.0,
        use_lora=False, reduction_factor=8,
        use_encoder_multihead_up_zero_init=True, use_encoder_gating_large_x_lowrank_up_zero_init=True,
        use_decoder_enc_vpa_up_zero_init=True, freeze_vis_emb=False,
    )
    for k, v in over.items():
        if not hasattr(c, k):
            raise AttributeError(f"unknown config field {k}")
        setattr(c, k, v)
    c.task_list = [t for t in c.tasks.replace(" ", ",").split(",") if t]
    c.adapter_config = AdapterConfig(
        tasks=c.task_list, input_dim=c.d_model, d_model=c.d_model, use_single_adapter=c.use_single_adapter,
        reduction_factor=c.reduction_factor, use_adapter_down_dim=bool(c.use_adapter_down_dim),
        adapter_down_dim=c.adapter_down_dim)
    c.lora_config = None
    return c


def relative_position_bucket(relative_position, bidirectional, num_buckets=32, max_distance=128):
    """T5's log-spaced relative-position buckets (my_transformers/modeling_t5.py:464-507; Mesh-TensorFlow rule)."""
    buckets = torch.zeros_like(relative_position)
    if bidirectional:
        num_buckets //= 2
        buckets = buckets + (relative_position > 0).long() * num_buckets
        relative_position = relative_position.abs()
    else:
        relative_position = -torch.min(relative_position, torch.zeros_like(relative_position))
    max_exact = num_buckets // 2
    small = relative_position < max_exact
    large = max_exact + (torch.log(relative_position.float() / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).long()
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(small, relative_position, large)


EAGER_ATTENTION = False   # A/B switch (tools/ab_switches.py): True = torch SDPA with a dense additive mask for every T5 attention
FUSE_QKV = True           # A/B switch: False = separate q / k / v projections in self-attention
FUSE_CROSS_KEYS = True    # A/B switch: False = every decoder block projects its own cross-attention keys (host/bart.py: the same fusion)


class AttnSpec:
    """What a T5 attention adds to its scores, kept in parts: the relative-position bias shared by the batch ``rel``
    [1, H, Lq, Lk] (or None), the key padding ``keep`` [B, Lk] (1 = attend, or None) and causality.  ``dense()`` is the merged
    additive mask the reference builds (my_transformers/modeling_t5.py:640-660; src/modeling_t5.py:311-327 for the joint
    encoder) and torch's SDPA takes; ``fast()`` the form of vlpet_amd.attention's on-chip kernels -- an ``AttnBias`` built once per
    forward for all layers + the boolean key mask + the causal flag -- which removes the [B, H, L, L] mask tensor and the library's
    long-sequence flash kernels from the step (round 4: 220 -> ~70 us per encoder attention backward at B = 300, S = 56)."""

    def __init__(self, rel, keep, causal, rel_trainable=False):
        self.rel, self.keep, self.causal, self.rel_trainable = rel, keep, bool(causal), bool(rel_trainable)
        self._dense, self._fast = {}, None

    def dense(self, dtype):
        if dtype not in self._dense:
            m = None
            if self.rel is not None:
                m = self.rel
            if self.causal:
                L = self.rel.shape[-1]
                tri = torch.tril(torch.ones(L, L, device=self.rel.device))
                m = m + (1.0 - tri)[None, None] * -10000.0
            if self.keep is not None:
                pad = (1.0 - self.keep[:, None, None, :].to(torch.float32)) * (-10000.0 if self.rel is not None else -1e9)
                m = pad if m is None else m + pad
            self._dense[dtype] = None if m is None else m.to(dtype)
        return self._dense[dtype]

    def fast(self):
        """(AttnBias or None, key_mask uint8 or None) for the short-sequence kernels, or None when they do not apply."""
        if self.rel_trainable:
            return None
        if self._fast is None:
            from .. import attention as A
            bias = A.AttnBias(self.rel) if self.rel is not None else None
            km = None if self.keep is None else (self.keep > 0.5).to(torch.uint8).contiguous()
            self._fast = (bias, km)
        return self._fast


class T5Attention(nn.Module):
    """softmax(Q K^T + bias) V without the 1/sqrt(d) scale (folded into T5's initialisation); optional
    value-parallel adapter on the cross-attention value (``project_vpa``, :588-613)."""

    def __init__(self, config, is_decoder, has_relative_attention_bias=False, value_adapter=False):
        super().__init__()
        self.is_decoder = is_decoder
        self.n_heads, self.d_kv = config.num_heads, config.d_kv
        self.inner = self.n_heads * self.d_kv
        self.dropout = config.dropout_rate
        self.num_buckets = config.relative_attention_num_buckets
        d = config.d_model
        self.q = nn.Linear(d, self.inner, bias=False)
        self.k = nn.Linear(d, self.inner, bias=False)
        self.v = nn.Linear(d, self.inner, bias=False)
        self.o = nn.Linear(self.inner, d, bias=False)
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
        self.has_relative_attention_bias = has_relative_attention_bias
        if has_relative_attention_bias:
            self.relative_attention_bias = nn.Embedding(self.num_buckets, self.n_heads)

    def compute_bias(self, q_len, k_len):
        dev = self.relative_attention_bias.weight.device
        ctx = torch.arange(q_len, dtype=torch.long, device=dev)[:, None]
        mem = torch.arange(k_len, dtype=torch.long, device=dev)[None, :]
        bucket = relative_position_bucket(mem - ctx, bidirectional=not self.is_decoder, num_buckets=self.num_buckets)
        return self.relative_attention_bias(bucket).permute(2, 0, 1).unsqueeze(0)        # [1, H, q, k]

    def _shape(self, t, B):
        return t.view(B, -1, self.n_heads, self.d_kv).transpose(1, 2)

    def _fused_qkv(self, dtype):
        """The frozen q | k | v projections of a self-attention as one [3 inner, d] weight (as host/bart.py: a derived cache keyed on the
        three modules' tensors, never a parameter -- the state dict keeps q / k / v)."""
        mods = (self.q, self.k, self.v)
        key = tuple((m.weight.data_ptr(), m.weight._version) for m in mods) + (dtype, VF.FROZEN_EPOCH)
        c = getattr(self, "_qkv_cache", None)
        if c is None or c[0] != key:
            with torch.no_grad():
                w = torch.cat([m.weight.to(dtype) for m in mods], 0).contiguous()
            c = (key, w)
            self._qkv_cache = c
        return c[1]

    def forward(self, hidden, bias, kv=None, task=None, k_pre=None):
        """``k_pre`` = (k, k_slot): this block's cross-attention keys, a column block of the decoder's fused key projection (T5Stack._cross_keys)"""
        B, Lq, _ = hidden.shape
        src = hidden if kv is None else kv
        from .. import attention as A
        k_slot = None
        spec = bias if isinstance(bias, AttnSpec) else None
        if (kv is None and FUSE_QKV and spec is not None and not EAGER_ATTENTION and hidden.is_cuda and hidden.dtype == torch.bfloat16
                and self.d_kv == A.HEAD_DIM and Lq <= A.MAX_LEN and not any(m.weight.requires_grad for m in (self.q, self.k, self.v))):
            fast = spec.fast()
            if fast is not None:
                # self-attention with frozen projections: ONE [d -> 3 inner] GEMM each way (T5's projections carry no bias), the
                # attention kernels read / write the q | k | v column blocks in place -- 2 library launches instead of 6 and one
                # input gradient instead of three (at the per-rank batch of an 8-GPU run each of those GEMMs is a ~9 us launch)
                w = self._fused_qkv(hidden.dtype)
                if FUSE_RESIDUAL_GRAD and FUSE_NORM_GRAD and torch.is_grad_enabled() and hidden.requires_grad:
                    from ..functional import linear_acc
                    qkv = linear_acc(hidden, None, (w, None))
                else:
                    qkv = F.linear(hidden, w)
                out = A.short_self_attention(qkv, self.n_heads, fast[1], spec.causal, self.dropout, self.training, scale=1.0, bias=fast[0])
                return _linear(self.o, out)
        fused = (FUSE_RESIDUAL_GRAD and FUSE_NORM_GRAD and hidden.is_cuda and torch.is_grad_enabled() and src.requires_grad
                 and not any(m.weight.requires_grad for m in (self.q, self.k, self.v)))
        if fused:
            # the projections that read one tensor are ONE autograd node whose dgrad GEMMs accumulate into one gradient
            # (functional.linear_acc): q | k | v of a self-attention; k | v of a cross-attention, onto K2's parked d/dsrc
            from ..functional import ResidualLink, linear_acc
            if kv is None:
                q, k, v = linear_acc(hidden, None, self.q, self.k, self.v)
            else:
                q = _linear(self.q, hidden)
                kv_link = ResidualLink() if self.attn_value_parallel_adapter is not None else None
                if k_pre is not None:
                    (k, k_slot), v = k_pre, linear_acc(src, kv_link, self.v)
                else:
                    k, v = linear_acc(src, kv_link, self.k, self.v)
                if self.attn_value_parallel_adapter is not None:
                    v = self.attn_value_parallel_adapter(src, task, y=v, link=kv_link)    # K2
        else:
            q, k, v = _linear(self.q, hidden), _linear(self.k, src), _linear(self.v, src)
            if kv is not None and self.attn_value_parallel_adapter is not None:
                v = self.attn_value_parallel_adapter(src, task, y=v)                      # K2
        if spec is not None and not EAGER_ATTENTION and self.d_kv == A.HEAD_DIM and A.supported(q, k, self.n_heads):
            fast = spec.fast()
            if fast is not None:        # on-chip kernels: bias shared by the batch + boolean key mask + causal flag; T5 has no 1/sqrt(d)
                out = A.short_attention(q, k, v, self.n_heads, fast[1], spec.causal, self.dropout, self.training, scale=1.0, bias=fast[0],
                                        k_slot=k_slot)
                return _linear(self.o, out)
        mask = spec.dense(q.dtype) if spec is not None else (None if bias is None else bias.to(q.dtype))
        out = F.scaled_dot_product_attention(self._shape(q, B), self._shape(k.contiguous(), B), self._shape(v, B), attn_mask=mask,
                                             dropout_p=self.dropout if self.training else 0.0, scale=1.0)
        return _linear(self.o, out.transpose(1, 2).reshape(B, Lq, self.inner))


class T5DenseReluDense(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.wi = nn.Linear(config.d_model, config.d_ff, bias=False)
        self.wo = nn.Linear(config.d_ff, config.d_model, bias=False)
        self.dropout = config.dropout_rate

    def forward(self, x):
        h = ffn_activation(_linear(self.wi, x), "relu", self.dropout, self.training)
        return _linear(self.wo, h)


class T5LayerSelfAttention(nn.Module):
    def __init__(self, config, is_decoder, has_relative_attention_bias):
        super().__init__()
        self.config, self.is_decoder, self.p = config, is_decoder, config.dropout_rate
        self.SelfAttention = T5Attention(config, is_decoder, has_relative_attention_bias)
        self.layer_norm = T5LayerNorm(config.d_model, eps=config.layer_norm_epsilon)
        if not is_decoder:
            build_pet(self, config, config.d_model, ("attn",))

    def forward(self, hidden, bias, task=None):
        nl = _new_norm_link(hidden)
        y = self.SelfAttention(_normed(self.layer_norm, hidden, nl), bias)
        if not self.is_decoder and has_pet(self, "attn"):                                 # K1 (x1 = un-normalised stream) + K5
            return _pet_then_tail(self, "attn", hidden, y, None, self.p, self.training, self.config, norm_link=nl)
        return _tail_linked(hidden, y, self.p, self.training, nl, layer=self)             # K5 (+ the next sublayer's norm)


class T5LayerCrossAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.p = config.dropout_rate
        self.EncDecAttention = T5Attention(config, True, False,
                                           value_adapter=bool(config.use_decoder_enc_attn_value_parallel_adapter_down_dim))
        self.layer_norm = T5LayerNorm(config.d_model, eps=config.layer_norm_epsilon)

    def forward(self, hidden, enc, bias, task=None, k_pre=None):
        nl = _new_norm_link(hidden)
        y = self.EncDecAttention(_normed(self.layer_norm, hidden, nl), bias, kv=enc, task=task, k_pre=k_pre)
        return _tail_linked(hidden, y, self.p, self.training, nl, layer=self)


class T5LayerFF(nn.Module):
    def __init__(self, config, is_decoder):
        super().__init__()
        self.config, self.is_decoder, self.p = config, is_decoder, config.dropout_rate
        self.DenseReluDense = T5DenseReluDense(config)
        self.layer_norm = T5LayerNorm(config.d_model, eps=config.layer_norm_epsilon)
        if not is_decoder:
            build_pet(self, config, config.d_model, ("ff",))

    def forward(self, hidden, task=None):
        nl = _new_norm_link(hidden)
        y = self.DenseReluDense(_normed(self.layer_norm, hidden, nl))
        if not self.is_decoder and has_pet(self, "ff"):
            return _pet_then_tail(self, "ff", hidden, y, None, self.p, self.training, self.config, norm_link=nl)      # K1 + K5
        return _tail_linked(hidden, y, self.p, self.training, nl, layer=self)


class T5Block(nn.Module):
    def __init__(self, config, is_decoder, has_relative_attention_bias):
        super().__init__()
        self.is_decoder = is_decoder
        layers = [T5LayerSelfAttention(config, is_decoder, has_relative_attention_bias)]
        if is_decoder:
            layers.append(T5LayerCrossAttention(config))
        layers.append(T5LayerFF(config, is_decoder))
        self.layer = nn.ModuleList(layers)

    def forward(self, hidden, self_bias, enc=None, cross_bias=None, task=None, k_pre=None):
        hidden = self.layer[0](hidden, self_bias, task)
        if self.is_decoder:
            hidden = self.layer[1](hidden, enc, cross_bias, task, k_pre=k_pre)
        return self.layer[-1](hidden, task)


def _pad_bias(mask, dtype):
    """[B, K] keep-mask -> additive [B, 1, 1, K] (the reference's (1 - mask) * -10000 form)."""
    return (1.0 - mask[:, None, None, :].to(torch.float32)) * -10000.0


class JointEncoder(nn.Module):
    def __init__(self, config, embed_tokens):
        super().__init__()
        self.config = config
        self.embed_tokens = embed_tokens
        self.block = nn.ModuleList([T5Block(config, False, i == 0) for i in range(config.num_layers)])
        self.final_layer_norm = T5LayerNorm(config.d_model, eps=config.layer_norm_epsilon)
        _wire_next_norms(self.block, self.final_layer_norm)
        self.p = config.dropout_rate
        vcfg = copy.copy(config)
        self.visual_embedding = VisualEmbedding(vcfg, embed_tokens, rms_norm=True)
        self.downsample = None
        if config.downsample:
            s = int(config.n_boxes ** 0.5)
            self.downsample = Downsample((s, s))

    def forward(self, input_ids, vis_inputs, attention_mask=None, task=None):
        B, L = input_ids.shape
        x = self.embed_tokens(input_ids)
        if self.downsample is not None:
            vis_inputs = self.downsample(vis_inputs, out_dtype=x.dtype)
        elif vis_inputs[0].dtype != x.dtype:
            vis_inputs = (vis_inputs[0].to(x.dtype),) + tuple(vis_inputs[1:])
        feats, boxes = vis_inputs[0], vis_inputs[1]
        img_ids = vis_inputs[2] if len(vis_inputs) >= 3 else None
        obj_ids = vis_inputs[3] if len(vis_inputs) == 4 else None
        vis = self.visual_embedding(feats, boxes, img_ids, obj_ids).to(x.dtype)             # K4
        V = vis.shape[1]
        from ..act import concat_dropout
        x = concat_dropout(x, vis, self.p, self.training)      # cat + dropout (src/modeling_t5.py:263, 300): one pass each way
        if attention_mask is None:
            attention_mask = input_ids.ne(self.config.pad_token_id)
        full = torch.cat([attention_mask.to(torch.float32), torch.ones(B, V, device=x.device)], dim=1)
        # relative position bias only between text positions (src/modeling_t5.py:311-327)
        sa0 = self.block[0].layer[0].SelfAttention
        text_bias = sa0.compute_bias(L, L)
        rel = text_bias.new_zeros(1, text_bias.shape[1], L + V, L + V)
        rel[:, :, :L, :L] = text_bias
        bias = AttnSpec(rel, full, causal=False, rel_trainable=sa0.relative_attention_bias.weight.requires_grad and torch.is_grad_enabled())
        for blk in self.block:
            x = blk(x, bias, task=task)
        x = F.dropout(self.final_layer_norm(x), p=self.p, training=self.training)
        return x, full


class T5Decoder(nn.Module):
    def __init__(self, config, embed_tokens):
        super().__init__()
        self.config = config
        self.embed_tokens = embed_tokens
        self.block = nn.ModuleList([T5Block(config, True, i == 0) for i in range(config.num_decoder_layers)])
        self.final_layer_norm = T5LayerNorm(config.d_model, eps=config.layer_norm_epsilon)
        _wire_next_norms(self.block, self.final_layer_norm)
        self.p = config.dropout_rate

    def _cross_keys_ok(self, enc, cross_bias) -> bool:
        """The blocks' cross-attention key projections as ONE GEMM (functional.cross_key_blocks; host/bart.py BartDecoder._cross_keys_ok)"""
        from .. import attention as A
        if not FUSE_CROSS_KEYS or EAGER_ATTENTION or len(self.block) < 2 or not enc.is_cuda or enc.dtype != torch.bfloat16:
            return False
        if not (FUSE_RESIDUAL_GRAD and FUSE_NORM_GRAD and torch.is_grad_enabled() and enc.requires_grad):
            return False
        atts = [blk.layer[1].EncDecAttention for blk in self.block]
        if enc.shape[1] > A.MAX_LEN or atts[0].d_kv != A.HEAD_DIM or cross_bias.fast() is None:
            return False
        return not any(m.weight.requires_grad or m.bias is not None for a in atts for m in (a.q, a.k, a.v))

    def _cross_keys(self, enc):
        mods = [blk.layer[1].EncDecAttention.k for blk in self.block]
        key = (enc.dtype, VF.FROZEN_EPOCH) + tuple((m.weight.data_ptr(), m.weight._version) for m in mods)
        c = getattr(self, "_ck_cache", None)
        if c is None or c[0] != key:
            with torch.no_grad():
                w = torch.cat([m.weight.to(enc.dtype) for m in mods], 0).contiguous()
            c = self._ck_cache = (key, w)
        return VF.cross_key_blocks(enc, c[1], None, len(mods))

    def forward(self, input_ids, enc, enc_keep, task=None):
        B, L = input_ids.shape
        x = F.dropout(self.embed_tokens(input_ids), p=self.p, training=self.training)
        sa0 = self.block[0].layer[0].SelfAttention
        self_bias = AttnSpec(sa0.compute_bias(L, L), None, causal=True,
                             rel_trainable=sa0.relative_attention_bias.weight.requires_grad and torch.is_grad_enabled())
        cross_bias = AttnSpec(None, enc_keep, causal=False)      # (dense form: invert_attention_mask, (1 - keep) * -1e9 in fp32)
        from ..functional import fanout
        n = len(self.block)
        fused_keys = self._cross_keys_ok(enc, cross_bias)
        encs = fanout(enc, n + (1 if fused_keys else 0))         # one gradient sum for the encoder output instead of autograd's pairwise adds
        ks = self._cross_keys(encs[n]) if fused_keys else None
        for i, (blk, e) in enumerate(zip(self.block, encs)):
            x = blk(x, self_bias, e, cross_bias, task, k_pre=None if ks is None else (ks[0][i], None if ks[1] is None else (ks[1], i)))
        return F.dropout(self.final_layer_norm(x), p=self.p, training=self.training)


def shift_right(labels, pad_id, start_id):
    """T5PreTrainedModel._shift_right (my_transformers/modeling_t5.py:1068-1087)."""
    out = labels.new_zeros(labels.shape)
    out[:, 1:] = labels[:, :-1]
    out[:, 0] = start_id
    return out.masked_fill(out == -100, pad_id)


class VLT5(nn.Module):
    """``forward`` returns (per-token loss [B, L], logits) like the reference's ``reduce_loss=False`` path
    (src/modeling_t5.py:670-700); the LM head is the shared embedding, inputs rescaled by d_model**-0.5."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.shared = nn.Embedding(config.vocab_size, config.d_model)
        self.encoder = JointEncoder(config, self.shared)
        self.decoder = T5Decoder(config, self.shared)
        self.apply(self._init_weights)
        self.register_load_state_dict_post_hook(lambda module, incompatible: VF.invalidate_caches())   # (see host/bart.py)

    def _init_weights(self, m):
        # my_transformers/modeling_t5.py:1026-1066: T5's Mesh-TensorFlow rules for its own layers; adapter / gate
        # Linears keep the nn.Linear default (no generic branch there)
        f = self.config.initializer_factor
        d, dkv, H = self.config.d_model, self.config.d_kv, self.config.num_heads
        if isinstance(m, T5LayerNorm):
            m.weight.data.fill_(f * 1.0)
        elif isinstance(m, VLT5):
            m.shared.weight.data.normal_(0.0, f * 1.0)
        elif isinstance(m, T5DenseReluDense):
            m.wi.weight.data.normal_(0.0, f * d ** -0.5)
            m.wo.weight.data.normal_(0.0, f * self.config.d_ff ** -0.5)
        elif isinstance(m, T5Attention):
            m.q.weight.data.normal_(0.0, f * (d * dkv) ** -0.5)
            m.k.weight.data.normal_(0.0, f * d ** -0.5)
            m.v.weight.data.normal_(0.0, f * d ** -0.5)
            m.o.weight.data.normal_(0.0, f * (H * dkv) ** -0.5)
            if m.has_relative_attention_bias:
                m.relative_attention_bias.weight.data.normal_(0.0, f * d ** -0.5)

    def forward(self, input_ids, vis_inputs, labels, task, attention_mask=None, no_padding=False):
        cfg = self.config       # (no_padding: accepted for interface symmetry with host/bart.py; the T5 bias always carries the mask)
        enc, keep = self.encoder(input_ids, vis_inputs, attention_mask, task)
        dec_in = shift_right(labels, cfg.pad_token_id, cfg.decoder_start_token_id)
        h = self.decoder(dec_in, enc, keep, task) * (cfg.d_model ** -0.5)
        return lm_loss(h, self.shared.weight, labels)
