"""Visual front end: CLIP-grid down-sampling and the visual-feature projection.

``VisualEmbedding`` keeps the reference's constructor, forward signature and state-dict keys
(src/modeling_bart.py:77-192; T5 variant src/modeling_t5.py:44-174):

    LN(Linear(feat_dim -> d)(feats)) + LN(Linear(5 -> d)([box, area]))
        + img_order_embedding[img_ids] + obj_order_embedding[V - 1 - obj_ids]

``Downsample`` is the AdaptiveMaxPool2d 7x7 -> 6x6 of src/modeling_bart.py:556-613 (frozen, no
parameters), a HIP gather kernel with the fp32 -> compute-dtype cast fused.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


EAGER_RMS_NORM = False              # A/B switch (tools/ab_switches.py): library-op T5LayerNorm


class T5LayerNorm(nn.Module):
    """RMS norm without bias (my_transformers/modeling_t5.py:235-252)."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x, link=None):
        """``link`` (not in the reference's signature): see vlpet_amd.tail.rms_norm; ignored on the eager path."""
        fused = getattr(x, "_vlpet_norm", None)         # the tail that produced x already normalised it for this norm (tail.sublayer_tail_rms)
        if fused is not None and fused.weight is self.weight:
            return fused.normed
        if x.is_cuda and not EAGER_RMS_NORM and x.shape[-1] % 8 == 0 and x.dtype in (torch.bfloat16, torch.float32):
            from .tail import rms_norm
            return rms_norm(x, self.weight, self.variance_epsilon, link)  # one HIP pass each way (csrc/tail.hip, rms mode)
        # eager form (CPU tensors of the parity harness; VLPET_EAGER_RMS_NORM=1 for A/B): statistics and scaling in fp32, result in the activation dtype (the reference casts the normalised rows to the
        # weight's half dtype, :248-251; here the weight may be an fp32 master next to bf16 activations -- a bf16 row times an
        # fp32 tensor silently promoted the whole encoder to fp32 before this line said otherwise)
        xf = x.to(torch.float32)
        var = xf.pow(2).mean(-1, keepdim=True)
        y = self.weight.to(torch.float32) * (xf * torch.rsqrt(var + self.variance_epsilon))
        return y.to(x.dtype)


class Downsample(nn.Module):
    """AdaptiveMaxPool2d over the CLIP token grid (src/modeling_bart.py:556-613), on the HIP kernel
    (csrc/downsample.hip) with the cast to the compute dtype fused: ``out_dtype`` (default: the input's)."""

    def __init__(self, output_size):
        super().__init__()
        self.output_size = output_size

    def downsample_inputs(self, x, out_dtype=None):
        from . import _lib
        from .functional import _io_dtype, _need_cuda, _stream
        _need_cuda(x)
        B, L, dim = x.shape
        s = int(L ** 0.5)
        if s * s != L or self.output_size[0] != self.output_size[1]:
            raise ValueError("Downsample expects a square token grid and a square output size")
        so = int(self.output_size[0])
        x = x if x.is_contiguous() else x.contiguous()
        out = torch.empty(B, so * so, dim, dtype=out_dtype or x.dtype, device=x.device)
        lib = _lib.load()
        rc = lib.vlpet_downsample_fwd(x.data_ptr(), out.data_ptr(), B, s, so, dim, _io_dtype(x), _io_dtype(out), _stream())
        _lib.check(rc, "vlpet_downsample_fwd")
        return out

    def forward(self, inputs_tuple, out_dtype=None):
        if len(inputs_tuple) == 4:   # NLVR: two images side by side along the token axis = 2B independent grids
            x, boxes, img_ids, obj_ids = inputs_tuple
            B, L2, dim = x.shape
            y = self.downsample_inputs(x.reshape(2 * B, L2 // 2, dim), out_dtype)
            half = y.shape[1]
            y = y.reshape(B, 2 * half, dim)

            def crop(t):
                t = torch.cat(torch.chunk(t, 2, 1), 0)[:, :half]
                return torch.cat(torch.chunk(t, 2, 0), 1)
            return y, crop(boxes), crop(img_ids), crop(obj_ids)
        x, boxes = inputs_tuple
        x = self.downsample_inputs(x, out_dtype)
        return x, boxes[:, :x.shape[1]]


def _order_lookup(table: nn.Embedding, ids: torch.Tensor) -> torch.Tensor:
    """``table(ids)`` for the image-order embedding (src/modeling_bart.py:170-171: ``n_images`` rows, trainable).  NLVR hands over one id
    per visual token of every sample ([B, 72] = 11,952 indices of 2 distinct values); torch's embedding backward then takes its sort-based
    path (radix sort + unique-by-key + segmented sums, ~15 launches), and THAT sequence does not survive hipGraph replay on this stack:
    the second replay of a captured NLVR step, once another shape had been captured after it, died with
    HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION (found in round 6 with tools/lora_graph_probe.py: full-batch BART and LoRA alike, gone with
    this form).  A table of a few rows is a [n_ids, n_rows] one-hot GEMM both ways instead -- two small library GEMMs, no sort."""
    if table.num_embeddings <= 8 and ids.numel() > 3072 and table.weight.requires_grad:
        onehot = F.one_hot(ids.reshape(-1), table.num_embeddings).to(table.weight.dtype)
        return (onehot @ table.weight).view(*ids.shape, table.embedding_dim)
    return table(ids)


def _position_terms(mod, pos, img_order_ids, obj_order_ids, per_branch: bool, B: int, N: int):
    """R = [LN](Linear(5 -> d)([box, area])) + the two order embeddings, fp32, with library ops (src/modeling_bart.py:129-141,162-183):
    the form for whatever csrc/vispos.hip does not cover (other widths, a trainable object-order table, more than four image-order rows)."""
    pos = pos.float()
    pos5 = torch.cat([pos, mod.get_area(pos).unsqueeze(2)], dim=2)
    pl = mod.absolute_vis_pos_embedding[0]
    R = F.linear(pos5, pl.weight.float(), pl.bias.float())
    if per_branch:
        R = mod.absolute_vis_pos_embedding[1](R).float()
    if mod.config.use_vis_order_embedding:
        dev = pos.device
        if img_order_ids is None:
            img_order_ids = torch.zeros(N, dtype=torch.long, device=dev).unsqueeze(0)
        if obj_order_ids is None:
            obj_order_ids = torch.arange(N, dtype=torch.long, device=dev).unsqueeze(0)
        obj_order_ids = mod.obj_order_embedding.num_embeddings - obj_order_ids - 1
        R = R + _order_lookup(mod.img_order_embedding, img_order_ids).float() + mod.obj_order_embedding(obj_order_ids).float()
    return R.expand(B, N, R.shape[-1])


def _position_terms_fused(mod, pos, img_order_ids, obj_order_ids, per_branch: bool, out_dtype):
    """The same R in ``out_dtype`` through the HIP kernel, or None (visproj.position_terms says where it applies)."""
    if not per_branch or not pos.is_cuda:
        return None
    from .visproj import position_terms
    order = bool(mod.config.use_vis_order_embedding)
    return position_terms(pos, mod.absolute_vis_pos_embedding[0], mod.absolute_vis_pos_embedding[1],
                          mod.img_order_embedding if order else None, mod.obj_order_embedding if order else None,
                          img_order_ids, obj_order_ids, isinstance(mod.absolute_vis_pos_embedding[1], T5LayerNorm), out_dtype)


class VisualEmbedding(nn.Module):
    def __init__(self, config, obj_order_embedding: nn.Embedding, rms_norm: bool = False):
        super().__init__()
        self.config = config
        d = config.d_model
        feat_dim, pos_dim = int(config.feat_dim), int(config.pos_dim)
        self.rms_norm = rms_norm
        norm = (lambda: T5LayerNorm(d, eps=getattr(config, "layer_norm_epsilon", 1e-6))) if rms_norm \
            else (lambda: nn.LayerNorm(d))
        individual = config.use_vis_layer_norm and config.individual_vis_layer_norm
        fe = [nn.Linear(feat_dim, d)]
        if individual:
            fe.append(norm())
        self.feat_embedding = nn.Sequential(*fe)
        pe = [nn.Linear(pos_dim + 1, d)]
        if individual:
            pe.append(norm())
        self.absolute_vis_pos_embedding = nn.Sequential(*pe)
        if config.use_vis_order_embedding:
            self.obj_order_embedding = obj_order_embedding
            self.img_order_embedding = nn.Embedding(config.n_images, d)
        if config.use_vis_layer_norm and not config.individual_vis_layer_norm:
            self.layer_norm = norm()

    @staticmethod
    def get_area(pos):
        return (pos[:, :, 3] - pos[:, :, 2]) * (pos[:, :, 1] - pos[:, :, 0])

    def forward(self, feats, pos, img_order_ids=None, obj_order_ids=None):
        """K4.  The feature projection (Linear(feat_dim -> d) + LayerNorm, the K = feat_dim contraction) runs in
        the fused HIP kernel, which also adds R = position branch + order embeddings after the norm; R itself is
        a 5-wide linear + norm + two gathers (host ops)."""
        B, N, _ = feats.shape
        assert pos.shape == (B, N, 4)
        per_branch = bool(self.config.use_vis_layer_norm and self.config.individual_vis_layer_norm)
        R = _position_terms_fused(self, pos, img_order_ids, obj_order_ids, per_branch, feats.dtype)    # one launch each way (csrc/vispos.hip)
        if R is None:
            R = _position_terms(self, pos, img_order_ids, obj_order_ids, per_branch, B, N)
        if not per_branch:
            # no LayerNorm behind the feature projection (use_vis_layer_norm off, or ONE norm over the sum: src/modeling_bart.py:186-188):
            # not the fused kernel's shape -- plain torch ops ("else eager fallback"); no launch script sets these flags
            from . import eager
            return eager.visual_embedding(feats, R, self.feat_embedding, getattr(self, "layer_norm", None)).to(feats.dtype)
        from .visproj import VisProjPackCache, visproj
        if not hasattr(self, "_vis_cache"):
            self._vis_cache = VisProjPackCache()
        return visproj(feats, R, self.feat_embedding[0], self.feat_embedding[1], self._vis_cache, self.rms_norm)


class LowRankVisualEmbedding(nn.Module):
    """Low-rank visual projector (src/modeling_bart.py:195-334): ``LN(up(gelu_new(cat_i down_i(feats))))`` with an
    optional low-rank sigmoid gate on the features, plus the position / order-embedding terms of ``VisualEmbedding``.
    Same constructor, forward signature and state-dict keys as the reference.

    No launch script of the reference enables it (``--use_lowrank_visual_projector`` is an ablation flag).  The feature
    branch -- both low-rank chains over the feat_dim-wide features, the gate product, LayerNorm and the post-norm add of
    the position / order-embedding term -- runs on the HIP path (vl-pet_amd/lowrank.py: the K1 kernels in their
    rectangular form, the weight-gradient kernels, the K5 kernel with the residual after the norm), plain and gated,
    for bottlenecks up to 96 and feat_dim / d_model multiples of 64.  The 5-wide position branch and the two embedding
    lookups stay library ops, as in ``VisualEmbedding``."""

    def __init__(self, config, obj_order_embedding: nn.Embedding):
        super().__init__()
        from .activations import get_activation
        self.config = config
        d, feat_dim, pos_dim = config.d_model, int(config.feat_dim), int(config.pos_dim)
        nh, r = int(config.visual_projector_multihead_num_head), int(config.visual_projector_down_dim)
        self.visual_projector_multihead_dim = int(r / nh)
        self.visual_projector_multihead_down = nn.ModuleList([nn.Linear(feat_dim, int(r / nh)) for _ in range(nh)])
        self.visual_projector_multihead_up = nn.Linear(r, d)
        self.visual_projector_non_linear = get_activation("gelu_new")
        if getattr(config, "use_visual_projector_gating_large_x_lowrank", False):
            rg = int(config.visual_projector_gating_down_dim)
            self.visual_projector_gating_large_x_down = nn.Linear(feat_dim, rg)
            self.visual_projector_gating_large_x_up = nn.Linear(rg, d)
            self.gating_non_linear = get_activation("gelu_new")
        self._per_branch = bool(config.use_vis_layer_norm and config.individual_vis_layer_norm)
        if self._per_branch:
            self.visual_projector_layer_norm = nn.LayerNorm(d)
            self.absolute_vis_pos_embedding = nn.Sequential(nn.Linear(pos_dim + 1, d), nn.LayerNorm(d))
        else:       # (src/modeling_bart.py:231-252: bare projections, one LayerNorm over the sum if use_vis_layer_norm)
            self.absolute_vis_pos_embedding = nn.Sequential(nn.Linear(pos_dim + 1, d))
            if config.use_vis_layer_norm:
                self.layer_norm = nn.LayerNorm(d)
        if config.use_vis_order_embedding:
            self.obj_order_embedding = obj_order_embedding
            self.img_order_embedding = nn.Embedding(config.n_images, d)

    get_area = staticmethod(VisualEmbedding.get_area)

    def forward(self, feats, pos, img_order_ids=None, obj_order_ids=None):
        B, N, _ = feats.shape
        assert pos.shape == (B, N, 4)
        gated = hasattr(self, "visual_projector_gating_large_x_down")
        r_max = max(self.visual_projector_multihead_up.weight.shape[1],
                    self.visual_projector_gating_large_x_down.weight.shape[0] if gated else 0)
        plain = not self._per_branch or r_max > 96
        R = None if plain else _position_terms_fused(self, pos, img_order_ids, obj_order_ids, True, feats.dtype)
        if R is None:
            R = _position_terms(self, pos, img_order_ids, obj_order_ids, self._per_branch, B, N)
        if plain:
            # bottlenecks above 96 or no per-branch LayerNorm: outside the rectangular kernels -- plain torch ops
            from . import eager
            fe = eager.lowrank_visual_features(
                feats, list(self.visual_projector_multihead_down), self.visual_projector_multihead_up, self.visual_projector_non_linear,
                self.visual_projector_gating_large_x_down if gated else None, self.visual_projector_gating_large_x_up if gated else None,
                getattr(self, "gating_non_linear", None), bool(getattr(self.config, "use_visual_projector_residual_connection", False)),
                self.visual_projector_layer_norm if self._per_branch else None)
            v = fe.to(R.dtype) + R
            if not self._per_branch and hasattr(self, "layer_norm"):
                v = self.layer_norm(v)
            return v.to(feats.dtype)
        from .lowrank import LowRankPackCache, lowrank_project
        if not hasattr(self, "_lr_caches"):
            self._lr_caches = (LowRankPackCache(), LowRankPackCache())
        return lowrank_project(
            feats, R, list(self.visual_projector_multihead_down), self.visual_projector_multihead_up,
            self.visual_projector_layer_norm,
            self.visual_projector_gating_large_x_down if gated else None,
            self.visual_projector_gating_large_x_up if gated else None,
            bool(getattr(self.config, "use_visual_projector_residual_connection", False)),
            self._lr_caches[0], self._lr_caches[1])
