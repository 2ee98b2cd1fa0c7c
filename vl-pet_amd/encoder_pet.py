"""Encoder granularity-controlled adapter ("VL-PET") as a mixin for transformer blocks.

The reference has no class for it: the parameters hang directly off ``BartEncoderLayer``
(my_transformers/modeling_bart.py:1000-1005,1044-1056) resp. ``T5LayerSelfAttention`` / ``T5LayerFF``
(my_transformers/modeling_t5.py:706-724,312-330) and ~60 lines of inline tensor code use them
(modeling_bart.py:1147-1155,1195-1209,1256-1257).  ``build_pet`` creates attributes with exactly
those names (so state-dict keys, the 'adapter' / 'gating' freeze substrings and the zero-init rules
of trainer_base.py:497-533,557-599 keep working) and ``apply_pet`` replaces the inline code with
one fused HIP kernel call (large gate) or the adapter-only kernel plus the row kernels of ``gates``
(small / middleX / middleY, modeling_bart.py:1210-1231).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as VF
from . import gates
from .activations import get_activation

GRANULARITY_FLAGS = ("use_encoder_adapter_gating_large_x_lowrank", "use_encoder_adapter_gating_small_xy_cat",
                     "use_encoder_adapter_gating_middle_xy_add", "use_encoder_adapter_gating_middle_ia3_add")


def _names(which: str):
    return dict(down=f"{which}_adapter_multihead_down", up=f"{which}_adapter_multihead_up",
                gdown=f"encoder_{which}_adapter_gating_large_x_down", gup=f"encoder_{which}_adapter_gating_large_x_up",
                small=f"encoder_{which}_adapter_gating_small_xy_cat", midx=f"encoder_{which}_adapter_gating_middle_xy_add",
                midy=f"encoder_{which}_adapter_gating_middle_ia3_add")


def build_pet(module: nn.Module, config, embed_dim: int, which=("attn", "ff")):
    """Create the multi-head down / up projections and the low-rank gate on ``module``."""
    module.adapter_non_linear = get_activation("gelu_new")
    module.gating_non_linear = get_activation("gelu_new")
    if not hasattr(module, "_pet_caches"):
        module._pet_caches = {}
    for w in which:
        n = _names(w)
        if getattr(config, "use_encoder_adapter_down_multihead", False):
            nh = int(config.encoder_adapter_multihead_num_head)
            r = int(config.adapter_down_dim)
            module.encoder_adapter_multihead_dim = int(r / nh)
            setattr(module, n["down"], nn.ModuleList([nn.Linear(embed_dim, int(r / nh)) for _ in range(nh)]))
            setattr(module, n["up"], nn.Linear(r, embed_dim))
        else:
            setattr(module, n["down"], None)
            setattr(module, n["up"], None)
        if getattr(config, "use_encoder_adapter_gating_large_x_lowrank", False):
            rg = int(config.adapter_gating_down_dim)
            setattr(module, n["gdown"], nn.Linear(embed_dim, rg))
            setattr(module, n["gup"], nn.Linear(rg, embed_dim))
        else:
            setattr(module, n["gdown"], None)
            setattr(module, n["gup"], None)
        # the three other granularity gates (my_transformers/modeling_bart.py:976-998): same attribute names
        setattr(module, n["small"], nn.Linear(2 * embed_dim, 1)
                if getattr(config, "use_encoder_adapter_gating_small_xy_cat", False) else None)
        setattr(module, n["midx"], nn.Linear(embed_dim, 1)
                if getattr(config, "use_encoder_adapter_gating_middle_xy_add", False) else None)
        if getattr(config, "use_encoder_adapter_gating_middle_ia3_add", False):
            setattr(module, n["midy"], nn.Parameter(torch.zeros(embed_dim)))
        else:
            setattr(module, n["midy"], None)
        module._pet_caches[w] = (VF.PackCache(), VF.PackCache())


# r, r_g in (96, 192] (the T5 script) run on the fused 6-tile kernels: two-chain forward (pet_gate_fwd.hip, 64-row
# workgroups) with saved activations and the two-pass backward (pet_gate_bwd3.hip).  T5 bench, same box: 2,121 samples/s
# fused vs 1,989 through the composition of 3-tile kernels below (profiles/r02_bench_t5_*.json.log).  The composition
# (exact: the bottleneck splits into two halves whose up projections add,  lin = s2*x2 + sd*(D1 + D2),  G = G1 + G2,
# y = lin (*|+) sigmoid(G) * gs) stays for A/B (VLPET_SPLIT_WIDE=1) and as the reference of test_apply_pet_wide_bottleneck.
SPLIT_WIDE_BOTTLENECK = False        # A/B switch (set by tools / tests, tools/ab_switches.py): r = 192 as a composition of 3-tile kernels


def _apply_pet_split(module, which, x1, x2, dws, dbs, up, gdown, gup, mode, sd, s2, gs, io):
    caches = module._pet_caches.setdefault(which + "/split", tuple(VF.PackCache() for _ in range(4)))
    nh = len(dws)
    r = up.weight.shape[1]
    if nh % 2 == 0:
        wa, ba, wb, bb = list(dws[:nh // 2]), list(dbs[:nh // 2]), list(dws[nh // 2:]), list(dbs[nh // 2:])
        ra = sum(w.shape[0] for w in wa)
    elif nh == 1:                           # a single down projection: split its rows
        ra = r // 2
        wa, ba, wb, bb = [dws[0][:ra]], [dbs[0][:ra]], [dws[0][ra:]], [dbs[0][ra:]]
    else:       # an odd number (> 1) of heads does not split into two 3-tile halves: plain torch ops (vl-pet_amd/eager.py)
        from . import eager
        from .activations import get_activation
        act = get_activation("gelu_new")
        gp = None if mode == VF.GATE_NONE else (gdown.weight, gdown.bias, gup.weight, gup.bias)
        return eager.adapter_gate(x1, x2, list(dws), list(dbs), up.weight, up.bias, gp, act, act,
                                  {VF.GATE_MUL: "mul", VF.GATE_ADD: "add", VF.GATE_NONE: "none"}[mode], sd, s2, gs)
    cat = lambda ts: ts[0] if len(ts) == 1 else torch.cat(list(ts), 0)
    # Every pack below is keyed on the SOURCE parameters (PackCache.get_derived): the slices / concatenations / contiguous
    # copies are temporaries, rebuilt only when a source changed.
    src_a = list(dws) + list(dbs) + [up.weight, up.bias]
    t3 = 3
    # adapter chain: h1 = s2*x2 + sd*(up_a(gelu(down_a x2)) + bu);  lin = h1 + sd*up_b(gelu(down_b x2))
    ua = up.weight[:, :ra]
    pk = caches[0].get_derived(src_a, ("a0", ra), lambda: (wa, ba, ua.contiguous(), up.bias), io, t3)
    h1 = VF.adapter_gate(None, x2, wa, ba, ua, up.bias, None, pk, None, VF.GATE_NONE, sd, s2, 1.0)
    wbc, bbc, ub = cat(wb), cat(bb), up.weight[:, ra:]
    pk = caches[1].get_derived(src_a, ("a1", ra), lambda: ([wbc], [bbc], ub.contiguous(), None), io, t3)
    lin = VF.parallel_adapter(x2, h1, wbc, bbc, ub, None, pk, sd)
    # gate chain on x1
    rg = gup.weight.shape[1]
    ga = rg // 2
    src_g = [gdown.weight, gdown.bias, gup.weight, gup.bias]
    gwa, gba, gwb, gbb = gdown.weight[:ga], gdown.bias[:ga], gdown.weight[ga:], gdown.bias[ga:]
    gua, gub = gup.weight[:, :ga], gup.weight[:, ga:]
    pk = caches[2].get_derived(src_g, ("g0", ga), lambda: ([gwa], [gba], gua.contiguous(), gup.bias), io, t3)
    g1 = VF.adapter_gate(None, x1, [gwa], [gba], gua, gup.bias, None, pk, None, VF.GATE_NONE, 1.0, 0.0, 1.0)
    pk = caches[3].get_derived(src_g, ("g1", ga), lambda: ([gwb], [gbb], gub.contiguous(), None), io, t3)
    g = torch.sigmoid(VF.parallel_adapter(x1, g1, gwb, gbb, gub, None, pk, 1.0).float()).to(lin.dtype)
    y = lin + g if mode == VF.GATE_ADD else lin * g
    return y * gs if gs != 1.0 else y


def has_pet(module: nn.Module, which: str) -> bool:
    return getattr(module, _names(which)["down"], None) is not None


def apply_pet(module: nn.Module, which: str, x1: torch.Tensor, x2: torch.Tensor, config, link=None, out_link=None) -> torch.Tensor:
    """y = ((x2*s2 + sd*up(gelu_new(cat_i down_i(x2)))) (*|+) sigmoid(up_g(gelu_new(down_g(x1))))) * gs

    x1 = sublayer input ("residual"), x2 = frozen attention / FFN output; the caller then does
    ``LayerNorm(x1 + dropout(y))`` (BART) or ``x1 + dropout(y)`` (T5)."""
    n = _names(which)
    downs = getattr(module, n["down"])
    up = getattr(module, n["up"])
    gdown, gup = getattr(module, n["gdown"]), getattr(module, n["gup"])
    io = VF._io_dtype(x2)
    r = up.weight.shape[1]
    gate = gdown is not None
    tiles = VF.rank_tiles(r)
    if gate:
        tiles = max(tiles, VF.rank_tiles(gdown.weight.shape[0]))
    ca, cg = module._pet_caches[which]
    dws = [m.weight for m in downs]
    dbs = [m.bias for m in downs]
    pk_a = ca.get(dws, dbs, up.weight, up.bias, io, tiles)
    pk_g = cg.get([gdown.weight], [gdown.bias], gup.weight, gup.bias, io, tiles) if gate else None
    mode = VF.GATE_NONE
    if gate:
        mode = VF.GATE_ADD if getattr(config, "use_encoder_adapter_gating_add", False) else VF.GATE_MUL
    sd = float(config.encoder_adapter_scaling_factor) if getattr(config, "use_encoder_adapter_scaling", False) else 1.0
    s2 = float(config.encoder_x2_scaling_factor) if getattr(config, "use_encoder_x2_scaling", False) else 1.0
    gs = float(config.encoder_gating_scaling_factor) if getattr(config, "use_encoder_gating_scaling", False) else 1.0
    gp = (gdown.weight, gdown.bias, gup.weight, gup.bias) if gate else None
    if x1.dtype != x2.dtype:
        x1 = x1.to(x2.dtype)
    if gate and tiles == 6 and SPLIT_WIDE_BOTTLENECK:
        return _apply_pet_split(module, which, x1, x2, dws, dbs, up, gdown, gup, mode, sd, s2, gs, io)
    y = VF.adapter_gate(x1, x2, dws, dbs, up.weight, up.bias, gp, pk_a, pk_g, mode, sd, s2, gs if gate else 1.0, link=link,
                        out_link=out_link if x1.dim() == x2.dim() else None)
    if gate:
        return y
    # same precedence as the reference's elif chain (large > small > middleX > middleY)
    add = bool(getattr(config, "use_encoder_adapter_gating_add", False))
    small, midx, midy = getattr(module, n["small"], None), getattr(module, n["midx"], None), getattr(module, n["midy"], None)
    if small is not None:
        return gates.small_gate(x1, y, small, add, gs)
    if midx is not None:
        return gates.middle_x_gate(x1, y, midx, add, gs)
    if midy is not None:
        return gates.middle_y_gate(y, midy, add, gs)
    if gs != 1.0:
        y = y * gs
    return y
