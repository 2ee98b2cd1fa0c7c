#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz from the reference's OWN classes.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_goldens.py

Nothing of the reference is copied: the script imports its modules (through a small
import shim for the transformers-4.2.1-only symbols they expect), instantiates
``BartEncoderLayer`` / ``BartDecoderLayer`` / T5 layers / ``AdapterController`` /
``LoRALinearController`` / ``VisualEmbedding`` with the launch-script flag sets, runs
forward + backward on CPU in fp32 and stores inputs, weights, outputs and gradients.
Fixtures are data only (inputs + expected outputs).
"""
import os
import sys
import types
import copy

import numpy as np
import torch

REF = os.environ.get("VLPET_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "src")
OUT = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------- import shim
def install_shim():
    import transformers
    import transformers.file_utils as fu
    import transformers.modeling_utils as mu

    def _noop_decorator(*a, **k):
        def deco(fn):
            return fn
        return deco

    for name in ("add_code_sample_docstrings", "add_end_docstrings", "add_start_docstrings",
                 "add_start_docstrings_to_model_forward", "replace_return_docstrings"):
        setattr(fu, name, _noop_decorator)
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = lambda *a, **k: (set(), None)
    if not hasattr(mu, "prune_linear_layer"):
        mu.prune_linear_layer = lambda layer, *a, **k: layer
    mp = types.ModuleType("transformers.utils.model_parallel_utils")
    mp.assert_device_map = lambda *a, **k: None
    mp.get_device_map = lambda *a, **k: {}
    sys.modules["transformers.utils.model_parallel_utils"] = mp
    if SRC not in sys.path:
        sys.path.insert(0, SRC)


def ref_args(extra):
    """The reference's own argparse namespace for a launch-script flag set."""
    import param
    argv = sys.argv
    sys.argv = ["x"] + extra
    try:
        args = param.parse_args()
    finally:
        sys.argv = argv
    return args


VLPET_LARGE_FLAGS = [
    "--tasks", "vqa,gqa,nlvr,caption", "--use_adapter", "--use_single_adapter", "--no_encoder_adapter",
    "--use_adapter_down_dim", "--use_encoder_adapter_down_multihead", "--unfreeze_encoder_layer_norms",
    "--use_encoder_adapter_gating_large_x_lowrank", "--no_decoder_adapter",
    "--use_decoder_enc_attn_value_parallel_adapter_down_dim",
]


def make_config(kind, flags, d_model=768, heads=12, ffn=3072, dropout=0.0):
    """Mirror of TrainerBase.create_config (trainer_base.py:71-222) without from_pretrained."""
    import re
    from transformers import BartConfig, T5Config
    from adapters import AdapterConfig
    from lora import LoraConfig
    args = ref_args(flags)
    if kind == "bart":
        config = BartConfig(d_model=d_model, encoder_attention_heads=heads, decoder_attention_heads=heads,
                            encoder_ffn_dim=ffn, decoder_ffn_dim=ffn, encoder_layers=2, decoder_layers=2,
                            vocab_size=128, max_position_embeddings=64)
    else:
        config = T5Config(d_model=d_model, num_heads=heads, d_kv=d_model // heads, d_ff=ffn,
                          num_layers=2, vocab_size=128)
    for k, v in vars(args).items():
        setattr(config, k, v)
    tasks = re.split("[, ]+", args.tasks)
    config.n_images = 2
    need_adapter_cfg = args.use_adapter or args.use_decoder_enc_attn_value_parallel_adapter_down_dim
    if need_adapter_cfg:
        ac = AdapterConfig()
        ac.tasks = tasks
        ac.input_dim = config.d_model
        ac.d_model = config.d_model
        ac.use_single_adapter = args.use_single_adapter
        ac.hypercomplex_division = args.hypercomplex_division
        ac.phm_rank = args.phm_rank
        ac.shared_phm_rule = args.shared_phm_rule
        ac.factorized_phm = args.factorized_phm
        ac.low_rank_rank = args.low_rank_rank
        ac.phm_init_range = args.phm_init_range
        ac.share_down_sampler = args.share_down_sampler
        ac.share_up_sampler = args.share_up_sampler
        ac.reduction_factor = args.reduction_factor
        ac.shared_phm_rule_over_tasks = args.shared_phm_rule_over_tasks
        ac.add_layer_norm_before_adapter = args.add_layer_norm_before_adapter
        ac.add_layer_norm_after_adapter = args.add_layer_norm_after_adapter
        ac.track_z = args.track_z
        ac.use_adapter_down_dim = bool(args.use_adapter_down_dim)
        ac.adapter_down_dim = args.adapter_down_dim
        ac.use_parallel_adapter = False
        ac.use_scaling_factor = False
        ac.scaling_factor = 1.0
        config.adapter_config = ac
    else:
        config.adapter_config = None
    if args.use_lora:
        lc = LoraConfig()
        lc.lora_dim = args.lora_dim
        lc.lora_alpha = args.lora_alpha
        lc.tasks = tasks
        lc.use_single_lora = args.use_single_lora
        config.lora_config = lc
    config.dropout_rate = dropout
    config.dropout = dropout
    config.attention_dropout = dropout
    config.activation_dropout = dropout
    return config, args


def randomize(module, gen, std=0.05):
    """Give every parameter a reproducible non-trivial value (reference zero-inits some)."""
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * std)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.0f} KiB)")


def T(x):
    return x.detach().clone()


# ------------------------------------------------------------------ K1 (BART)
def capture_sublayer_io(layer, which):
    """Hook the frozen op whose output is x2 so the fixture records (x1, x2) exactly as the
    reference layer sees them."""
    store = {}
    if which == "attn":
        def hook(mod, inp, out):
            store["x2"] = out[0]
        h = layer.self_attn.register_forward_hook(hook)
    else:
        def hook(mod, inp, out):
            store["x2"] = out
        h = layer.fc2.register_forward_hook(hook)
    return store, h


def golden_k1_bart(tag, d, r, nh, rg, B, S, extra_flags=(), gate_attr="large", seed=0):
    from my_transformers.modeling_bart import BartEncoderLayer
    flags = list(VLPET_LARGE_FLAGS)
    if gate_attr != "large":
        flags.remove("--use_encoder_adapter_gating_large_x_lowrank")
    flags += ["--adapter_down_dim", str(r), "--encoder_adapter_multihead_num_head", str(nh),
              "--adapter_gating_down_dim", str(rg),
              "--decoder_enc_attn_value_parallel_adapter_down_dim", str(r)] + list(extra_flags)
    heads = 12 if d == 768 else 4
    config, args = make_config("bart", flags, d_model=d, heads=heads, ffn=4 * d)
    gen = torch.Generator().manual_seed(seed)
    layer = BartEncoderLayer(config)
    randomize(layer, gen)
    layer.eval()
    x = torch.randn(B, S, d, generator=gen)
    x.requires_grad_(True)
    store_a, ha = capture_sublayer_io(layer, "attn")
    store_f, hf = capture_sublayer_io(layer, "ff")
    # capture the FFN sublayer input (= output of self_attn_layer_norm)
    ff_in = {}
    hl = layer.self_attn_layer_norm.register_forward_hook(lambda m, i, o: ff_in.__setitem__("x1", o))
    out = layer(x, None, task="vqa")[0]
    dy = torch.randn(out.shape, generator=gen)
    store_a["x2"].retain_grad(); store_f["x2"].retain_grad(); ff_in["x1"].retain_grad()
    out.backward(dy)
    ha.remove(); hf.remove(); hl.remove()

    def stack(ml):
        return torch.cat([m.weight for m in ml], 0), torch.cat([m.bias for m in ml], 0)

    def stackg(ml):
        return torch.cat([m.weight.grad for m in ml], 0), torch.cat([m.bias.grad for m in ml], 0)

    arrs = dict(meta=np.array([d, r, nh, rg, B, S]), x=T(x), out=T(out), dy=T(dy), dx=T(x.grad),
                gating_add=np.array(int(config.use_encoder_adapter_gating_add)),
                gate_scale=np.array(float(config.encoder_gating_scaling_factor)
                                    if config.use_encoder_gating_scaling else 1.0),
                attn_x2=T(store_a["x2"]), attn_dx2=T(store_a["x2"].grad),
                ff_x1=T(ff_in["x1"]), ff_x2=T(store_f["x2"]), ff_dx2=T(store_f["x2"].grad),
                ln1_w=T(layer.self_attn_layer_norm.weight), ln1_b=T(layer.self_attn_layer_norm.bias),
                ln2_w=T(layer.final_layer_norm.weight), ln2_b=T(layer.final_layer_norm.bias),
                ln1_dw=T(layer.self_attn_layer_norm.weight.grad), ln1_db=T(layer.self_attn_layer_norm.bias.grad),
                ln2_dw=T(layer.final_layer_norm.weight.grad), ln2_db=T(layer.final_layer_norm.bias.grad))
    for pre in ("attn", "ff"):
        wd, bd = stack(getattr(layer, pre + "_adapter_multihead_down"))
        dwd, dbd = stackg(getattr(layer, pre + "_adapter_multihead_down"))
        up = getattr(layer, pre + "_adapter_multihead_up")
        arrs.update({pre + "_wd": T(wd), pre + "_bd": T(bd), pre + "_dwd": T(dwd), pre + "_dbd": T(dbd),
                     pre + "_wu": T(up.weight), pre + "_bu": T(up.bias),
                     pre + "_dwu": T(up.weight.grad), pre + "_dbu": T(up.bias.grad)})
        if gate_attr == "large":
            gd = getattr(layer, f"encoder_{pre}_adapter_gating_large_x_down")
            gu = getattr(layer, f"encoder_{pre}_adapter_gating_large_x_up")
            arrs.update({pre + "_wgd": T(gd.weight), pre + "_bgd": T(gd.bias), pre + "_dwgd": T(gd.weight.grad),
                         pre + "_dbgd": T(gd.bias.grad), pre + "_wgu": T(gu.weight), pre + "_bgu": T(gu.bias),
                         pre + "_dwgu": T(gu.weight.grad), pre + "_dbgu": T(gu.bias.grad)})
        elif gate_attr == "small":
            g = getattr(layer, f"encoder_{pre}_adapter_gating_small_xy_cat")
            arrs.update({pre + "_gw": T(g.weight), pre + "_gb": T(g.bias),
                         pre + "_dgw": T(g.weight.grad), pre + "_dgb": T(g.bias.grad)})
        elif gate_attr == "middle_x":
            g = getattr(layer, f"encoder_{pre}_adapter_gating_middle_xy_add")
            arrs.update({pre + "_gw": T(g.weight), pre + "_gb": T(g.bias),
                         pre + "_dgw": T(g.weight.grad), pre + "_dgb": T(g.bias.grad)})
        elif gate_attr == "middle_y":
            g = getattr(layer, f"encoder_{pre}_adapter_gating_middle_ia3_add")
            arrs.update({pre + "_gz": T(g), pre + "_dgz": T(g.grad)})
    # the y that enters residual+LN: recover from hooks is awkward; recompute with reference modules
    save(tag, **arrs)


# -------------------------------------------------------------------- K1 (T5)
def golden_k1_t5(tag, d, r, nh, rg, B, S, extra_flags=(), seed=1):
    from my_transformers.modeling_t5 import T5LayerFF
    flags = list(VLPET_LARGE_FLAGS) + [
        "--adapter_down_dim", str(r), "--encoder_adapter_multihead_num_head", str(nh),
        "--adapter_gating_down_dim", str(rg),
        "--decoder_enc_attn_value_parallel_adapter_down_dim", str(r)] + list(extra_flags)
    heads = 12 if d == 768 else 4
    config, args = make_config("t5", flags, d_model=d, heads=heads, ffn=4 * d)
    gen = torch.Generator().manual_seed(seed)
    layer = T5LayerFF(config, is_decoder=False)
    randomize(layer, gen)
    layer.eval()
    x = torch.randn(B, S, d, generator=gen).requires_grad_(True)
    store = {}
    h = layer.DenseReluDense.register_forward_hook(lambda m, i, o: store.__setitem__("x2", o))
    out = layer(x, None, "vqa")
    dy = torch.randn(out.shape, generator=gen)
    store["x2"].retain_grad()
    out.backward(dy)
    h.remove()
    wd = torch.cat([m.weight for m in layer.ff_adapter_multihead_down], 0)
    bd = torch.cat([m.bias for m in layer.ff_adapter_multihead_down], 0)
    dwd = torch.cat([m.weight.grad for m in layer.ff_adapter_multihead_down], 0)
    dbd = torch.cat([m.bias.grad for m in layer.ff_adapter_multihead_down], 0)
    up, gd, gu = layer.ff_adapter_multihead_up, layer.encoder_ff_adapter_gating_large_x_down, \
        layer.encoder_ff_adapter_gating_large_x_up
    save(tag, meta=np.array([d, r, nh, rg, B, S]), x=T(x), out=T(out), dy=T(dy), dx=T(x.grad),
         x2=T(store["x2"]), dx2=T(store["x2"].grad),
         delta_scale=np.array(float(config.encoder_adapter_scaling_factor) if config.use_encoder_adapter_scaling else 1.0),
         x2_scale=np.array(float(config.encoder_x2_scaling_factor) if config.use_encoder_x2_scaling else 1.0),
         gate_scale=np.array(float(config.encoder_gating_scaling_factor) if config.use_encoder_gating_scaling else 1.0),
         wd=T(wd), bd=T(bd), dwd=T(dwd), dbd=T(dbd), wu=T(up.weight), bu=T(up.bias), dwu=T(up.weight.grad),
         dbu=T(up.bias.grad), wgd=T(gd.weight), bgd=T(gd.bias), dwgd=T(gd.weight.grad), dbgd=T(gd.bias.grad),
         wgu=T(gu.weight), bgu=T(gu.bias), dwgu=T(gu.weight.grad), dbgu=T(gu.bias.grad))


# ------------------------------------------------------------------------- K2
def golden_k2(tag, d, r, B, S, scaling=None, single=True, seed=2):
    from adapters import AdapterController, AdapterConfig
    ac = AdapterConfig()
    ac.tasks = ["vqa", "gqa", "nlvr", "caption"]
    ac.input_dim = d; ac.d_model = d
    ac.use_single_adapter = single
    ac.share_down_sampler = False; ac.share_up_sampler = False
    ac.shared_phm_rule_over_tasks = False
    ac.track_z = False
    ac.use_adapter_down_dim = True; ac.adapter_down_dim = r
    ac.use_parallel_adapter = True
    ac.use_scaling_factor = scaling is not None
    ac.scaling_factor = scaling if scaling is not None else 1.0
    gen = torch.Generator().manual_seed(seed)
    ctl = AdapterController(ac)
    randomize(ctl, gen)
    x = torch.randn(B, S, d, generator=gen).requires_grad_(True)
    y = torch.randn(B, S, d, generator=gen).requires_grad_(True)
    task = "gqa"
    out = ctl(x, task, y=y)
    dy = torch.randn(out.shape, generator=gen)
    out.backward(dy)
    ad = ctl.adapters[task]
    keys = sorted(ctl.state_dict().keys())
    save(tag, meta=np.array([d, r, B, S]), scaling=np.array(-1.0 if scaling is None else scaling),
         x=T(x), y=T(y), out=T(out), dy=T(dy), dx=T(x.grad), dyin=T(y.grad),
         wd=T(ad.down_sampler.weight), bd=T(ad.down_sampler.bias), wu=T(ad.up_sampler.weight), bu=T(ad.up_sampler.bias),
         dwd=T(ad.down_sampler.weight.grad), dbd=T(ad.down_sampler.bias.grad),
         dwu=T(ad.up_sampler.weight.grad), dbu=T(ad.up_sampler.bias.grad),
         state_keys=np.array(keys))


# ------------------------------------------------------------------------- K3
def golden_k3(tag, d, r, alpha, M, single=True, seed=3):
    from lora import LoRALinearController, LoraConfig
    lc = LoraConfig()
    lc.lora_dim = r; lc.lora_alpha = alpha
    lc.tasks = ["vqa", "gqa", "nlvr", "caption"]; lc.use_single_lora = single
    gen = torch.Generator().manual_seed(seed)
    lin = LoRALinearController(d, d, config=lc, bias=True)
    randomize(lin, gen)
    lin.eval()   # eval == train with p=0 (lora/controller.py:65-68)
    x = torch.randn(M, d, generator=gen).requires_grad_(True)
    task = "nlvr"
    out = lin(x, task)
    dy = torch.randn(out.shape, generator=gen)
    out.backward(dy)
    keys = sorted(lin.state_dict().keys())
    save(tag, meta=np.array([d, r, alpha, M]), scaling=np.array(lin.scaling), x=T(x), out=T(out), dy=T(dy),
         dx=T(x.grad), w=T(lin.weight), b=T(lin.bias), a=T(lin.lora_As[task]), bb=T(lin.lora_Bs[task]),
         da=T(lin.lora_As[task].grad), dbb=T(lin.lora_Bs[task].grad), dbias=T(lin.bias.grad),
         state_keys=np.array(keys))


# ------------------------------------------------------------------------- K4
def golden_k4(tag, kind, d, feat_dim, B, N, nlvr_ids=False, seed=4):
    """VisualEmbedding from src/modeling_bart.py / src/modeling_t5.py.  Those files import the
    whole VL wrapper stack; only the class is needed, so the module is executed with the few
    extra HF-4.2.1 names stubbed."""
    import transformers
    import torch.nn as nn
    flags = list(VLPET_LARGE_FLAGS) + ["--feat_dim", str(feat_dim)]
    config, args = make_config(kind, flags, d_model=d, heads=4 if d != 768 else 12, ffn=4 * d)
    config.feat_dim = int(feat_dim); config.pos_dim = 4
    config.vis_use_transformer = False
    config.use_vis_order_embedding = True
    config.use_vis_layer_norm = True
    config.individual_vis_layer_norm = True
    config.default_obj_order_ids = None
    config.additional_visual_embedding_layers = 0
    mod = load_vl_module(kind)
    gen = torch.Generator().manual_seed(seed)
    table = nn.Embedding(200, d)
    ve = mod.VisualEmbedding(config, table)
    randomize(ve, gen, std=0.05)
    with torch.no_grad():
        for m in ve.modules():
            if isinstance(m, nn.LayerNorm) or type(m).__name__ == "T5LayerNorm":
                m.weight.add_(1.0)
    feats = torch.randn(B, N, int(feat_dim), generator=gen)
    pos = torch.rand(B, N, 4, generator=gen)
    img_ids = obj_ids = None
    if nlvr_ids:
        half = N // 2
        img_ids = torch.cat([torch.zeros(half, dtype=torch.long), torch.ones(N - half, dtype=torch.long)]).unsqueeze(0)
        obj_ids = torch.cat([torch.arange(half), torch.arange(N - half)]).unsqueeze(0)
    out = ve(feats, pos, img_ids, obj_ids)
    dy = torch.randn(out.shape, generator=gen)
    out.backward(dy)
    fe, pe = ve.feat_embedding, ve.absolute_vis_pos_embedding
    rms = (kind == "t5")   # T5LayerNorm: no bias, no mean subtraction (my_transformers/modeling_t5.py:235-252)
    zeros = torch.zeros(d)

    def b(ln, grad=False):
        if rms:
            return zeros
        return ln.bias.grad if grad else ln.bias
    save(tag, meta=np.array([d, int(feat_dim), B, N]), rms=np.array(int(rms)),
         eps=np.array(float(config.layer_norm_epsilon) if rms else 1e-5),
         feats=T(feats), pos=T(pos), out=T(out), dy=T(dy),
         img_ids=(img_ids if img_ids is not None else np.zeros((0,), np.int64)),
         obj_ids=(obj_ids if obj_ids is not None else np.zeros((0,), np.int64)),
         feat_w=T(fe[0].weight), feat_b=T(fe[0].bias), feat_ln_w=T(fe[1].weight), feat_ln_b=T(b(fe[1])),
         pos_w=T(pe[0].weight), pos_b=T(pe[0].bias), pos_ln_w=T(pe[1].weight), pos_ln_b=T(b(pe[1])),
         img_table=T(ve.img_order_embedding.weight), obj_table=T(table.weight),
         d_feat_w=T(fe[0].weight.grad), d_feat_b=T(fe[0].bias.grad), d_feat_ln_w=T(fe[1].weight.grad),
         d_feat_ln_b=T(b(fe[1], True)), d_pos_w=T(pe[0].weight.grad), d_pos_b=T(pe[0].bias.grad),
         d_pos_ln_w=T(pe[1].weight.grad), d_pos_ln_b=T(b(pe[1], True)),
         d_img_table=T(ve.img_order_embedding.weight.grad), d_obj_table=T(table.weight.grad),
         state_keys=np.array(sorted(ve.state_dict().keys())))


_VL = {}


def load_vl_module(kind):
    if kind in _VL:
        return _VL[kind]
    import transformers
    import my_transformers.modeling_bart as mb
    import transformers.models.bart.modeling_bart as hb
    for n in ("_make_causal_mask", "_expand_mask"):
        if not hasattr(hb, n):
            setattr(hb, n, getattr(mb, n))
    import importlib
    name = "modeling_bart" if kind == "bart" else "modeling_t5"
    try:
        mod = importlib.import_module(name)
    except ImportError:
        class _Dummy:  # noqa
            pass
        for n in ("BeamScorer", "BeamSearchScorer"):
            if not hasattr(sys.modules["transformers"], n):
                setattr(sys.modules["transformers"], n, _Dummy)
        mod = importlib.import_module(name)
    _VL[kind] = mod
    return mod


# ------------------------------------------------ decoder layer (hook placement)
def golden_decoder_layer(tag, d, r, B, S_enc, S_dec, seed=5):
    from my_transformers.modeling_bart import BartDecoderLayer
    flags = list(VLPET_LARGE_FLAGS) + ["--adapter_down_dim", str(r), "--encoder_adapter_multihead_num_head", "4",
                                       "--adapter_gating_down_dim", str(r),
                                       "--decoder_enc_attn_value_parallel_adapter_down_dim", str(r)]
    config, args = make_config("bart", flags, d_model=d, heads=4, ffn=4 * d)
    gen = torch.Generator().manual_seed(seed)
    layer = BartDecoderLayer(config)
    randomize(layer, gen)
    layer.eval()
    hid = torch.randn(B, S_dec, d, generator=gen)
    enc = torch.randn(B, S_enc, d, generator=gen).requires_grad_(True)
    # causal self-attention mask exactly as BartDecoder.forward builds it (my_transformers/modeling_bart.py:93-106)
    from my_transformers.modeling_bart import _make_causal_mask
    causal = _make_causal_mask((B, S_dec), hid.dtype)
    out = layer(hid, attention_mask=causal, encoder_hidden_states=enc, task="vqa")[0]
    dy = torch.randn(out.shape, generator=gen)
    out.backward(dy)
    sd = {k: T(v) for k, v in layer.state_dict().items()}
    grads = {"grad::" + n: T(p.grad) for n, p in layer.named_parameters() if p.grad is not None and "adapter" in n}
    save(tag, meta=np.array([d, r, B, S_enc, S_dec]), hid=T(hid), enc=T(enc), out=T(out), dy=T(dy),
         denc=T(enc.grad), **{"sd::" + k: v for k, v in sd.items()}, **grads)


# ------------------------------------------------------------------ Downsample
def golden_downsample(tag, B=2, dim=64, seed=6):
    """Downsample (src/modeling_bart.py:556-613): AdaptiveMaxPool2d 7x7 -> 6x6 on [B, 49, dim] and the NLVR
    two-image form on [B, 98, dim] with its box / id cropping."""
    mod = load_vl_module("bart")
    ds = mod.Downsample((6, 6))
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 49, dim, generator=gen)
    boxes = torch.rand(B, 49, 4, generator=gen)
    y, yb = ds((x, boxes))
    x2 = torch.randn(B, 98, dim, generator=gen)
    boxes2 = torch.rand(B, 98, 4, generator=gen)
    img_ids = torch.cat([torch.zeros(B, 49), torch.ones(B, 49)], 1).long()
    obj_ids = torch.cat([torch.arange(49), torch.arange(49)]).unsqueeze(0).expand(B, -1).contiguous()
    y2, yb2, yi2, yo2 = ds((x2, boxes2, img_ids, obj_ids))
    save(tag, x=T(x), boxes=T(boxes), y=T(y), yb=T(yb), x2=T(x2), boxes2=T(boxes2), img_ids=img_ids.numpy(),
         obj_ids=obj_ids.numpy(), y2=T(y2), yb2=T(yb2), yi2=yi2.numpy(), yo2=yo2.numpy())


# ------------------------------------------------------------------ whole tiny VLBart (host + trainer pin)
LORA_FLAGS = ["--tasks", "vqa,gqa,nlvr,caption", "--use_lora", "--lora_dim", "8", "--use_single_lora"]   # single_lora.sh:49-56


def install_t5_runtime_shim():
    """HF 4.2.1 ``ModuleUtilsMixin`` helpers the reference's T5 stack calls with their old signatures
    (get_extended_attention_mask(mask, shape, device), get_head_mask, invert_attention_mask), restated on the
    harness side: additive masks (1 - m) * -10000 (self) / -1e9 (cross, fp32), causal for the decoder."""
    from transformers import PreTrainedModel

    def gext(self, attention_mask, input_shape, device=None, dtype=None):
        if attention_mask.dim() == 3:
            ext = attention_mask[:, None, :, :]
        elif self.config.is_decoder:
            b, L = input_shape
            ids = torch.arange(L, device=attention_mask.device)
            causal = (ids[None, None, :].repeat(b, L, 1) <= ids[None, :, None]).to(attention_mask.dtype)
            ext = causal[:, None, :, :] * attention_mask[:, None, None, :]
        else:
            ext = attention_mask[:, None, None, :]
        return (1.0 - ext.to(torch.float32)) * -10000.0

    def inv(self, m):
        e = m[:, None, None, :] if m.dim() == 2 else m[:, None, :, :]
        return (1.0 - e.to(torch.float32)) * -1e9
    PreTrainedModel.get_extended_attention_mask = gext
    PreTrainedModel.get_head_mask = lambda self, head_mask, n, is_attention_chunked=False: [None] * n
    PreTrainedModel.invert_attention_mask = inv


T5_VLPET_FLAGS = ["--tasks", "vqa,gqa,nlvr,caption", "--use_adapter", "--use_single_adapter", "--no_encoder_adapter",
                  "--no_decoder_adapter", "--use_adapter_down_dim", "--use_encoder_adapter_down_multihead",
                  "--unfreeze_encoder_layer_norms", "--use_encoder_adapter_gating_large_x_lowrank",
                  "--use_decoder_enc_attn_value_parallel_adapter_down_dim",
                  "--use_encoder_gating_scaling", "--encoder_gating_scaling_factor", "0.3"]   # T5-VL-PET-large.sh:41-59


def golden_vlbart_tiny(tag="vlbart_tiny_d64", seed=7, lora=False, kind="bart", video=False):
    """2+2-layer, d=64 ``VLBart`` built from the reference's own classes (src/modeling_bart.py:1458-1530 over
    JointEncoder :690-1010 and my_transformers BartDecoder): state dict, three task batches, eval-mode per-token
    losses + logits (pins the host: [text ; visual] concat order, text-only LayerNorm before the concat, hook
    placement, Downsample, label shifting), and the losses of 5 training steps (pins the trainer: loss reduction
    per task vqa_model.py:216-227 / nlvr_model.py / caption_model.py, clip_grad_norm_ 5.0, AdamW, linear warm-up,
    which parameters train).  transformers 4.2.1's AdamW is not importable here, so the optimizer update is the
    oracle's restatement (oracle.hf_adamw_step); everything else is reference code."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import vlpet_oracle as O
    mod = load_vl_module(kind)
    if lora:
        flags = list(LORA_FLAGS) + ["--downsample", "--n_boxes", "36"]
    else:
        base = VLPET_LARGE_FLAGS if kind == "bart" else T5_VLPET_FLAGS
        flags = list(base) + ["--adapter_down_dim", "8", "--encoder_adapter_multihead_num_head", "4",
                              "--adapter_gating_down_dim", "16",
                              "--decoder_enc_attn_value_parallel_adapter_down_dim", "8",
                              "--downsample", "--n_boxes", "64" if video else "36"]
        if video:   # scripts/video-text/VL-PET-large.sh:50-52: four video tasks, 64 frame features, Downsample((8, 8))
            flags[flags.index("--tasks") + 1] = "tvqa,how2qa,tvc,yc2c"
    config, args = make_config(kind, flags, d_model=64, heads=4, ffn=128)
    if kind == "t5":
        install_t5_runtime_shim()
        config.decoder_start_token_id = 0
        config.pad_token_id = 0
    if lora:
        config.lora_config.lora_dropout = 0.0      # LoraConfig's default 0.1 would make the captured steps random
    config.vocab_size = 500
    config.feat_dim = 128
    config.default_obj_order_ids = list(range(400, 500))
    config.encoder_prompt_config = None
    config.decoder_prompt_config = None
    from transformers import PreTrainedModel
    PreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)
    torch.manual_seed(seed)
    model = mod.VLBart(config) if kind == "bart" else mod.VLT5(config)
    gen = torch.Generator().manual_seed(seed)
    randomize(model, gen)
    model.lm_head.weight = model.model.shared.weight if kind == "bart" else model.shared.weight   # tied head
    # trainable set: TrainerBase.unfreeze_parameters' substring rules (trainer_base.py:308-542) for this flag set
    for n, p in model.named_parameters():
        if lora:    # trainer_base.py:339-344: lora matrices and every bias; visual embedding stays trainable (:318-322)
            p.requires_grad = ("lora" in n) or ("bias" in n) or ("visual_embedding" in n)
        else:
            p.requires_grad = ("adapter" in n) or ("gating" in n) or ("visual_embedding" in n) or \
                ("encoder." in n and ("layer_norm" in n or "layernorm" in n))
    B = 3
    V = 300

    def batch(task, L, T):
        ids = torch.randint(5, V, (B, L), generator=gen)
        labels = torch.randint(5, V, (B, T), generator=gen)
        if task == "nlvr":
            feats = torch.randn(B, 98, 128, generator=gen)
            boxes = torch.zeros(B, 98, 4)
            img = torch.cat([torch.zeros(49), torch.ones(49)]).long().unsqueeze(0).expand(B, -1).contiguous()
            obj = torch.cat([torch.arange(49), torch.arange(49)]).unsqueeze(0).expand(B, -1).contiguous()
            vis = (feats, boxes, img, obj)
        else:
            vis = (torch.randn(B, 49, 128, generator=gen), torch.zeros(B, 49, 4))
        scores = torch.rand(B, generator=gen) * 0.5 + 0.5
        return dict(task=task, ids=ids, labels=labels, vis=vis, scores=scores)
    batches = [batch("vqa", 20, 5), batch("nlvr", 20, 2), batch("caption", 12, 9)]
    if video:
        # video/video_model.py:34-46: vis_inputs = (frame features [B, 64, feat_dim], zero boxes); ragged text padded with
        # the pad id (BART: 1), so the default attention mask input_ids.ne(pad) (src/modeling_bart.py:817-818) matters
        def vbatch(task, L, T, lens):
            b = batch("vqa", L, T)
            b["task"] = task
            b["vis"] = (torch.randn(B, 64, 128, generator=gen), torch.zeros(B, 64, 4))
            for i, n in enumerate(lens):
                b["ids"][i, n:] = config.pad_token_id
            return b
        batches = [vbatch("tvqa", 30, 3, (30, 17, 9)), vbatch("tvc", 24, 9, (11, 24, 20)), vbatch("how2qa", 40, 3, (40, 40, 25))]

    def run(b):
        out = model(input_ids=b["ids"], vis_inputs=b["vis"], labels=b["labels"], return_dict=True, task=b["task"])
        return out["loss"].view(b["labels"].shape), out["logits"]

    def reduce(b, per):
        mask = (b["labels"] != -100).float()
        if b["task"] in ("vqa", "gqa"):                      # vqa_model.py:216-227
            return ((per * mask).sum(1) / mask.sum(1).clamp(min=1) * b["scores"]).mean()
        if video:                                            # video/video_model.py:77-87: no score weighting
            return ((per * mask).sum(1) / mask.sum(1).clamp(min=1)).mean()
        return (per * mask).sum() / mask.sum().clamp(min=1)   # reduce_loss=True: token mean

    arrs = {"sd::" + k: T(v) for k, v in model.state_dict().items()}
    model.eval()
    with torch.no_grad():
        for i, b in enumerate(batches):
            per, logits = run(b)
            arrs.update({f"b{i}::task": np.array(b["task"]), f"b{i}::ids": b["ids"].numpy(), f"b{i}::labels": b["labels"].numpy(),
                         f"b{i}::scores": T(b["scores"]), f"b{i}::per_token": T(per), f"b{i}::logits": T(logits),
                         f"b{i}::loss": T(reduce(b, per))})
            for j, v in enumerate(b["vis"]):
                arrs[f"b{i}::vis{j}"] = v.numpy()
    # ---- 5 training steps (multitask.py:217-342): batches cycled, lr 5e-3, 10 total steps, 10 % warm-up
    model.train()
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    state = {n: (torch.zeros_like(p), torch.zeros_like(p)) for n, p in named}
    losses = []
    total, warm, base_lr = 10, 1, 5e-3
    for step in range(5):
        b = batches[step % 3]
        for _, p in named:
            p.grad = None
        per, _ = run(b)
        loss = reduce(b, per)
        loss.backward()
        losses.append(float(loss))
        torch.nn.utils.clip_grad_norm_([p for _, p in named], 5.0)
        lr = O.linear_warmup_lr(step, base_lr, warm, total)
        for n, p in named:
            wd = 0.0 if any(nd in n for nd in ("bias", "LayerNorm.weight")) else 0.01
            m, v = state[n]
            if p.grad is None:
                continue
            O.hf_adamw_step(p.data, p.grad, m, v, step + 1, lr, eps=1e-6, weight_decay=wd)
    arrs["train::losses"] = np.array(losses, dtype=np.float64)
    arrs["train::hparams"] = np.array([base_lr, total, warm, 5.0])
    for n, p in named:
        arrs["final::" + n] = T(p.data)
    save(tag, **arrs)


# ------------------------------------------------------------------ LowRankVisualEmbedding
def golden_lowrank_vis(tag, gated, d=64, feat_dim=128, r=16, nh=4, rg=8, B=2, N=6, seed=10, residual=False):
    """LowRankVisualEmbedding (src/modeling_bart.py:195-334), plain, with the low-rank gate, and with the gate in its
    ``fe + fe * gate`` form (``--use_visual_projector_residual_connection``, :292-293)."""
    import torch.nn as nn
    flags = list(VLPET_LARGE_FLAGS) + ["--feat_dim", str(feat_dim), "--visual_projector_down_dim", str(r),
                                       "--visual_projector_multihead_num_head", str(nh),
                                       "--visual_projector_gating_down_dim", str(rg)]
    if gated:
        flags.append("--use_visual_projector_gating_large_x_lowrank")
    if residual:
        flags.append("--use_visual_projector_residual_connection")
    config, args = make_config("bart", flags, d_model=d, heads=4, ffn=4 * d)
    config.feat_dim = int(feat_dim); config.pos_dim = 4
    config.vis_use_transformer = False
    config.use_vis_order_embedding = True
    config.use_vis_layer_norm = True
    config.individual_vis_layer_norm = True
    config.default_obj_order_ids = None
    mod = load_vl_module("bart")
    gen = torch.Generator().manual_seed(seed)
    table = nn.Embedding(200, d)
    ve = mod.LowRankVisualEmbedding(config, table)
    randomize(ve, gen, std=0.05)
    with torch.no_grad():
        for m in ve.modules():
            if isinstance(m, nn.LayerNorm):
                m.weight.add_(1.0)
    feats = torch.randn(B, N, int(feat_dim), generator=gen)
    pos = torch.rand(B, N, 4, generator=gen)
    out = ve(feats, pos)
    dy = torch.randn(out.shape, generator=gen)
    out.backward(dy)
    arrs = {"sd::" + k: T(v) for k, v in ve.state_dict().items()}
    arrs.update({"grad::" + n: T(p.grad) for n, p in ve.named_parameters() if p.grad is not None})
    save(tag, meta=np.array([d, feat_dim, r, nh, rg, B, N, int(gated)] + ([1] if residual else [])), feats=T(feats), pos=T(pos),
         out=T(out), dy=T(dy), **arrs)


# -------------------------------------------------------- trainable-name lists
# ------------------------------------------------------------------------- eager fallbacks ("else eager fallback")
def golden_fallbacks():
    """Reference outputs for the configurations the fused kernels do not cover and vl-pet_amd/eager.py runs as plain torch ops:
    an Adapter with another non-linearity + track_z, the low-rank adapter, a rectangular and a fan_in_fan_out LoRA layer, the
    VisualEmbedding with ONE LayerNorm over the sum."""
    import torch.nn as nn
    from adapters import AdapterController, AdapterConfig
    from lora import LoRALinearController, LoraConfig
    d, r, B, S = 64, 8, 2, 5
    for tag, low_rank in (("fb_adapter_relu_trackz_d64_r8", False), ("fb_lowrank_adapter_d64", True)):
        ac = AdapterConfig()
        ac.tasks = ["vqa", "gqa"]; ac.input_dim = d; ac.d_model = d
        ac.use_single_adapter = True; ac.share_down_sampler = False; ac.share_up_sampler = False
        ac.shared_phm_rule_over_tasks = False
        ac.track_z = True
        ac.non_linearity = "relu"
        ac.use_adapter_down_dim = True; ac.adapter_down_dim = r; ac.reduction_factor = 8
        ac.use_parallel_adapter = True; ac.use_scaling_factor = True; ac.scaling_factor = 2.0
        ac.low_rank_adapters = low_rank; ac.low_rank_rank = 2; ac.low_rank_w_init = "glorot-uniform"
        gen = torch.Generator().manual_seed(20 + int(low_rank))
        ctl = AdapterController(ac)
        randomize(ctl, gen, std=0.2)
        x = torch.randn(B, S, d, generator=gen).requires_grad_(True)
        y = torch.randn(B, S, d, generator=gen).requires_grad_(True)
        out = ctl(x, "gqa", y=y)
        dy = torch.randn(out.shape, generator=gen)
        out.backward(dy)
        ad = ctl.adapters["gqa"]
        params = {k.replace(".", "__"): T(v) for k, v in ad.state_dict().items()}
        grads = {"g__" + k.replace(".", "__"): T(v.grad) for k, v in ad.named_parameters()}
        save(tag, meta=np.array([d, r, B, S]), x=T(x), y=T(y), out=T(out), dy=T(dy), dx=T(x.grad), dyin=T(y.grad), z=T(ad.z),
             state_keys=np.array(sorted(ctl.state_dict().keys())), **params, **grads)
    for tag, din, dout, fifo in (("fb_lora_rect_64x48_r4", 64, 48, False), ("fb_lora_fanin_64_r4", 64, 64, True)):
        lc = LoraConfig()
        lc.lora_dim = 4; lc.lora_alpha = 32; lc.tasks = ["vqa", "nlvr"]; lc.use_single_lora = True
        gen = torch.Generator().manual_seed(22 + int(fifo))
        lin = LoRALinearController(din, dout, fan_in_fan_out=fifo, config=lc, bias=True)
        randomize(lin, gen)
        lin.eval()
        x = torch.randn(7, din, generator=gen).requires_grad_(True)
        out = lin(x, "nlvr")
        dy = torch.randn(out.shape, generator=gen)
        out.backward(dy)
        save(tag, meta=np.array([din, dout, 4, 32, int(fifo)]), x=T(x), out=T(out), dy=T(dy), dx=T(x.grad), w=T(lin.weight), b=T(lin.bias),
             a=T(lin.lora_As["nlvr"]), bb=T(lin.lora_Bs["nlvr"]), da=T(lin.lora_As["nlvr"].grad), dbb=T(lin.lora_Bs["nlvr"].grad),
             dbias=T(lin.bias.grad))
    # VisualEmbedding, one LayerNorm over the sum (use_vis_layer_norm, not individual_vis_layer_norm)
    feat_dim, N = 128, 6
    flags = list(VLPET_LARGE_FLAGS) + ["--feat_dim", str(feat_dim)]
    config, args = make_config("bart", flags, d_model=d, heads=4, ffn=4 * d)
    config.feat_dim = feat_dim; config.pos_dim = 4; config.vis_use_transformer = False
    config.use_vis_order_embedding = True; config.use_vis_layer_norm = True; config.individual_vis_layer_norm = False
    config.default_obj_order_ids = None; config.additional_visual_embedding_layers = 0
    mod = load_vl_module("bart")
    gen = torch.Generator().manual_seed(24)
    table = nn.Embedding(200, d)
    ve = mod.VisualEmbedding(config, table)
    randomize(ve, gen, std=0.05)
    with torch.no_grad():
        ve.layer_norm.weight.add_(1.0)
    feats = torch.randn(B, N, feat_dim, generator=gen); pos = torch.rand(B, N, 4, generator=gen)
    out = ve(feats, pos)
    dy = torch.randn(out.shape, generator=gen)
    out.backward(dy)
    params = {k.replace(".", "__"): T(v) for k, v in ve.state_dict().items()}
    grads = {"g__" + k.replace(".", "__"): T(v.grad) for k, v in ve.named_parameters() if v.grad is not None}
    save("fb_visemb_sharedln_d64_f128", meta=np.array([d, feat_dim, B, N]), feats=T(feats), pos=T(pos), out=T(out), dy=T(dy),
         obj_table=T(table.weight), **params, **grads)


def golden_trainable_names():
    """Parameter-name lists + trainable flags for the VL-PET-large BART encoder/decoder layer
    under TrainerBase.unfreeze_parameters' substring rules (trainer_base.py:308-542).  The rule
    restated: name contains 'adapter' or 'gating' or 'visual_embedding', or (encoder + layer_norm)."""
    from my_transformers.modeling_bart import BartEncoderLayer, BartDecoderLayer
    flags = list(VLPET_LARGE_FLAGS) + ["--adapter_down_dim", "96", "--encoder_adapter_multihead_num_head", "4",
                                       "--adapter_gating_down_dim", "96",
                                       "--decoder_enc_attn_value_parallel_adapter_down_dim", "96"]
    config, args = make_config("bart", flags)
    enc = BartEncoderLayer(config)
    dec = BartDecoderLayer(config)
    enc_names = [(n, int(p.numel())) for n, p in enc.named_parameters()]
    dec_names = [(n, int(p.numel())) for n, p in dec.named_parameters()]
    save("names_bart_vlpet_large",
         enc_names=np.array([n for n, _ in enc_names]), enc_numel=np.array([k for _, k in enc_names]),
         dec_names=np.array([n for n, _ in dec_names]), dec_numel=np.array([k for _, k in dec_names]))


def main():
    install_shim()
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == "downsample":      # add one fixture without regenerating the rest
        golden_downsample("downsample_7to6_d64")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "lowrank":
        golden_lowrank_vis("lowrank_vis_d64", gated=False)
        golden_lowrank_vis("lowrank_vis_gated_d64", gated=True)
        # the fe + fe * gate form; a second bottleneck geometry (r = 24 over 3 heads, r_g = 32, 3 x 11 rows)
        golden_lowrank_vis("lowrank_vis_gated_res_d64", gated=True, residual=True, r=24, nh=3, rg=32, B=3, N=11, seed=12)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fallbacks":
        golden_fallbacks()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "video":
        golden_vlbart_tiny("vlbart_tiny_video_d64", seed=11, video=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "vlbart":
        golden_vlbart_tiny()
        golden_vlbart_tiny("vlbart_tiny_lora_d64", seed=8, lora=True)
        golden_vlbart_tiny("vlt5_tiny_d64", seed=9, kind="t5")
        return
    # (i) K1 BART, full width and tiny, gate variants
    golden_k1_bart("k1_bart_large_d768_r96", 768, 96, 4, 96, B=2, S=8)
    golden_k1_bart("k1_bart_large_d64_r8", 64, 8, 4, 8, B=2, S=5)
    golden_k1_bart("k1_bart_large_add_d64_r8", 64, 8, 4, 8, B=2, S=5,
                   extra_flags=["--use_encoder_adapter_gating_add"])
    golden_k1_bart("k1_bart_large_scale_d64_r16", 64, 16, 4, 8, B=2, S=5,
                   extra_flags=["--use_encoder_gating_scaling", "--encoder_gating_scaling_factor", "0.3"])
    golden_k1_bart("k1_bart_small_d64_r8", 64, 8, 4, 8, B=2, S=5, gate_attr="small",
                   extra_flags=["--use_encoder_adapter_gating_small_xy_cat"])
    golden_k1_bart("k1_bart_middlex_d64_r8", 64, 8, 4, 8, B=2, S=5, gate_attr="middle_x",
                   extra_flags=["--use_encoder_adapter_gating_middle_xy_add"])
    golden_k1_bart("k1_bart_middley_d64_r8", 64, 8, 4, 8, B=2, S=5, gate_attr="middle_y",
                   extra_flags=["--use_encoder_adapter_gating_middle_ia3_add"])
    # (ii) K1 T5 incl. scalings and r=192
    golden_k1_t5("k1_t5_d128_r192", 128, 192, 4, 192, B=1, S=6,
                 extra_flags=["--use_encoder_gating_scaling", "--encoder_gating_scaling_factor", "0.3"])
    golden_k1_t5("k1_t5_scaled_d64_r16", 64, 16, 4, 16, B=2, S=5,
                 extra_flags=["--use_encoder_adapter_scaling", "--encoder_adapter_scaling_factor", "4.0",
                              "--use_encoder_x2_scaling", "--encoder_x2_scaling_factor", "0.5",
                              "--use_encoder_gating_scaling", "--encoder_gating_scaling_factor", "0.3"])
    # (iii) K2
    golden_k2("k2_d768_r96", 768, 96, B=2, S=7)
    golden_k2("k2_scaled_d64_r8", 64, 8, B=2, S=5, scaling=4.0, single=False)
    # (iv) K3
    golden_k3("k3_d256_r8", 256, 8, 32, M=12)
    golden_k3("k3_d256_r64", 256, 64, 32, M=12)
    golden_k3("k3_d64_r4", 64, 4, 32, M=9, single=False)
    golden_k3("k3_d128_r128", 128, 128, 32, M=9)
    # (v) K4
    golden_k4("k4_bart_d64_f128", "bart", 64, 128, B=2, N=6)
    golden_k4("k4_bart_nlvr_d64_f128", "bart", 64, 128, B=2, N=6, nlvr_ids=True)
    golden_k4("k4_bart_d128_f256", "bart", 128, 256, B=1, N=4)
    golden_k4("k4_t5_d64_f128", "t5", 64, 128, B=2, N=6)
    # (vi) decoder layer hook placement, (vii) names
    golden_decoder_layer("dec_layer_d64_r8", 64, 8, B=2, S_enc=6, S_dec=3)
    golden_trainable_names()
    golden_downsample("downsample_7to6_d64")
    golden_vlbart_tiny()
    golden_vlbart_tiny("vlbart_tiny_lora_d64", seed=8, lora=True)
    golden_vlbart_tiny("vlt5_tiny_d64", seed=9, kind="t5")
    golden_lowrank_vis("lowrank_vis_d64", gated=False)
    golden_lowrank_vis("lowrank_vis_gated_d64", gated=True)
    # the fe + fe * gate form; a second bottleneck geometry (r = 24 over 3 heads, r_g = 32, 3 x 11 rows)
    golden_lowrank_vis("lowrank_vis_gated_res_d64", gated=True, residual=True, r=24, nh=3, rg=32, B=3, N=11, seed=12)
    golden_vlbart_tiny("vlbart_tiny_video_d64", seed=11, video=True)


if __name__ == "__main__":
    main()
