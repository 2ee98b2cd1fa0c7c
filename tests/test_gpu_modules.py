"""GPU: golden fixtures (made by the reference's own classes) through the HIP path, and the drop-in
modules / host layers against them.  fp32 IO, tolerance 1e-3 (BASELINE.json)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from gpu_cases import rel_err  # noqa: E402

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-3


def load(name):
    z = np.load(os.path.join(G, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind == "f" else z[k]) for k in z.files}


def cuda(t):
    return t.detach().clone().cuda().requires_grad_(True)


def ok(a, ref, what):
    e = rel_err(a, ref)
    assert e <= TOL, f"{what}: rel err {e:.3e}"


@pytest.mark.parametrize("name,add", [("k1_bart_large_d768_r96", False), ("k1_bart_large_d64_r8", False),
                                      ("k1_bart_large_add_d64_r8", True), ("k1_bart_large_scale_d64_r16", False)])
def test_k1_bart_golden(name, add):
    import vlpet_amd.functional as F
    g = load(name)
    d, r, nh, rg, B, S = [int(v) for v in g["meta"]]
    rh = r // nh
    x1, x2 = cuda(g["ff_x1"]), cuda(g["ff_x2"])
    dws = [cuda(g["ff_wd"][i * rh:(i + 1) * rh]) for i in range(nh)]
    dbs = [cuda(g["ff_bd"][i * rh:(i + 1) * rh]) for i in range(nh)]
    wu, bu, wgd, bgd, wgu, bgu = (cuda(g["ff_" + k]) for k in ("wu", "bu", "wgd", "bgd", "wgu", "bgu"))
    lw, lb = cuda(g["ln2_w"]), cuda(g["ln2_b"])
    tiles = max(F.rank_tiles(r), F.rank_tiles(rg))
    pa = F.pack_pair(dws, dbs, wu, bu, 0, tiles)
    pg = F.pack_pair([wgd], [bgd], wgu, bgu, 0, tiles)
    y = F.adapter_gate(x1, x2, dws, dbs, wu, bu, (wgd, bgd, wgu, bgu), pa, pg,
                       F.GATE_ADD if add else F.GATE_MUL, 1.0, 1.0, float(g["gate_scale"]))
    out = torch.nn.functional.layer_norm(x1 + y, (d,), lw, lb)          # BART tail, dropout off
    ok(out, g["out"], "layer output")
    out.backward(g["dy"].cuda())
    ok(x2.grad, g["ff_dx2"], "dx2")
    ok(torch.cat([w.grad for w in dws]), g["ff_dwd"], "dwd")
    ok(torch.cat([b.grad for b in dbs]), g["ff_dbd"], "dbd")
    for t, k in ((wu, "dwu"), (bu, "dbu"), (wgd, "dwgd"), (bgd, "dbgd"), (wgu, "dwgu"), (bgu, "dbgu")):
        ok(t.grad, g["ff_" + k], k)
    ok(lw.grad, g["ln2_dw"], "ln dw")


@pytest.mark.parametrize("name", ["k1_t5_d128_r192", "k1_t5_scaled_d64_r16"])
def test_k1_t5_golden(name):
    import vlpet_amd.functional as F
    g = load(name)
    d, r, nh, rg, B, S = [int(v) for v in g["meta"]]
    rh = r // nh
    x1, x2 = cuda(g["x"]), cuda(g["x2"])
    dws = [cuda(g["wd"][i * rh:(i + 1) * rh]) for i in range(nh)]
    dbs = [cuda(g["bd"][i * rh:(i + 1) * rh]) for i in range(nh)]
    wu, bu, wgd, bgd, wgu, bgu = (cuda(g[k]) for k in ("wu", "bu", "wgd", "bgd", "wgu", "bgu"))
    tiles = max(F.rank_tiles(r), F.rank_tiles(rg))
    pa = F.pack_pair(dws, dbs, wu, bu, 0, tiles)
    pg = F.pack_pair([wgd], [bgd], wgu, bgu, 0, tiles)
    y = F.adapter_gate(x1, x2, dws, dbs, wu, bu, (wgd, bgd, wgu, bgu), pa, pg, F.GATE_MUL,
                       float(g["delta_scale"]), float(g["x2_scale"]), float(g["gate_scale"]))
    out = x1 + y                                                         # T5 tail
    ok(out, g["out"], "layer output")
    out.backward(g["dy"].cuda())
    ok(x2.grad, g["dx2"], "dx2")
    ok(torch.cat([w.grad for w in dws]), g["dwd"], "dwd")
    for t, k in ((wu, "dwu"), (bu, "dbu"), (wgd, "dwgd"), (bgd, "dbgd"), (wgu, "dwgu"), (bgu, "dbgu")):
        ok(t.grad, g[k], k)


@pytest.mark.parametrize("name", ["k2_d768_r96", "k2_scaled_d64_r8"])
def test_k2_adapter_controller_golden(name):
    from vlpet_amd.adapters import AdapterConfig, AdapterController
    g = load(name)
    d, r, B, S = [int(v) for v in g["meta"]]
    sc = float(g["scaling"])
    cfg = AdapterConfig(tasks=["vqa", "gqa", "nlvr", "caption"], d_model=d, input_dim=d, use_single_adapter=sc < 0,
                        use_adapter_down_dim=True, adapter_down_dim=r, use_parallel_adapter=True,
                        use_scaling_factor=sc >= 0, scaling_factor=max(sc, 1.0))
    ctl = AdapterController(cfg).cuda()
    ad = ctl.adapters["gqa"]
    with torch.no_grad():
        ad.down_sampler.weight.copy_(g["wd"]); ad.down_sampler.bias.copy_(g["bd"])
        ad.up_sampler.weight.copy_(g["wu"]); ad.up_sampler.bias.copy_(g["bu"])
    x, y = cuda(g["x"]), cuda(g["y"])
    out = ctl(x, "gqa", y=y)
    ok(out, g["out"], "out")
    out.backward(g["dy"].cuda())
    ok(x.grad, g["dx"], "dx"); ok(y.grad, g["dyin"], "dy")
    ok(ad.down_sampler.weight.grad, g["dwd"], "dwd"); ok(ad.down_sampler.bias.grad, g["dbd"], "dbd")
    ok(ad.up_sampler.weight.grad, g["dwu"], "dwu"); ok(ad.up_sampler.bias.grad, g["dbu"], "dbu")


@pytest.mark.parametrize("name", ["k3_d256_r8", "k3_d256_r64", "k3_d64_r4", "k3_d128_r128"])
def test_k3_lora_controller_golden(name):
    from vlpet_amd.lora import LoraConfig, LoRALinearController
    g = load(name)
    d, r, alpha, M = [int(v) for v in g["meta"]]
    lin = LoRALinearController(d, d, config=LoraConfig(lora_dim=r, lora_alpha=alpha,
                                                       tasks=["vqa", "gqa", "nlvr", "caption"]), bias=True).cuda()
    with torch.no_grad():
        lin.weight.copy_(g["w"]); lin.bias.copy_(g["b"])
        lin.lora_As["nlvr"].copy_(g["a"]); lin.lora_Bs["nlvr"].copy_(g["bb"])
    lin.eval()
    x = cuda(g["x"])
    out = lin(x, "nlvr")
    ok(out, g["out"], "out")
    out.backward(g["dy"].cuda())
    ok(x.grad, g["dx"], "dx")
    ok(lin.lora_As["nlvr"].grad, g["da"], "dA"); ok(lin.lora_Bs["nlvr"].grad, g["dbb"], "dB")
    ok(lin.bias.grad, g["dbias"], "dbias")


def test_decoder_layer_hook_placement_golden():
    """The reference BartDecoderLayer's state dict loads into the host layer by name and the fused
    value-parallel adapter sits where the reference applies it (my_transformers/modeling_bart.py:427-430)."""
    import vlpet_amd.host.bart as HB
    g = load("dec_layer_d64_r8")
    d, r, B, S_enc, S_dec = [int(v) for v in g["meta"]]
    cfg = HB.vlpet_config(d_model=d, decoder_attention_heads=4, encoder_attention_heads=4, decoder_ffn_dim=4 * d,
                          encoder_ffn_dim=4 * d, adapter_down_dim=r, adapter_gating_down_dim=r,
                          decoder_enc_attn_value_parallel_adapter_down_dim=r, dropout=0.0, attention_dropout=0.0,
                          activation_dropout=0.0)
    layer = HB.BartDecoderLayer(cfg)
    sd = {k[4:]: v for k, v in g.items() if k.startswith("sd::")}
    missing, unexpected = layer.load_state_dict(sd, strict=True)
    layer = layer.cuda().eval()
    enc = cuda(g["enc"])
    out = layer(g["hid"].cuda(), enc, None, "vqa")
    ok(out, g["out"], "decoder layer out")
    out.backward(g["dy"].cuda())
    ok(enc.grad, g["denc"], "d enc")
    for k, v in g.items():
        if k.startswith("grad::"):
            p = dict(layer.named_parameters())[k[6:]]
            ok(p.grad, v, k)


def test_tiny_host_model_fused_equals_eager():
    """Tiny 2+2-layer BART host: loss and every trainable gradient from the HIP path (GPU) equal the same
    model with the PET ops routed to the oracle on CPU."""
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    from vlpet_amd.adapters.adapter_modeling import Adapter
    from oracle import vlpet_oracle as O
    cfg = HB.vlpet_config(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                          decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, vocab_size=300 + 200,
                          max_position_embeddings=64, feat_dim=128, adapter_down_dim=8, adapter_gating_down_dim=16,
                          decoder_enc_attn_value_parallel_adapter_down_dim=8, dropout=0.0, attention_dropout=0.0,
                          activation_dropout=0.0)
    torch.manual_seed(0)
    model = HB.VLBart(cfg)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    TR.trainable_names(model, cfg)
    gen = torch.Generator().manual_seed(5)
    batch = TR.synthetic_batch("vqa", 6, cfg, "cpu", gen)

    def loss_of(m, b):
        per, _ = m(b["input_ids"], b["vis_inputs"], b["labels"], "vqa")
        return TR.task_loss(per, b["labels"], b["scores"], "vqa")

    # eager CPU (checker)
    from oracle.host_patch import cpu_reference_ops
    with cpu_reference_ops():
        model.eval()
        l_ref = loss_of(model, batch)
        l_ref.backward()
        g_ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.requires_grad}
        model.zero_grad()
    model.cuda()
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    b["vis_inputs"] = tuple(t.cuda() for t in batch["vis_inputs"])
    l = loss_of(model, b)
    l.backward()
    assert abs(float(l) - float(l_ref)) <= 1e-3 * abs(float(l_ref))
    for n, p in model.named_parameters():
        if p.requires_grad:
            ok(p.grad, g_ref[n], n)


def test_empty_batches_need_no_launch():
    """Zero rows (an empty task shard): every op returns an empty result of the right shape and the parameters
    still receive (zero) gradients -- the reference's op chain behaves the same on an empty batch."""
    import torch.nn as nn
    from types import SimpleNamespace
    from vlpet_amd.encoder_pet import apply_pet, build_pet
    from vlpet_amd.tail import sublayer_tail
    from vlpet_amd.adapters import AdapterConfig, AdapterController
    cfg = SimpleNamespace(use_encoder_adapter_down_multihead=True, encoder_adapter_multihead_num_head=4, adapter_down_dim=8,
                          use_encoder_adapter_gating_large_x_lowrank=True, adapter_gating_down_dim=16)
    m = nn.Module(); build_pet(m, cfg, 64, ("attn",)); m.ln = nn.LayerNorm(64); m.cuda()
    x1 = torch.zeros(0, 7, 64, device="cuda", requires_grad=True)
    x2 = torch.zeros(0, 7, 64, device="cuda", requires_grad=True)
    y = apply_pet(m, "attn", x1, x2, cfg)
    out = sublayer_tail(x1, y, m.ln, 0.1, True)
    assert out.shape == (0, 7, 64)
    out.sum().backward()
    assert m.attn_adapter_multihead_up.weight.grad is not None and float(m.attn_adapter_multihead_up.weight.grad.abs().sum()) == 0.0
    assert m.ln.weight.grad is not None


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("add,gs", [(False, 0.3), (True, 1.0)])
def test_apply_pet_wide_bottleneck(dtype, add, gs):
    """r = r_g = 192 (T5-VL-PET-large.sh): apply_pet runs the layer as a composition of 3-tile kernels (two bottleneck
    halves); forward and every gradient against the oracle, and against the fused 6-tile kernels."""
    import torch.nn as nn
    from types import SimpleNamespace
    import vlpet_amd.encoder_pet as EP
    from oracle import vlpet_oracle as O
    torch.manual_seed(3)
    B, S, d, r, nh = 5, 56, 768, 192, 4
    cfg = SimpleNamespace(use_encoder_adapter_down_multihead=True, encoder_adapter_multihead_num_head=nh, adapter_down_dim=r,
                          use_encoder_adapter_gating_large_x_lowrank=True, adapter_gating_down_dim=r,
                          use_encoder_adapter_gating_add=add, use_encoder_gating_scaling=gs != 1.0,
                          encoder_gating_scaling_factor=gs)
    m = nn.Module(); EP.build_pet(m, cfg, d, ("ff",))
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.04)
    q = lambda t: t.to(dtype).float()
    x1, x2, dy = q(torch.randn(B, S, d)), q(torch.randn(B, S, d)), q(torch.randn(B, S, d))
    P = {n: p.detach().clone().requires_grad_(True) for n, p in m.named_parameters()}
    x1r, x2r = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    gate = dict(down_w=P["encoder_ff_adapter_gating_large_x_down.weight"], down_b=P["encoder_ff_adapter_gating_large_x_down.bias"],
                up_w=P["encoder_ff_adapter_gating_large_x_up.weight"], up_b=P["encoder_ff_adapter_gating_large_x_up.bias"])
    ref = O.encoder_adapter_gate(x1r, x2r, [P[f"ff_adapter_multihead_down.{i}.weight"] for i in range(nh)],
                                 [P[f"ff_adapter_multihead_down.{i}.bias"] for i in range(nh)],
                                 P["ff_adapter_multihead_up.weight"], P["ff_adapter_multihead_up.bias"], gate, O.GATE_LARGE,
                                 add, 1.0, 1.0, gs)
    ref.backward(dy)
    m.cuda()
    tol = 1e-3 if dtype == torch.float32 else 1e-2
    rel = lambda a, b: float((a.detach().float().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-6))
    for split in (True, False):
        EP.SPLIT_WIDE_BOTTLENECK = split
        try:
            for p in m.parameters():
                p.grad = None
            X1, X2 = x1.cuda().to(dtype).requires_grad_(True), x2.cuda().to(dtype).requires_grad_(True)
            y = EP.apply_pet(m, "ff", X1, X2, cfg)
            y.backward(dy.cuda().to(dtype))
            assert rel(y, ref.detach()) <= tol, ("y", split)
            assert rel(X1.grad, x1r.grad) <= tol and rel(X2.grad, x2r.grad) <= tol, ("dx", split)
            for n, p in m.named_parameters():
                assert rel(p.grad, P[n].grad) <= tol, (n, split, rel(p.grad, P[n].grad))
        finally:
            EP.SPLIT_WIDE_BOTTLENECK = False      # the default: fused 6-tile kernels


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_residual_gradient_handover_between_tail_and_k1(dtype):
    """Encoder layer with the K1 -> K5 residual link on (the tail's dx1 is summed inside K1's backward kernel, no autograd
    add) and off (plain autograd): same output, same gradients.  Rows = 5000: the chain-split row kernel's in-epilogue add."""
    import copy
    import vlpet_amd.host.bart as HB
    torch.manual_seed(4)
    cfg = HB.vlpet_config(encoder_layers=1, decoder_layers=1, vocab_size=300, dropout=0.0, attention_dropout=0.0,
                          activation_dropout=0.0)
    layer = HB.BartEncoderLayer(cfg)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(torch.randn_like(p) * 0.03)
    layer.cuda().train()
    x = torch.randn(50, 100, 768, device="cuda").to(dtype)
    dy = torch.randn(50, 100, 768, device="cuda").to(dtype)
    import vlpet_amd.train as TR
    for n, p in layer.named_parameters():          # the reference's trainable set for this layer; frozen weights in the IO dtype
        p.requires_grad = ("adapter" in n) or ("gating" in n) or ("layer_norm" in n)
    TR.cast_frozen(layer, dtype)
    outs = []
    for fuse in (True, False):
        HB.FUSE_RESIDUAL_GRAD = fuse
        try:
            for p in layer.parameters():
                p.grad = None
            xi = x.clone().requires_grad_(True)
            y = layer(xi)
            y.backward(dy)
            outs.append((y.detach().float(), xi.grad.float(), {n: p.grad.float().clone() for n, p in layer.named_parameters() if p.grad is not None}))
        finally:
            HB.FUSE_RESIDUAL_GRAD = True
    assert torch.equal(outs[0][0], outs[1][0])
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert (outs[0][1] - outs[1][1]).abs().max().item() <= tol * outs[1][1].abs().max().item()
    for n in outs[1][2]:
        a, b = outs[0][2][n], outs[1][2][n]
        assert (a - b).abs().max().item() <= tol * max(b.abs().max().item(), 1e-6), n


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("which", ["encoder", "decoder"])
def test_gemm_gradient_handover(which, p_drop):
    """The dgrad GEMM of a sublayer's first frozen projection accumulates onto the gradient K1 / K5 / K2 parked
    (functional.linear_acc, host/bart.py _first_linear) -- against plain autograd sums (FUSE_GEMM_GRAD off): same output,
    same gradients for the layer input, the encoder output and every trainable parameter.  p_drop = 0 is the case where the
    tail's dy and dx1 are one tensor (the consumer must not accumulate in place)."""
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    torch.manual_seed(11)
    dtype = torch.bfloat16
    cfg = HB.vlpet_config(encoder_layers=1, decoder_layers=1, vocab_size=300, dropout=p_drop, attention_dropout=p_drop,
                          activation_dropout=p_drop)
    layer = HB.BartEncoderLayer(cfg) if which == "encoder" else HB.BartDecoderLayer(cfg)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(torch.randn_like(p) * 0.03)
    layer.cuda().train()
    for n, p in layer.named_parameters():
        p.requires_grad = ("adapter" in n) or ("gating" in n) or ("layer_norm" in n)
    TR.cast_frozen(layer, dtype)
    B, L, S = 40, 20, 56
    x = torch.randn(B, S if which == "encoder" else L, 768, device="cuda").to(dtype)
    enc = torch.randn(B, S, 768, device="cuda").to(dtype)
    dy = torch.randn_like(x)
    outs = []
    for fuse in (True, False):
        HB.FUSE_GEMM_GRAD = fuse
        try:
            for p in layer.parameters():
                p.grad = None
            torch.manual_seed(5)                    # the dropout seeds of the kernels are drawn from torch's generator
            xi, ei = x.clone().requires_grad_(True), enc.clone().requires_grad_(True)
            y = layer(xi) if which == "encoder" else layer(xi, ei, task="vqa")
            y.backward(dy)
            g = {n: p.grad.float().clone() for n, p in layer.named_parameters() if p.grad is not None}
            g["<input>"] = xi.grad.float()
            if which == "decoder":
                g["<encoder output>"] = ei.grad.float()
            outs.append((y.detach().float(), g))
        finally:
            HB.FUSE_GEMM_GRAD = True
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1].keys() == outs[1][1].keys() and len(outs[0][1]) > 2
    for n in outs[1][1]:
        a, b = outs[0][1][n], outs[1][1][n]
        assert (a - b).abs().max().item() <= 2e-2 * max(b.abs().max().item(), 1e-6), n


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("which", ["encoder", "decoder"])
def test_t5_norm_gradient_handover(which, p_drop):
    """T5 block with the residual-stream hand-over into the RMS norm's backward kernel (host/t5.py FUSE_NORM_GRAD) against
    plain autograd sums: same output, same gradients for the block input, the encoder output and every trainable parameter."""
    import vlpet_amd.host.t5 as HT
    import vlpet_amd.train as TR
    torch.manual_seed(13)
    dtype = torch.bfloat16
    cfg = HT.vlt5_config(num_layers=1, num_decoder_layers=1, vocab_size=300, dropout_rate=p_drop)
    blk = HT.T5Block(cfg, which == "decoder", True)
    with torch.no_grad():
        for p in blk.parameters():
            p.copy_(torch.randn_like(p) * 0.03)
    blk.cuda().train()
    for n, p in blk.named_parameters():
        p.requires_grad = ("adapter" in n) or ("gating" in n) or ("layer_norm" in n and which == "encoder")
    TR.cast_frozen(blk, dtype)
    B, L, S = 24, 20, 56
    x = torch.randn(B, S if which == "encoder" else L, 768, device="cuda").to(dtype)
    enc = torch.randn(B, S, 768, device="cuda").to(dtype)
    dy = torch.randn_like(x)
    outs = []
    for fuse in (True, False):
        HT.FUSE_NORM_GRAD = fuse
        try:
            for p in blk.parameters():
                p.grad = None
            torch.manual_seed(5)
            xi, ei = x.clone().requires_grad_(True), enc.clone().requires_grad_(True)
            y = blk(xi, None) if which == "encoder" else blk(xi, None, enc=ei, cross_bias=None, task="vqa")
            y.backward(dy)
            g = {n: p.grad.float().clone() for n, p in blk.named_parameters() if p.grad is not None}
            g["<input>"] = xi.grad.float()
            if which == "decoder":
                g["<encoder output>"] = ei.grad.float()
            outs.append((y.detach().float(), g))
        finally:
            HT.FUSE_NORM_GRAD = True
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1].keys() == outs[1][1].keys() and len(outs[0][1]) > 2
    for n in outs[1][1]:
        a, b = outs[0][1][n], outs[1][1][n]
        assert (a - b).abs().max().item() <= 2e-2 * max(b.abs().max().item(), 1e-6), n


@pytest.mark.parametrize("which", ["encoder", "decoder"])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_t5_tail_fused_with_the_next_norm(which, p_drop):
    """A stack of two T5 blocks + the final norm with every sublayer tail applying the NEXT sublayer's RMS norm in the same pass
    (host/t5.py FUSE_TAIL_NORM, tail.sublayer_tail_rms: my_transformers/modeling_t5.py:408 + :366 of the next sublayer as one launch each
    way) against the two-launch form: same output (bit for bit: the statistic is taken on the rounded sum either way, the masks come
    from the same seeds), same gradients for the input, the encoder output and every trainable parameter incl. the norms' weights."""
    import vlpet_amd.host.t5 as HT
    import vlpet_amd.train as TR
    from vlpet_amd.visual import T5LayerNorm
    torch.manual_seed(17)
    dtype = torch.bfloat16
    cfg = HT.vlt5_config(num_layers=2, num_decoder_layers=2, vocab_size=300, dropout_rate=p_drop)
    blocks = torch.nn.ModuleList([HT.T5Block(cfg, which == "decoder", i == 0) for i in range(2)])
    final = T5LayerNorm(cfg.d_model, eps=cfg.layer_norm_epsilon)
    HT._wire_next_norms(blocks, final)
    mods = torch.nn.ModuleList([blocks, final])
    with torch.no_grad():
        for n, p in mods.named_parameters():
            p.copy_(torch.randn_like(p) * 0.03 + (1.0 if "layer_norm" in n or n.startswith("1.") else 0.0))
    mods.cuda().train()
    for n, p in mods.named_parameters():
        p.requires_grad = ("adapter" in n) or ("gating" in n) or ("layer_norm" in n) or n.startswith("1.")
    TR.cast_frozen(mods, dtype)
    B, L, S = 16, 20, 56
    x = torch.randn(B, S if which == "encoder" else L, 768, device="cuda").to(dtype)
    enc = torch.randn(B, S, 768, device="cuda").to(dtype)
    outs = []
    for fuse in (True, False):
        HT.FUSE_TAIL_NORM = fuse
        try:
            for p in mods.parameters():
                p.grad = None
            torch.manual_seed(5)
            xi, ei = x.clone().requires_grad_(True), enc.clone().requires_grad_(True)
            h = xi
            for blk in blocks:
                h = blk(h, None) if which == "encoder" else blk(h, None, enc=ei, cross_bias=None, task="vqa")
            assert (getattr(h, "_vlpet_norm", None) is not None) == fuse
            y = final(h)
            dy = torch.randn(y.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)).to(dtype)
            y.backward(dy)
            g = {n: p.grad.float().clone() for n, p in mods.named_parameters() if p.grad is not None}
            g["<input>"] = xi.grad.float()
            if which == "decoder":
                g["<encoder output>"] = ei.grad.float()
            outs.append((y.detach().float(), g))
        finally:
            HT.FUSE_TAIL_NORM = True
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1].keys() == outs[1][1].keys() and len(outs[0][1]) > 6
    for n in outs[1][1]:
        a, b = outs[0][1][n], outs[1][1][n]
        assert (a - b).abs().max().item() <= 2e-2 * max(b.abs().max().item(), 1e-6), n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("training", [False, True])
def test_lora_base_dgrad_takes_over_the_delta_gradient(dtype, training):
    """LoRA projection with a frozen weight and a trainable bias (the LoRA runs' parameter rule): K3's d/dx handed to the base
    projection's dgrad GEMM (lora/controller.py LINK_DELTA_GRAD) against autograd's add -- output, d/dx, dA, dB, dbias."""
    import vlpet_amd.lora.controller as LC
    from vlpet_amd.lora import LoraConfig, LoRALinearController
    torch.manual_seed(2)
    d, r = 768, 64
    lin = LoRALinearController(d, d, config=LoraConfig(lora_dim=r, lora_alpha=32, tasks=["vqa"]), bias=True).cuda()
    with torch.no_grad():
        lin.weight.copy_(torch.randn(d, d) * 0.03); lin.bias.copy_(torch.randn(d) * 0.1)
        lin.lora_As["vqa"].copy_(torch.randn(r, d) * 0.05); lin.lora_Bs["vqa"].copy_(torch.randn(d, r) * 0.05)
    lin.weight.requires_grad = False
    lin.weight.data = lin.weight.data.to(dtype)
    lin.train(training)
    x0 = torch.randn(700, d, device="cuda").to(dtype)
    dy = torch.randn(700, d, device="cuda").to(dtype)
    outs = []
    for on in (True, False):
        LC.LINK_DELTA_GRAD = on
        try:
            for p in lin.parameters():
                p.grad = None
            torch.manual_seed(7)
            x = x0.clone().requires_grad_(True)
            out = lin(x, "vqa")
            out.backward(dy)
            outs.append((out.detach().float(), x.grad.float(), lin.lora_As["vqa"].grad.float().clone(),
                         lin.lora_Bs["vqa"].grad.float().clone(), lin.bias.grad.float().clone()))
        finally:
            LC.LINK_DELTA_GRAD = True
    assert torch.equal(outs[0][0], outs[1][0])
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert (a - b).abs().max().item() <= tol * max(b.abs().max().item(), 1e-6)
