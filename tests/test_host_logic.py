"""Host-side logic on CPU: parameter naming / freeze rules against the reference-derived fixtures,
state-dict keys of the drop-in modules, task order, loss reduction, flat-gradient bookkeeping."""
import os

import numpy as np
import pytest
import torch

import vlpet_amd.host.bart as HB
import vlpet_amd.train as TR
from vlpet_amd.adapters import AdapterConfig, AdapterController
from vlpet_amd.lora import LoraConfig, LoRALinearController
from vlpet_amd.visual import VisualEmbedding

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


def test_layer_parameter_names_match_reference():
    g = load("names_bart_vlpet_large")
    cfg = HB.vlpet_config()
    enc = HB.BartEncoderLayer(cfg)
    dec = HB.BartDecoderLayer(cfg)
    mine = {n: p.numel() for n, p in enc.named_parameters()}
    ref = dict(zip([str(s) for s in g["enc_names"]], [int(v) for v in g["enc_numel"]]))
    assert mine == ref
    mine = {n: p.numel() for n, p in dec.named_parameters()}
    ref = dict(zip([str(s) for s in g["dec_names"]], [int(v) for v in g["dec_numel"]]))
    assert mine == ref


@pytest.mark.timeout(600)
def test_trainable_set_bart_vlpet_large():
    cfg = HB.vlpet_config()
    model = HB.VLBart(cfg)
    names = TR.trainable_names(model, cfg)
    n = sum(p.numel() for p in model.parameters() if p.requires_grad)
    total = sum(p.numel() for p in model.parameters())
    assert (n, total) == (6052416, 145606464)          # README.md:360 -> 4.16 %
    assert all(("adapter" in s or "gating" in s or "visual_embedding" in s or "layer_norm" in s or "layernorm" in s)
               for s in names)
    assert not any(s.startswith("model.decoder.") and "adapter" not in s for s in names)
    assert "model.shared.weight" not in names


def test_state_dict_keys_of_drop_in_modules():
    keys = [str(s) for s in load("k2_d768_r96")["state_keys"]]
    ctl = AdapterController(AdapterConfig(tasks=["vqa", "gqa", "nlvr", "caption"], d_model=768, input_dim=768,
                                          use_single_adapter=True, use_adapter_down_dim=True, adapter_down_dim=96,
                                          use_parallel_adapter=True))
    assert sorted(ctl.state_dict().keys()) == keys
    keys = [str(s) for s in load("k3_d64_r4")["state_keys"]]
    lin = LoRALinearController(64, 64, config=LoraConfig(lora_dim=4, tasks=["vqa", "gqa", "nlvr", "caption"]), bias=True)
    assert sorted(lin.state_dict().keys()) == keys
    assert lin.scaling == 32 / 4 and not lin.weight.requires_grad
    assert all(float(lin.lora_Bs[t].abs().sum()) == 0 for t in lin.tasks)    # B zero-init: step-0 output == frozen
    keys = [str(s) for s in load("k4_bart_d64_f128")["state_keys"]]
    cfg = HB.vlpet_config(d_model=64, feat_dim=128)
    ve = VisualEmbedding(cfg, torch.nn.Embedding(200, 64))
    assert sorted(ve.state_dict().keys()) == keys


def test_task_order_and_batches():
    steps = {"vqa": 3, "gqa": 2, "nlvr": 1, "caption": 2}
    a = TR.epoch_task_order(["vqa", "gqa", "nlvr", "caption"], steps, 5)
    b = TR.epoch_task_order(["vqa", "gqa", "nlvr", "caption"], steps, 5)
    assert a == b and sorted(a) == sorted(sum(([t] * k for t, k in steps.items()), []))
    assert a != TR.epoch_task_order(["vqa", "gqa", "nlvr", "caption"], steps, 6)
    assert [TR.TASK_BATCH[t](500) for t in ("vqa", "gqa", "nlvr", "caption")] == [500, 833, 166, 416]


def test_task_loss_reduction():
    per = torch.tensor([[1.0, 3.0, 5.0], [2.0, 2.0, 2.0]])
    labels = torch.tensor([[4, 5, -100], [7, -100, -100]])
    scores = torch.tensor([1.0, 0.5])
    assert torch.isclose(TR.task_loss(per, labels, scores, "vqa"), torch.tensor((2.0 * 1.0 + 2.0 * 0.5) / 2))
    assert torch.isclose(TR.task_loss(per, labels, None, "caption"), torch.tensor((1 + 3 + 2) / 3.0))


def test_flat_grads_views_and_clip():
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 2))
    fg = TR.FlatGrads(m, world_size=1, n_buckets=2)
    fg.zero()
    x = torch.randn(4, 8)
    m(x).sum().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in fg.params])
    assert torch.equal(ref, fg.flat[:fg.total]) and float(fg.flat.abs().sum()) > 0      # flat is padded to 16 bytes
    norm = fg.clip_(0.1)
    assert torch.isclose(torch.linalg.vector_norm(fg.flat), torch.tensor(0.1), atol=1e-5) and norm > 0.1
    assert len(fg.buckets) == 2 and fg.buckets[0][0] == 0 and fg.buckets[-1][1] == fg.total


def test_cpu_reference_ops_run_the_host_model_on_cpu():
    """The checker path used by bench.py's cpu_baseline and the GPU model-parity test: host model on CPU with
    every HIP-backed op routed to the oracle (one tiny train step; the product ops themselves refuse CPU tensors)."""
    import pytest
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    from oracle.host_patch import cpu_reference_ops
    cfg = HB.vlpet_config(d_model=64, encoder_layers=1, decoder_layers=1, encoder_attention_heads=4,
                          decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, vocab_size=500,
                          max_position_embeddings=64, feat_dim=128, adapter_down_dim=8, adapter_gating_down_dim=16,
                          decoder_enc_attn_value_parallel_adapter_down_dim=8)
    torch.manual_seed(0)
    model = HB.VLBart(cfg)
    TR.trainable_names(model, cfg)
    model.train()
    gen = torch.Generator().manual_seed(5)
    batch = TR.synthetic_batch("nlvr", 3, cfg, "cpu", gen)
    with cpu_reference_ops():
        tr = TR.Trainer(model, cfg, total_steps=10)
        l0 = tr.step(batch)
        l1 = tr.step(batch)
    assert l0 == l0 and l1 == l1          # finite
    with pytest.raises(RuntimeError):      # outside the patch the product path refuses CPU tensors
        model(batch["input_ids"], batch["vis_inputs"], batch["labels"], "nlvr")


def test_per_task_parameter_detection():
    """FlatGrads.per_task drives the optimizer's per-parameter step counts (transformers.AdamW skips grad-None parameters)."""
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    small = dict(d_model=64, encoder_layers=1, decoder_layers=1, encoder_attention_heads=4, decoder_attention_heads=4,
                 encoder_ffn_dim=128, decoder_ffn_dim=128, vocab_size=300, max_position_embeddings=64, feat_dim=128,
                 adapter_down_dim=8, adapter_gating_down_dim=8, decoder_enc_attn_value_parallel_adapter_down_dim=8)
    for over, expect in ((dict(), False), (dict(use_single_adapter=False), True)):
        cfg = HB.vlpet_config(**small, **over)
        m = HB.VLBart(cfg)
        names = TR.trainable_names(m, cfg)
        ps = dict(m.named_parameters())
        assert TR._has_per_task_params(names, [ps[n] for n in names]) is expect
    lora = dict(use_adapter=False, use_encoder_adapter_down_multihead=False, use_encoder_adapter_gating_large_x_lowrank=False,
                use_decoder_enc_attn_value_parallel_adapter_down_dim=False, use_lora=True, lora_dim=4)
    for single, expect in ((True, False), (False, True)):
        cfg = HB.vlpet_config(**small, **lora, use_single_lora=single)
        m = HB.VLBart(cfg)
        names = TR.trainable_names(m, cfg)
        ps = dict(m.named_parameters())
        assert TR._has_per_task_params(names, [ps[n] for n in names]) is expect


# ---------------------------------------------------------------------------------------------------------------------------
# functional.linear_acc: the dgrad GEMM of a frozen projection accumulates onto a parked gradient (pure torch: runs on CPU)

def _park_op():
    """A stand-in for K1 / K5 / K2: reads x and y = proj(x), parks its d/dx in the link when that is armed."""
    import torch

    class Park(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, y, link, shared):
            ctx.link = link if (link is not None and link.armed) else None
            ctx.shared = shared
            ctx.save_for_backward(x, y)
            return y * x.sum(-1, keepdim=True) + 3.0 * x[..., : y.shape[-1]]

        @staticmethod
        def backward(ctx, g):
            x, y = ctx.saved_tensors
            dy = g * x.sum(-1, keepdim=True)
            dx = (g * y).sum(-1, keepdim=True).expand_as(x).clone()
            dx[..., : y.shape[-1]] += 3.0 * g
            if ctx.link is not None:
                ctx.link.dx1, ctx.link.shared = dx, ctx.shared
                return None, dy, None, None
            return dx, dy, None, None
    return Park


@pytest.mark.parametrize("n_proj", [1, 2])
@pytest.mark.parametrize("shared", [False, True])
def test_linear_acc_accumulates_onto_parked_gradient(n_proj, shared):
    import torch
    import vlpet_amd.functional as VF
    torch.manual_seed(0)
    Park = _park_op()
    lins = [torch.nn.Linear(12, 8).requires_grad_(False) for _ in range(n_proj)]
    x0 = torch.randn(3, 5, 12)
    g = torch.randn(3, 5, 8)

    def run(linked):
        x = x0.clone().requires_grad_(True)
        link = VF.ResidualLink() if linked else None
        ys = VF.linear_acc(x, link, *lins) if linked else tuple(l(x) for l in lins)
        ys = ys if isinstance(ys, tuple) else (ys,)
        if linked:
            assert link.armed
        out = Park.apply(x, ys[0], link, shared)
        for y in ys[1:]:
            out = out + y.tanh()
        out.backward(g)
        if linked:
            assert link.dx1 is None         # taken
        return out.detach(), x.grad

    (o1, g1), (o0, g0) = run(True), run(False)
    assert torch.equal(o1, o0)
    assert torch.allclose(g1, g0, rtol=1e-5, atol=1e-6)


def test_linear_acc_without_a_park_and_without_grad():
    import torch
    import vlpet_amd.functional as VF
    lin = torch.nn.Linear(6, 4).requires_grad_(False)
    x = torch.randn(7, 6, requires_grad=True)
    link = VF.ResidualLink()
    y = VF.linear_acc(x, link, lin)         # armed, nobody parks: the plain dgrad
    y.sum().backward()
    assert torch.allclose(x.grad, lin.weight.sum(0).expand(7, 6))
    link = VF.ResidualLink()
    VF.linear_acc(x.detach(), link, lin)
    assert not link.armed                   # nothing to hand over when the input needs no gradient
    with pytest.raises(RuntimeError):
        VF.linear_acc(x, None, torch.nn.Linear(6, 4))      # trainable projections are refused


def test_t5_attn_spec_dense_forms_are_the_reference_masks():
    """host/t5.AttnSpec keeps T5's relative bias / key padding / causality apart for the on-chip attention kernels; its dense() is
    what the reference adds to the scores (my_transformers/modeling_t5.py:640-660, src/modeling_t5.py:311-327): bias + (1 - mask) *
    -10000 in the joint encoder, bias + causal * -10000 in the decoder, (1 - mask) * -1e9 for the cross attention."""
    import vlpet_amd.host.t5 as HT
    g = torch.Generator().manual_seed(0)
    H, L = 3, 7
    rel = torch.randn(1, H, L, L, generator=g)
    keep = torch.tensor([[1, 1, 1, 1, 1, 0, 0], [1, 1, 1, 1, 1, 1, 1]], dtype=torch.float32)
    enc = HT.AttnSpec(rel, keep, causal=False).dense(torch.float32)
    assert torch.equal(enc, rel + (1.0 - keep[:, None, None, :]) * -10000.0)
    dec = HT.AttnSpec(rel, None, causal=True).dense(torch.float32)
    tri = torch.tril(torch.ones(L, L))
    assert torch.equal(dec, rel + (1.0 - tri)[None, None] * -10000.0)
    cross = HT.AttnSpec(None, keep, causal=False).dense(torch.float32)
    assert torch.equal(cross, (1.0 - keep[:, None, None, :]) * -1e9)
    assert HT.AttnSpec(None, None, causal=False).dense(torch.float32) is None
    assert HT.AttnSpec(rel, keep, causal=False, rel_trainable=True).fast() is None      # a trainable bias needs its gradient: dense path


def test_trainer_batch_signature_separates_tasks_and_shapes():
    """train.Trainer captures one graph per batch signature: same task + same shapes -> same key, anything else a new one."""
    import vlpet_amd.train as TR
    b1 = dict(task="vqa", input_ids=torch.zeros(4, 20, dtype=torch.long), labels=torch.zeros(4, 5, dtype=torch.long),
              vis_inputs=(torch.zeros(4, 49, 8), torch.zeros(4, 49, 4)), scores=torch.ones(4), no_padding=False)
    b2 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in b1.items()}
    b3 = dict(b1, task="gqa")
    b4 = dict(b1, input_ids=torch.zeros(5, 20, dtype=torch.long))
    b5 = dict(b1, no_padding=True)
    sig = TR.Trainer._signature
    assert sig(b1) == sig(b2)
    assert len({sig(b) for b in (b1, b3, b4, b5)}) == 4


def test_key_mask_byte_conversion_is_cached_per_source_and_version():
    """attention._key_mask_u8: every layer passes a fresh view of the same boolean mask; the byte form is converted once and
    reconverted after an in-place write to the mask (the version counter is part of the key)."""
    import vlpet_amd.attention as A
    A._KM_CACHE.clear()
    m = torch.rand(4, 1, 1, 7) > 0.3
    a = A._key_mask_u8(m[:, 0, 0, :], 4, 7)
    b = A._key_mask_u8(m[:, 0, 0, :], 4, 7)
    assert a is b and a.dtype == torch.uint8 and a.shape == (4, 7) and a.is_contiguous()
    assert bool((a == m[:, 0, 0, :].to(torch.uint8)).all())
    m[0, 0, 0, 0] = ~m[0, 0, 0, 0]                                # in place: same storage, new version
    c = A._key_mask_u8(m[:, 0, 0, :], 4, 7)
    assert c is not a and bool((c == m[:, 0, 0, :].to(torch.uint8)).all())
    other = torch.ones(4, 7, dtype=torch.bool)
    d = A._key_mask_u8(other, 4, 7)
    assert bool((d == 1).all()) and A._key_mask_u8(m[:, 0, 0, :], 4, 7) is c     # two sources are kept
    u8 = torch.ones(4, 7, dtype=torch.uint8)
    assert A._key_mask_u8(u8, 4, 7) is u8                         # already bytes: passed through
    assert len(A._KM_CACHE) <= 2


def test_trainable_layernorm_form_recheck_is_one_batched_read_and_reports_flips():
    """tail.recheck_trainable_norms: the K5 backward form of every trainable LayerNorm with a cached decision is re-derived in
    one pass; a decision that flips is reported (a graph-replaying trainer drops its captures), frozen ones and untouched ones are left."""
    import torch.nn as nn
    from vlpet_amd import tail
    import vlpet_amd.functional as VF
    m = nn.Sequential(nn.LayerNorm(16), nn.LayerNorm(16), nn.LayerNorm(16))
    m[2].weight.requires_grad_(False); m[2].bias.requires_grad_(False)
    for ln in m:
        ln.weight._vlpet_prenorm = (("k",), VF.WEIGHTS_EPOCH, False)
    assert tail.recheck_trainable_norms(m) is False                      # gamma = 1, beta = 0: ratio 0 everywhere
    with torch.no_grad():
        m[1].bias.fill_(3.0); m[1].weight.fill_(0.1)                     # ratio 30 > PRENORM_RATIO: the exact form
        m[2].bias.fill_(100.0)                                           # frozen: not looked at here
    assert tail.recheck_trainable_norms(m) is True
    assert m[0].weight._vlpet_prenorm[2] is False and m[1].weight._vlpet_prenorm[2] is True and m[2].weight._vlpet_prenorm[2] is False
    assert tail.recheck_trainable_norms(m) is False                      # nothing flips the second time


def test_fused_key_projection_blocks_equal_separate_projections():
    """functional.cross_key_blocks on CPU tensors (no attention kernel, so no shared gradient buffer: the per-block fallback of its backward):
    n projections of one input as one GEMM whose column blocks are handed out == n separate nn.Linear calls, forward and d/dx."""
    import torch
    import vlpet_amd.functional as VF
    torch.manual_seed(0)
    n, E = 3, 16
    x = torch.randn(2, 5, E, requires_grad=True)
    lins = [torch.nn.Linear(E, E) for _ in range(n)]
    w = torch.cat([l.weight for l in lins], 0).detach()
    b = torch.cat([l.bias for l in lins], 0).detach()
    ks, slot = VF.cross_key_blocks(x, w, b, n)
    assert slot is not None and len(ks) == n and all(k.shape == (2, 5, E) for k in ks)
    cs = [torch.randn(2, 5, E) for _ in range(n)]
    sum((k * c).sum() for k, c in zip(ks, cs)).backward()
    g, x.grad = x.grad, None
    sum((l(x) * c).sum() for l, c in zip(lins, cs)).backward()
    for k, l in zip(ks, lins):
        assert torch.allclose(k, l(x), atol=1e-6)
    assert torch.allclose(g, x.grad, atol=1e-5)
    with torch.no_grad():
        assert VF.cross_key_blocks(x, w, b, n)[1] is None
