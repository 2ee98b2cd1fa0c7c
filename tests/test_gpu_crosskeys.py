"""GPU: the decoder layers' cross-attention key projections of the encoder output as ONE GEMM each way (functional.cross_key_blocks,
attention.KeyGradSlot, vlpet_attn_{fwd,bwd}_kv) -- my_transformers/modeling_bart.py:2300-2330 hands every decoder layer the same
encoder_hidden_states and each layer's encoder_attn.k_proj projects it (:425)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
H, E = 2, 128


def _abi_attn(q, k, v, do, ld_k, p=0.1):
    """forward + backward through vlpet_attn_{fwd,bwd}_kv; k is read with row stride ld_k, dk written with the same"""
    from vlpet_amd import _lib
    lib = _lib.load()
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    st = torch.cuda.current_stream().cuda_stream
    o = torch.empty_like(q)
    lse = torch.empty(B, H, Lq, dtype=torch.float32, device="cuda")
    rc = lib.vlpet_attn_fwd_kv(q.data_ptr(), k.data_ptr(), v.data_ptr(), None, None, o.data_ptr(), lse.data_ptr(), None, B, H, Lq, Lk,
                               E, ld_k, E, 0, 0.125, p, 7, st)
    assert rc == 0
    dq, dv = torch.empty_like(q), torch.empty_like(v)
    dk = torch.zeros(k.shape, dtype=k.dtype, device="cuda") if ld_k == E else None
    return o, lse, dq, dk, dv, st, lib


@pytest.mark.parametrize("Lq,Lk", [(5, 56), (20, 92), (33, 76)])
def test_attention_reads_its_keys_as_a_column_block_and_writes_dk_into_one(Lq, Lk):
    """k as block 1 of a [B, Lk, 3 E] buffer (ld_k = 3 E), v contiguous: the same bits as the contiguous call, forward and backward; the
    neighbouring blocks of the gradient buffer are left alone."""
    torch.manual_seed(Lq)
    B, n = 7, 3
    mk = lambda L: (torch.randn(B, L, E) * 0.5).cuda().bfloat16()
    q, v, do = mk(Lq), mk(Lk), mk(Lq)
    kall = (torch.randn(B, Lk, n * E) * 0.5).cuda().bfloat16()
    kblk = kall[..., E:2 * E]
    kc = kblk.contiguous()
    outs = []
    for k, ld in ((kc, E), (kblk, n * E)):
        o, lse, dq, _, dv, st, lib = _abi_attn(q, k, v, do, ld)
        if ld == E:
            dk = torch.empty_like(kc)
            dkp = dk
        else:
            dkall = torch.full((B, Lk, n * E), 3.0, device="cuda").bfloat16()
            dkp = dkall[..., E:2 * E]
        rc = lib.vlpet_attn_bwd_kv(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), None, None, None,
                                   dq.data_ptr(), dkp.data_ptr(), dv.data_ptr(), B, H, Lq, Lk, E, ld, E, 0, 0.125, 0.1, 7, st)
        assert rc == 0
        torch.cuda.synchronize()
        outs.append((o.clone(), dq.clone(), dkp.clone(), dv.clone()))
        if ld != E:
            assert float((dkall[..., :E].float() - 3.0).abs().max()) == 0.0 and float((dkall[..., 2 * E:].float() - 3.0).abs().max()) == 0.0
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_fused_key_projection_with_the_attention_kernels_takes_one_gemm_each_way(monkeypatch):
    """n attention calls on the column blocks of one key projection == n calls on separately projected keys: outputs, d/dx of the projected
    tensor, and the backward must have consumed the shared gradient buffer whole (one GEMM with K = n E)."""
    import vlpet_amd.functional as VF
    from vlpet_amd.attention import short_attention
    torch.manual_seed(1)
    B, Lq, Lk, n = 6, 9, 56, 4
    x = (torch.randn(B, Lk, E) * 0.5).cuda().bfloat16().requires_grad_(True)
    lins = [torch.nn.Linear(E, E).cuda().bfloat16() for _ in range(n)]
    for l in lins:
        l.requires_grad_(False)
    qs = [(torch.randn(B, Lq, E) * 0.5).cuda().bfloat16() for _ in range(n)]
    vs = [(torch.randn(B, Lk, E) * 0.5).cuda().bfloat16() for _ in range(n)]
    dos = [(torch.randn(B, Lq, E) * 0.5).cuda().bfloat16() for _ in range(n)]
    w = torch.cat([l.weight for l in lins], 0).contiguous()
    b = torch.cat([l.bias for l in lins], 0).contiguous()
    seen = []
    orig = VF._CrossKeyProjFn.backward
    monkeypatch.setattr(VF._CrossKeyProjFn, "backward",
                        staticmethod(lambda ctx, *g: (seen.append(ctx.slot is not None and ctx.slot.buf is not None
                                                                  and all(t.data_ptr() == ctx.slot.buf.data_ptr() + i * E * 2 for i, t in enumerate(g))),
                                                      orig(ctx, *g))[1]))
    ks, slot = VF.cross_key_blocks(x, w, b, n)
    outs = [short_attention(q, k, v, H, p=0.0, training=False, k_slot=(slot, i)) for i, (q, k, v) in enumerate(zip(qs, ks, vs))]
    sum((o.float() * d.float()).sum() for o, d in zip(outs, dos)).backward()
    g_fused, x.grad = x.grad, None
    assert seen == [True]
    refs = [short_attention(q, l(x), v, H, p=0.0, training=False) for q, l, v in zip(qs, lins, vs)]
    sum((o.float() * d.float()).sum() for o, d in zip(refs, dos)).backward()
    for o, r in zip(outs, refs):
        assert float((o.float() - r.float()).abs().max()) <= 2.0 ** -7 * float(r.float().abs().max())      # (one GEMM vs n: a bf16 ulp of the keys)
    err = float((g_fused.float() - x.grad.float()).abs().max()) / float(x.grad.float().abs().max())
    assert err <= 2e-2, err         # n bf16-rounded partial sums (autograd) vs one fp32-accumulated GEMM


def _small_bart():
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    cfg = HB.vlpet_config(d_model=128, encoder_layers=2, decoder_layers=3, encoder_attention_heads=2, decoder_attention_heads=2,
                          encoder_ffn_dim=256, decoder_ffn_dim=256, vocab_size=500, max_position_embeddings=64, feat_dim=128,
                          adapter_down_dim=8, adapter_gating_down_dim=16, decoder_enc_attn_value_parallel_adapter_down_dim=8,
                          dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    torch.manual_seed(0)
    model = HB.VLBart(cfg)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    TR.trainable_names(model, cfg)
    model.cuda()
    TR.cast_frozen(model, torch.bfloat16)
    model.train()
    return model, cfg


@pytest.mark.parametrize("graph", [False, True])
def test_trainer_with_fused_cross_keys_equals_per_layer_projections(graph, monkeypatch):
    """A head-dim-64 BART (the tiny fixtures' head dim is 16: they never reach the short-sequence attention kernels) trained for a few steps
    with the decoder's fused key projection and with per-layer projections: the same losses and parameters up to bf16 rounding; the fused
    path must actually have run (KeyGradSlot buffers consumed)."""
    import vlpet_amd.functional as VF
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    from vlpet_amd import _lib
    model, cfg = _small_bart()
    gen = torch.Generator().manual_seed(3)
    tasks = ("vqa", "nlvr", "vqa", "nlvr", "vqa", "nlvr")
    data = {}
    for t in ("vqa", "nlvr"):
        b = TR.synthetic_batch(t, 6, cfg, "cpu", gen)
        bb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
        bb["vis_inputs"] = tuple(x.cuda() for x in b["vis_inputs"])
        data[t] = bb
    used = []
    orig = VF.cross_key_blocks
    monkeypatch.setattr(VF, "cross_key_blocks", lambda *a: (used.append(1), orig(*a))[1])
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(HB, "FUSE_CROSS_KEYS", fused)
        m = copy.deepcopy(model)
        tr = TR.Trainer(m, cfg, lr=1e-2, total_steps=20, warmup_ratio=0.1, graph=graph)
        losses = [float(tr.step(data[t])) for t in tasks]
        res[fused] = (losses, {n: p.detach().float().clone() for n, p in m.named_parameters() if p.requires_grad})
        _lib.load().vlpet_set_seed_counter(None)
    assert used, "the fused key projection did not run"
    for a, b in zip(res[True][0], res[False][0]):
        assert abs(a - b) <= 2e-2 * abs(b), (res[True][0], res[False][0])
    worst = max(float((res[True][1][n] - res[False][1][n]).abs().max() / (res[False][1][n].abs().max() + 1e-12)) for n in res[True][1])
    assert worst <= 5e-2, worst


def test_t5_trainer_with_fused_cross_keys_equals_per_block_projections(monkeypatch):
    """The same for the T5 host (bias-free projections, relative-position bias in the attention kernels, RMS norms)."""
    import vlpet_amd.functional as VF
    import vlpet_amd.host.t5 as HT
    import vlpet_amd.train as TR
    from vlpet_amd import _lib
    cfg = HT.vlt5_config(d_model=128, d_kv=64, d_ff=256, num_layers=2, num_decoder_layers=3, num_heads=2, vocab_size=500, feat_dim=128,
                         adapter_down_dim=16, adapter_gating_down_dim=16, decoder_enc_attn_value_parallel_adapter_down_dim=8, dropout_rate=0.0)
    torch.manual_seed(0)
    model = HT.VLT5(cfg)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    TR.trainable_names(model, cfg)
    model.cuda()
    TR.cast_frozen(model, torch.bfloat16)
    model.train()
    gen = torch.Generator().manual_seed(4)
    b = TR.synthetic_batch("vqa", 6, cfg, "cpu", gen)
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    batch["vis_inputs"] = tuple(x.cuda() for x in b["vis_inputs"])
    used = []
    orig = VF.cross_key_blocks
    monkeypatch.setattr(VF, "cross_key_blocks", lambda *a: (used.append(1), orig(*a))[1])
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(HT, "FUSE_CROSS_KEYS", fused)
        m = copy.deepcopy(model)
        tr = TR.Trainer(m, cfg, lr=1e-2, total_steps=20, warmup_ratio=0.1)
        res[fused] = [float(tr.step(batch)) for _ in range(4)]
        _lib.load().vlpet_set_seed_counter(None)
    assert used, "the fused key projection did not run"
    for a, c in zip(res[True], res[False]):
        assert abs(a - c) <= 2e-2 * abs(c), res
