"""Fused clip + AdamW over the flat trainable buffer (csrc/optim.hip) against the oracle's restatement of
clip_grad_norm_ + transformers.optimization.AdamW, and the whole Trainer step (direct-write gradient sinks,
fused optimizer) against the same model stepped on the CPU through the oracle."""
import copy

import pytest
import torch

from oracle import vlpet_oracle as O

pytestmark = pytest.mark.gpu


def _run_kernel(p, g_steps, mask, max_norm, grad_scale, lr, wd, variant):
    from vlpet_amd import _lib
    lib = _lib.load()
    n = p.numel()
    P, M, V = p.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    mk = mask.cuda()
    nb = lib.vlpet_optim_blocks(n)
    part = torch.empty(nb, device="cuda")
    norm = torch.zeros((), device="cuda")
    norms = []
    st = torch.cuda.current_stream().cuda_stream
    for t, g in enumerate(g_steps, 1):
        G = g.clone().cuda()
        assert lib.vlpet_grad_sumsq(G.data_ptr(), n, part.data_ptr(), st) == 0
        assert lib.vlpet_adamw_step(P.data_ptr(), G.data_ptr(), M.data_ptr(), V.data_ptr(), mk.data_ptr(), n, part.data_ptr(), nb,
                                    max_norm, grad_scale, lr, 0.9, 0.999, 1e-6, wd, t, variant, 1, norm.data_ptr(), st) == 0
        assert float(G.abs().sum()) == 0.0          # zero_grad
        norms.append(float(norm))
    return P.cpu(), M.cpu(), V.cpu(), norms


@pytest.mark.parametrize("n", [10007, 4096, 6052416])
@pytest.mark.parametrize("max_norm,grad_scale", [(5.0, 1.0), (0.5, 0.5)])
def test_adamw_matches_hf_restatement(n, max_norm, grad_scale):
    gen = torch.Generator().manual_seed(n)
    p = torch.randn(n, generator=gen)
    gs = [torch.randn(n, generator=gen) * (0.01 if i else 0.1) for i in range(3)]
    mask = (torch.rand(n, generator=gen) < 0.7).to(torch.uint8)
    lr, wd = 1e-3, 0.01
    P, M, V, norms = _run_kernel(p, gs, mask, max_norm, grad_scale, lr, wd, 0)
    pr, m, v = p.clone(), torch.zeros(n), torch.zeros(n)
    for t, g in enumerate(gs, 1):
        g = g.clone() * grad_scale
        nr = O.clip_grad_norm([g], max_norm)
        assert abs(norms[t - 1] - float(nr)) <= 1e-4 * float(nr)
        dec, nod = mask.bool(), ~mask.bool()
        # per element decay: run the restatement on the two groups
        for sel, w in ((dec, wd), (nod, 0.0)):
            ps, ms, vs = pr[sel].clone(), m[sel].clone(), v[sel].clone()
            O.hf_adamw_step(ps, g[sel], ms, vs, t, lr, eps=1e-6, weight_decay=w)
            pr[sel], m[sel], v[sel] = ps, ms, vs
    torch.testing.assert_close(P, pr, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(M, m, rtol=2e-5, atol=1e-7)
    torch.testing.assert_close(V, v, rtol=2e-5, atol=1e-9)


def test_adamw_variant1_matches_torch_adamw():
    n = 5000
    gen = torch.Generator().manual_seed(1)
    p = torch.randn(n, generator=gen)
    gs = [torch.randn(n, generator=gen) * 0.05 for _ in range(4)]
    mask = torch.ones(n, dtype=torch.uint8)
    P, _, _, _ = _run_kernel(p, gs, mask, 0.0, 1.0, 2e-3, 0.01, 1)
    q = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([q], lr=2e-3, eps=1e-6, weight_decay=0.01)
    for g in gs:
        q.grad = g.clone()
        opt.step()
    torch.testing.assert_close(P, q.detach(), rtol=2e-5, atol=2e-6)


def test_trainer_step_gpu_equals_cpu_reference():
    """Three full train steps (tiny BART host, dropout off): the HIP path with gradients written straight into
    the flat buffer + the fused optimizer lands on the same parameters as the CPU path through the oracle."""
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    from oracle.host_patch import cpu_reference_ops
    cfg = HB.vlpet_config(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                          decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, vocab_size=500,
                          max_position_embeddings=64, feat_dim=128, adapter_down_dim=8, adapter_gating_down_dim=16,
                          decoder_enc_attn_value_parallel_adapter_down_dim=8, dropout=0.0, attention_dropout=0.0,
                          activation_dropout=0.0)
    torch.manual_seed(0)
    model = HB.VLBart(cfg)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    TR.trainable_names(model, cfg)
    model.train()
    gpu_model = copy.deepcopy(model).cuda()
    gen = torch.Generator().manual_seed(5)
    batches = [TR.synthetic_batch(t, 5, cfg, "cpu", gen) for t in ("vqa", "nlvr", "caption")]
    with cpu_reference_ops():
        tr = TR.Trainer(model, cfg, lr=1e-2, total_steps=10, warmup_ratio=0.1)
        ref_losses = [float(tr.step(b)) for b in batches]
    trg = TR.Trainer(gpu_model, cfg, lr=1e-2, total_steps=10, warmup_ratio=0.1)
    assert trg.flat.sinks_enabled
    losses = []
    for b in batches:
        bb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
        bb["vis_inputs"] = tuple(t.cuda() for t in b["vis_inputs"])
        losses.append(float(trg.step(bb)))
    # the direct-write path really ran: every K1/K2/K4 sink was taken in the last step
    taken = [s for p in trg.flat.params for s in (getattr(p, "_vlpet_sink", None), getattr(p, "_vlpet_block_sink", None))
             if s is not None and s.epoch == trg.flat.epoch - 1]
    assert len(taken) >= 2 * 2 * 6 + 2 * 4          # 2 enc layers x 2 sublayers x 6 blocks + 2 dec layers x 4
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 2e-3 * abs(b)
    ref = dict(model.named_parameters())
    worst = 0.0
    for n, p in gpu_model.named_parameters():
        if p.requires_grad:
            err = float((p.detach().cpu() - ref[n].detach()).abs().max())
            worst = max(worst, err / max(1e-3, float(ref[n].detach().abs().max())))
    assert worst <= 2e-2, worst        # Adam's m/sqrt(v) amplifies tiny gradient differences in the first steps


def test_training_with_dropout_reduces_the_loss():
    """Sanity of the whole stochastic path (in-kernel dropout of the tails regenerated in the backward, attention /
    activation dropout, fused optimizer): 40 steps on one fixed batch must overfit it."""
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    cfg = HB.vlpet_config(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                          decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, vocab_size=500,
                          max_position_embeddings=64, feat_dim=128, adapter_down_dim=8, adapter_gating_down_dim=16,
                          decoder_enc_attn_value_parallel_adapter_down_dim=8, dropout=0.1, attention_dropout=0.1,
                          activation_dropout=0.1)
    torch.manual_seed(1)
    model = HB.VLBart(cfg)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    TR.trainable_names(model, cfg)
    model.train().cuda()
    tr = TR.Trainer(model, cfg, lr=5e-3, total_steps=80, warmup_ratio=0.05)
    gen = torch.Generator().manual_seed(2)
    b = TR.synthetic_batch("caption", 8, cfg, "cpu", gen)
    bb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    bb["vis_inputs"] = tuple(t.cuda() for t in b["vis_inputs"])
    losses = [float(tr.step(bb)) for _ in range(40)]
    assert all(l == l for l in losses)                       # finite
    # only the PET parameters, the encoder LayerNorms and the visual embedding train (4 % of a random frozen backbone): the
    # loss falls slowly but steadily
    assert sum(losses[-5:]) / 5 < sum(losses[:3]) / 3 - 0.1, (losses[:3], losses[-5:])


def test_adamw_sliced_per_parameter_steps_and_skips():
    """vlpet_adamw_step_sliced: parameters that received no gradient are left untouched (no decay, no moment update) and
    every parameter is bias-corrected with its OWN update count -- transformers.AdamW with grads set to None."""
    import math
    from vlpet_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(9)
    sizes = [1000, 37, 4096, 3, 768, 250]
    n = sum(sizes)
    n_pad = (n + 3) // 4 * 4
    p = torch.randn(n_pad, generator=gen)
    mask = (torch.rand(n_pad, generator=gen) < 0.7).to(torch.uint8)
    slice_of = torch.zeros(n_pad, dtype=torch.int32)
    offs, o = [], 0
    for k, sz in enumerate(sizes):
        slice_of[o:o + sz] = k
        offs.append((o, o + sz)); o += sz
    slice_of[n:] = len(sizes) - 1
    schedule = [[1, 1, 1, 1, 1, 1], [1, 0, 1, 0, 0, 1], [0, 1, 1, 0, 1, 1], [1, 1, 0, 0, 1, 1]]
    lr, wd, b1, b2 = 1e-2, 0.01, 0.9, 0.999
    P, M, V = p.clone().cuda(), torch.zeros(n_pad, device="cuda"), torch.zeros(n_pad, device="cuda")
    pr, m, v = p.clone(), torch.zeros(n_pad), torch.zeros(n_pad)
    steps = [0] * len(sizes)
    nb = lib.vlpet_optim_blocks(n_pad)
    part = torch.empty(nb, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    so = slice_of.cuda()
    for act in schedule:
        g = torch.randn(n_pad, generator=gen) * 0.05
        bc = torch.zeros(len(sizes), 2)
        for k, on in enumerate(act):
            a, b = offs[k]
            if on:
                steps[k] += 1
                bc[k, 0] = 1 - b1 ** steps[k]; bc[k, 1] = math.sqrt(1 - b2 ** steps[k])
            else:
                g[a:b] = 0.0            # grad None: contributes nothing to the norm
                bc[k, 0] = -1.0
        g[n:] = 0.0
        G = g.clone().cuda()
        assert lib.vlpet_grad_sumsq(G.data_ptr(), n_pad, part.data_ptr(), st) == 0
        assert lib.vlpet_adamw_step_sliced(P.data_ptr(), G.data_ptr(), M.data_ptr(), V.data_ptr(), mask.cuda().data_ptr(), n_pad,
                                           part.data_ptr(), nb, 5.0, 1.0, lr, b1, b2, 1e-6, wd, so.data_ptr(),
                                           bc.cuda().data_ptr(), 0, 1, None, st) == 0
        torch.cuda.synchronize()
        assert float(G.abs().sum()) == 0.0
        gc = g.clone()
        O.clip_grad_norm([gc], 5.0)
        for k, on in enumerate(act):
            if not on:
                continue
            a, b = offs[k]
            for sel_dec, w in ((True, wd), (False, 0.0)):
                sel = torch.zeros(n_pad, dtype=torch.bool); sel[a:b] = (mask[a:b].bool() == sel_dec)
                ps, ms, vs = pr[sel].clone(), m[sel].clone(), v[sel].clone()
                O.hf_adamw_step(ps, gc[sel], ms, vs, steps[k], lr, eps=1e-6, weight_decay=w)
                pr[sel], m[sel], v[sel] = ps, ms, vs
    torch.testing.assert_close(P.cpu()[:n], pr[:n], rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(M.cpu()[:n], m[:n], rtol=2e-5, atol=1e-7)
    torch.testing.assert_close(V.cpu()[:n], v[:n], rtol=2e-5, atol=1e-9)


def test_trainer_per_task_adapters_gpu_equals_cpu_reference():
    """use_single_adapter off (one decoder value-parallel adapter per task): the other tasks' adapters get no gradient in a
    step and must not be decayed / moved on stale momentum / have their step count advanced (train.py:384)."""
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    from oracle.host_patch import cpu_reference_ops
    cfg = HB.vlpet_config(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                          decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, vocab_size=500,
                          max_position_embeddings=64, feat_dim=128, adapter_down_dim=8, adapter_gating_down_dim=16,
                          decoder_enc_attn_value_parallel_adapter_down_dim=8, dropout=0.0, attention_dropout=0.0,
                          activation_dropout=0.0, use_single_adapter=False)
    torch.manual_seed(0)
    model = HB.VLBart(cfg)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    TR.trainable_names(model, cfg)
    model.train()
    gpu_model = copy.deepcopy(model).cuda()
    gen = torch.Generator().manual_seed(5)
    order = ("vqa", "nlvr", "vqa", "caption", "nlvr")
    batches = [TR.synthetic_batch(t, 4, cfg, "cpu", gen) for t in order]
    with cpu_reference_ops():
        tr = TR.Trainer(model, cfg, lr=1e-2, total_steps=10, warmup_ratio=0.1)
        assert tr.flat.per_task
        ref_losses = [float(tr.step(b)) for b in batches]
    trg = TR.Trainer(gpu_model, cfg, lr=1e-2, total_steps=10, warmup_ratio=0.1)
    assert trg.optim.sliced
    losses = []
    for b in batches:
        bb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
        bb["vis_inputs"] = tuple(t.cuda() for t in b["vis_inputs"])
        losses.append(float(trg.step(bb)))
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 2e-3 * abs(b)
    ref = dict(model.named_parameters())
    init = dict(copy.deepcopy(model).named_parameters())
    worst = 0.0
    for n, p in gpu_model.named_parameters():
        if p.requires_grad:
            err = float((p.detach().cpu() - ref[n].detach()).abs().max())
            worst = max(worst, err / max(1e-3, float(ref[n].detach().abs().max())))
    assert worst <= 2e-2, worst
    # the gqa adapters never saw a gradient: bit-identical to their initial values on both paths
    sd0 = {n: p for n, p in copy.deepcopy(gpu_model).named_parameters()}
    never = [n for n in trg.flat.names if ".adapters.gqa." in n]
    assert never
    k = {n: i for i, n in enumerate(trg.flat.names)}
    assert all(trg.optim.steps[k[n]] == 0 for n in never)
    assert all(trg.optim.steps[k[n]] == 2 for n in trg.flat.names if ".adapters.vqa." in n)
