"""Host model + trainer against a tiny VLBart captured from the reference's own classes
(tests/golden/vlbart_tiny_d64.npz, make_goldens.golden_vlbart_tiny): state-dict keys, [text ; visual] encoder
order, hook placement, Downsample, label shifting, per-task loss reduction, clip + AdamW + warm-up schedule,
trainable set.  CPU leg: host model with the HIP-backed ops swapped for the oracle; GPU leg: the product path."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


FIXTURES = {"vlpet_large": ("vlbart_tiny_d64", {}),
            # scripts/image-text/single_lora.sh: LoRA r=8 on q_proj / v_proj of every attention, one LoRA for all tasks
            "lora": ("vlbart_tiny_lora_d64", dict(use_adapter=False, use_encoder_adapter_down_multihead=False,
                                                  use_encoder_adapter_gating_large_x_lowrank=False,
                                                  use_decoder_enc_attn_value_parallel_adapter_down_dim=False,
                                                  unfreeze_encoder_layer_norms=False, use_lora=True, lora_dim=8, lora_dropout=0.0,
                                                  use_single_lora=True)),
            # scripts/image-text/T5-VL-PET-large.sh on the reference's VLT5 (gate scale 0.3, RMS norms, pre-LN tails)
            "t5": ("vlt5_tiny_d64", None),
            # scripts/video-text/VL-PET-large.sh (BASELINE configs[4]): 64 frame features through Downsample((8, 8)), four
            # video tasks, ragged text padded with the pad id -> pins the default mask input_ids.ne(pad) in encoder
            # self-attention and decoder cross-attention (src/modeling_bart.py:817-818, 995-996) and the video loss
            "video": ("vlbart_tiny_video_d64", dict(tasks="tvqa,how2qa,tvc,yc2c", n_boxes=64))}


def _load(name):
    z = np.load(os.path.join(G, name + ".npz"), allow_pickle=False)
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    batches = []
    for i in range(3):
        vis = tuple(torch.from_numpy(z[f"b{i}::vis{j}"]) for j in range(4) if f"b{i}::vis{j}" in z.files)
        batches.append(dict(task=str(z[f"b{i}::task"]), input_ids=torch.from_numpy(z[f"b{i}::ids"]),
                            labels=torch.from_numpy(z[f"b{i}::labels"]), scores=torch.from_numpy(z[f"b{i}::scores"]),
                            vis_inputs=vis, per_token=torch.from_numpy(z[f"b{i}::per_token"]),
                            logits=torch.from_numpy(z[f"b{i}::logits"]), loss=float(z[f"b{i}::loss"])))
    final = {k[7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("final::")}
    return sd, batches, z["train::losses"], z["train::hparams"], final


def _build(sd, over):
    import vlpet_amd.train as TR
    if over is None:
        import vlpet_amd.host.t5 as HT
        cfg = HT.vlt5_config(d_model=64, d_kv=16, d_ff=128, num_layers=2, num_decoder_layers=2, num_heads=4, vocab_size=500,
                             feat_dim=128, adapter_down_dim=8, adapter_gating_down_dim=16,
                             decoder_enc_attn_value_parallel_adapter_down_dim=8, dropout_rate=0.0)
        model = HT.VLT5(cfg)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not missing, missing
        # aliases of the shared embedding in the reference's state dict
        assert set(unexpected) <= {"lm_head.weight", "encoder.embed_tokens.weight", "decoder.embed_tokens.weight"}, unexpected
        return model, cfg, TR.trainable_names(model, cfg)
    import vlpet_amd.host.bart as HB
    cfg = HB.vlpet_config(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                          decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, vocab_size=500,
                          max_position_embeddings=64, feat_dim=128, adapter_down_dim=8, adapter_gating_down_dim=16,
                          decoder_enc_attn_value_parallel_adapter_down_dim=8, dropout=0.0, attention_dropout=0.0,
                          activation_dropout=0.0, **over)
    model = HB.VLBart(cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing, missing                                       # every host parameter exists in the reference
    assert set(unexpected) <= {"lm_head.weight"}, unexpected          # tied head: the host reuses model.shared.weight
    names = TR.trainable_names(model, cfg)
    return model, cfg, names


def _to(b, dev):
    out = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    out["vis_inputs"] = tuple(t.to(dev) for t in b["vis_inputs"])
    return out


def _check(model, cfg, names, batches, ref_losses, hp, final, dev, tol):
    import vlpet_amd.train as TR
    assert sorted(names) == sorted(final.keys())                      # the reference's trainable set, name for name
    model.eval()
    with torch.no_grad():
        for b in batches:
            bb = _to(b, dev)
            per, logits = model(bb["input_ids"], bb["vis_inputs"], bb["labels"], b["task"])
            torch.testing.assert_close(logits.float().cpu(), b["logits"], rtol=tol, atol=tol)
            torch.testing.assert_close(per.float().cpu(), b["per_token"], rtol=tol, atol=tol)
            loss = TR.task_loss(per.float().cpu(), b["labels"], b["scores"], b["task"])
            assert abs(float(loss) - b["loss"]) <= tol * max(1.0, abs(b["loss"]))
    model.train()
    base_lr, total, warm, clip = [float(x) for x in hp]
    tr = TR.Trainer(model, cfg, lr=base_lr, clip=clip, total_steps=int(total), warmup_ratio=warm / total)
    losses = [float(tr.step(_to(batches[i % 3], dev))) for i in range(5)]
    for a, r in zip(losses, ref_losses):
        assert abs(a - float(r)) <= 5 * tol * max(1.0, abs(float(r))), (losses, list(ref_losses))
    cur = dict(model.named_parameters())
    # Adam normalises every gradient component to O(1): a component whose true gradient is ~0 (e.g. k_proj.bias,
    # which softmax is invariant to) moves by +-lr per step on rounding noise alone.  Judge the final parameters at
    # the scale of the distance travelled: 10 % of lr * steps, plus the usual relative term (the tight pins are the
    # logits, the per-token losses and the loss trajectory above).
    travelled = base_lr * 5
    for n, ref in final.items():
        err = (cur[n].detach().float().cpu() - ref).abs().reshape(-1)
        lim = 20 * tol * max(1e-2, float(ref.abs().max()))
        # typical element: tight;  noise-driven elements (|g| ~ eps = 1e-6, sign decided by rounding): bounded by the
        # distance travelled and rare
        assert float(err.median()) <= lim + 0.01 * travelled, (n, float(err.median()))
        assert float((err > lim + 0.1 * travelled).float().mean()) <= 0.05, (n, float(err.max()))
        assert float(err.max()) <= lim + 2.0 * travelled, (n, float(err.max()))


@pytest.mark.parametrize("which", list(FIXTURES))
def test_host_and_trainer_match_reference_vlbart_cpu(which):
    from oracle.host_patch import cpu_reference_ops
    name, over = FIXTURES[which]
    sd, batches, ref_losses, hp, final = _load(name)
    model, cfg, names = _build(sd, over)
    with cpu_reference_ops():
        _check(model, cfg, names, batches, ref_losses, hp, final, "cpu", 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("which", list(FIXTURES))
def test_host_and_trainer_match_reference_vlbart_gpu(which):
    name, over = FIXTURES[which]
    sd, batches, ref_losses, hp, final = _load(name)
    model, cfg, names = _build(sd, over)
    model.cuda()
    _check(model, cfg, names, batches, ref_losses, hp, final, "cuda", 1e-3)       # fp32 IO tolerance of north_star
