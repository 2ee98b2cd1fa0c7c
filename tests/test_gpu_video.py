"""GPU parity at the shapes of BASELINE configs[4] (BART-base + VL-PET video-text multitask) and at the full sizes of
configs[1] that the first round left untested.

Video shapes (reference): batch 50 for every task (scripts/video-text/VL-PET-large.sh:17,48), 64 frame features of
feat_dim 512 (multitask_video.py:738; video/how2qa_data.py:34-44,163), `--n_boxes 64 --downsample`
(VL-PET-large.sh:52) = Downsample((8, 8)) on an 8 x 8 grid, text truncated at 600 tokens
(video/how2qa_data.py:203, tvqa_data.py:211)  =>  S_enc = 664, M = 50 * 664 = 33,200 rows for K1 / K2 / K5 and
M_v = 50 * 64 = 3,200 rows for K4."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
import gpu_cases as C  # noqa: E402
from gpu_cases import rel_err  # noqa: E402
from oracle import vlpet_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2}
M_VIDEO = 50 * (600 + 64)


def check(errs, dtype):
    bad = {k: v for k, v in errs.items() if not v <= TOL[dtype]}
    assert not bad, f"over tolerance {TOL[dtype]}: {bad} (all: {errs})"


def test_k1_video_rows_bf16():
    check(C.run_k1(torch.bfloat16, M=M_VIDEO), torch.bfloat16)


def test_k2_video_rows_bf16():
    check(C.run_k2(torch.bfloat16, M=M_VIDEO), torch.bfloat16)


def test_k5_video_rows_bf16():
    from test_gpu_tail import _run
    mask = _run(M_VIDEO, 768, torch.bfloat16, 0.1)
    assert abs(float(mask.mean()) - 0.9) < 0.002


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_k4_video_feat512(dtype):
    """VisualEmbedding with feat_dim 512, 64 frames, batch 50 against the oracle: output + every parameter gradient."""
    from test_gpu_k4 import build
    torch.manual_seed(5)
    B, N, F, d = 50, 64, 512, 768
    ve, table = build(d, F, False, vocab=300)
    with torch.no_grad():
        for p in ve.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    feats = torch.randn(B, N, F).to(dtype)
    pos = torch.zeros(B, N, 4)                       # the video loaders pass zero boxes (video/how2qa_data.py:170)
    dy = torch.randn(B, N, d).to(dtype)
    fe, pe = ve.feat_embedding, ve.absolute_vis_pos_embedding
    names = [fe[0].weight, fe[0].bias, fe[1].weight, fe[1].bias, pe[0].weight, pe[0].bias, pe[1].weight, pe[1].bias,
             ve.img_order_embedding.weight, table.weight]
    ref = [t.detach().clone().requires_grad_(True) for t in names]
    out_ref = O.visual_embedding(feats.float(), pos, *ref[:8], ref[8], ref[9])
    out_ref.backward(dy.float())
    ve = ve.cuda()
    out = ve(feats.cuda(), pos.cuda())
    tol = TOL[dtype]
    assert rel_err(out, out_ref) <= tol
    out.backward(dy.cuda())
    got = [fe[0].weight, fe[0].bias, fe[1].weight, fe[1].bias, pe[0].weight, pe[0].bias, pe[1].weight, pe[1].bias,
           ve.img_order_embedding.weight]
    for a, b in zip(got, ref[:9]):
        if b.grad is not None and float(b.grad.abs().max()) > 0:
            assert rel_err(a.grad, b.grad) <= tol


def test_downsample_video_grid_is_identity_pool():
    """Downsample((8, 8)) on the 8 x 8 grid of 64 frame features: every output window is one input cell."""
    from vlpet_amd.visual import Downsample
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(50, 64, 512, generator=gen)
    boxes = torch.zeros(50, 64, 4)
    ds = Downsample((8, 8))
    ref = O.downsample(x, (8, 8))
    assert torch.equal(ref, x)
    y, b = ds((x.cuda(), boxes.cuda()))
    assert torch.equal(y.cpu(), ref) and b.shape == (50, 64, 4)
    yb, _ = ds((x.cuda(), boxes.cuda()), out_dtype=torch.bfloat16)
    assert torch.equal(yb.cpu(), ref.to(torch.bfloat16))


def test_video_step_tiny_host_runs_and_matches_cpu_checker():
    """One train step of a 2+2-layer video-config host (feat_dim 512, n_boxes 64, text 600) on the product path against the
    same host with the HIP ops swapped for the oracle (fp32, dropout off)."""
    import copy
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    from oracle.host_patch import cpu_reference_ops
    torch.manual_seed(21)
    cfg = HB.vlpet_config(encoder_layers=2, decoder_layers=2, vocab_size=1000, feat_dim=512, n_boxes=64,
                          tasks="tvqa,how2qa,tvc,yc2c", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    model = HB.VLBart(cfg)
    TR.trainable_names(model, cfg)
    ref_model = copy.deepcopy(model)
    gen = torch.Generator().manual_seed(5)
    b = TR.synthetic_batch("how2qa", 3, cfg, "cpu", gen)
    b["input_ids"][1, 400:] = cfg.pad_token_id          # ragged text: the default mask path
    b["no_padding"] = False
    with cpu_reference_ops():
        ref_model.train()
        per, _ = ref_model(b["input_ids"], b["vis_inputs"], b["labels"], b["task"])
        loss_ref = TR.task_loss(per, b["labels"], b["scores"], b["task"])
        loss_ref.backward()
    model.cuda().train()
    bg = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    bg["vis_inputs"] = tuple(t.cuda() for t in b["vis_inputs"])
    per, _ = model(bg["input_ids"], bg["vis_inputs"], bg["labels"], bg["task"])
    loss = TR.task_loss(per, bg["labels"], bg["scores"], bg["task"])
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) <= 1e-3 * max(1.0, abs(float(loss_ref)))
    ref_grads = dict(ref_model.named_parameters())
    worst = 0.0
    for n, p in model.named_parameters():
        if p.requires_grad and ref_grads[n].grad is not None and float(ref_grads[n].grad.abs().max()) > 1e-8:
            worst = max(worst, rel_err(p.grad, ref_grads[n].grad))
    assert worst <= 2e-3, worst


# ---- full sizes of BASELINE configs[1] not covered before
def test_k1_full_size_fp32():
    check(C.run_k1(torch.float32, M=28000), torch.float32)              # VQA step: 500 * 56 rows, fp32 IO


def test_k1_gqa_step_rows_bf16():
    check(C.run_k1(torch.bfloat16, M=46648), torch.bfloat16)            # GQA step: 833 * 56 rows, the largest launch of the bench
