"""Backward WITHOUT a gate in its round-3 two-pass form (csrc/pet_cols_ng.hip: bf16, r <= 96, the forward's saved activations) --
K2 (adapters/adapter_modeling.py:55-61), the adapter-only K1 of the small / middle gate scripts, K3 without dropout -- against
the CPU oracle through the product path: one row, the 32-row step and 128-row workgroup boundaries, other widths and ranks,
a scale, the identity activation; full size."""
import pytest
import torch

import gpu_cases as C

pytestmark = pytest.mark.gpu
TOL = 1e-2          # BASELINE.json: 1e-2 for bf16 IO


def _check(errs):
    bad = {k: v for k, v in errs.items() if k != "keep_frac" and not (v <= TOL)}
    assert not bad, errs


@pytest.mark.parametrize("kw", [dict(M=1), dict(M=31), dict(M=33), dict(M=127), dict(M=129), dict(M=777, r=8), dict(M=1000, r=32),
                                dict(M=640, d=256, r=16), dict(M=3000, d=1024), dict(M=5000, scale=0.25), dict(M=28000), dict(M=46648),
                                dict(M=100, r=192), dict(M=2100, r=192), dict(M=16800, r=192), dict(M=1000, r=128),    # six tiles
                                dict(M=3528), dict(M=15272), dict(M=17000), dict(M=33200), dict(M=41999, r=32)],       # round 4: two-pass forward sizes
                         ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_k2_two_pass_vs_oracle(kw):
    _check(C.run_k2(torch.bfloat16, **kw))


@pytest.mark.parametrize("kw", [dict(M=130), dict(M=1000, delta_scale=0.5, x2_scale=0.7), dict(M=777, r=8, rg=8, nh=4), dict(M=15272)],
                         ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_k1_adapter_only_two_pass_vs_oracle(kw):
    _check(C.run_k1(torch.bfloat16, gate_mode=0, **kw))


@pytest.mark.parametrize("M,r,p", [(200, 8, 0.0), (2500, 64, 0.0), (28000, 64, 0.0), (333, 8, 0.1), (2500, 64, 0.1), (28000, 64, 0.1), (2500, 128, 0.1),
                                   (28000, 128, 0.1)])
def test_k3_two_pass_vs_oracle(M, r, p):
    """p > 0: the training form with the forward's packed mask (the oracle gets the exported mask); r = 128: six tiles."""
    _check(C.run_k3(torch.bfloat16, M=M, r=r, p=p))


@pytest.mark.parametrize("M,d,r,p,explicit", [(1, 768, 8, 0.0, False), (63, 768, 8, 0.1, False), (257, 768, 4, 0.1, False), (1000, 768, 1, 0.0, False),
                                                (777, 512, 8, 0.1, False), (640, 1024, 6, 0.1, True), (3500, 768, 8, 0.1, True),
                                                (5000, 256, 8, 0.5, False), (28000, 768, 8, 0.1, False)])
def test_k3_rank8_streaming_form_vs_oracle(M, d, r, p, explicit):
    """K3 at lora_dim <= 8 runs as a streaming row kernel (csrc/lora8.hip; lora/controller.py:56-70): forward (+ the saved block the
    unchanged two-pass backward reads) against the oracle -- one row, ragged sizes, ranks below 8, one / two pieces per lane, both
    mask sources."""
    import vlpet_amd.functional as F
    from vlpet_amd import _lib
    assert F.LORA_R8_STREAMING and _lib.load().vlpet_lora_r8_applies(M, d, r, F._io_dtype(torch.empty(1, dtype=torch.bfloat16))) == 1
    _check(C.run_k3(torch.bfloat16, M=M, d=d, r=r, p=p, explicit_mask=explicit))



def test_k3_rank8_streaming_form_equals_the_mfma_form():
    """Same pack, same generator: the two forward forms give the same mask bit for bit, the same saved block, and outputs / gradients
    that differ by summation order only."""
    import vlpet_amd.functional as F
    g = torch.Generator().manual_seed(11)
    M, d, r, p = 4100, 768, 8, 0.1
    x = (torch.randn(M, d, generator=g)).cuda().bfloat16()
    base = (torch.randn(M, d, generator=g)).cuda().bfloat16()
    dy = (torch.randn(M, d, generator=g)).cuda().bfloat16()
    A0, B0 = (torch.randn(r, d, generator=g) / d ** 0.5).cuda(), (torch.randn(d, r, generator=g) * 0.3).cuda()
    res = []
    try:
        for streaming in (True, False):
            F.LORA_R8_STREAMING = streaming
            xg = x.clone().requires_grad_(True)
            Ag, Bg = A0.clone().requires_grad_(True), B0.clone().requires_grad_(True)
            pk = F.pack_pair([Ag], None, Bg, None, F._io_dtype(xg))
            out, mask = F.lora_delta(xg, base, Ag, Bg, pk, 4.0, None, p, 1234, return_mask=True)
            out.backward(dy)
            res.append((out.detach().float(), mask.clone(), xg.grad.float(), Ag.grad.clone(), Bg.grad.clone()))
    finally:
        F.LORA_R8_STREAMING = True
    (o1, m1, dx1, da1, db1), (o2, m2, dx2, da2, db2) = res
    assert torch.equal(m1, m2)
    for a, b in ((o1, o2), (dx1, dx2), (da1, da2), (db1, db2)):
        assert float((a - b).abs().max()) <= 1e-2 * float(b.abs().max())
