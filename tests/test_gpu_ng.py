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
