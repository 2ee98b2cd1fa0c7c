"""GPU parity tests: the HIP path (through the C ABI via vlpet_amd.functional) against the CPU oracle
on the same seeded inputs.  Tolerances are BASELINE.json's: 1e-3 (fp32 IO) / 1e-2 (bf16 IO), measured
as max-abs error over max-abs reference per tensor, forward AND backward."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
import gpu_cases as C  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2}


def check(errs, dtype):
    bad = {k: v for k, v in errs.items() if not v <= TOL[dtype]}
    assert not bad, f"over tolerance {TOL[dtype]}: {bad} (all: {errs})"


@pytest.mark.parametrize("r,d,nh,fp32", [(96, 768, 4, False), (96, 768, 4, True), (8, 64, 4, False), (192, 768, 4, False)])
def test_pack_matches_spec(r, d, nh, fp32):
    assert C.run_pack_check(r, d, nh, fp32) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,d,r,rg,nh", [(224, 768, 96, 96, 4), (40, 64, 8, 8, 4), (1000, 768, 96, 48, 4), (129, 128, 16, 40, 2)])
def test_k1_large_gate(dtype, M, d, r, rg, nh):
    check(C.run_k1(dtype, M=M, d=d, r=r, rg=rg, nh=nh), dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_k1_variants(dtype):
    check(C.run_k1(dtype, M=130, gate_mode=0), dtype)                       # adapter only
    check(C.run_k1(dtype, M=130, gate_mode=2, gate_scale=0.3), dtype)       # additive gate + gate scale
    check(C.run_k1(dtype, M=100, r=192, rg=192, delta_scale=4.0, x2_scale=0.5, gate_scale=0.3), dtype)  # T5 script


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rg", [2, 3, 4])
def test_k1_every_workgroup_size(dtype, rg, monkeypatch):
    # rows per workgroup are picked per launch (csrc/kernels.h pick_row_groups): 64 / 96 / 128-row workgroups of the
    # forward (with loader waves) and the backward rows kernel, forced one by one; M leaves a ragged last workgroup
    monkeypatch.setenv("VLPET_RG", str(rg))
    check(C.run_k1(dtype, M=1000, d=768, r=96, rg=96, nh=4), dtype)
    check(C.run_k1(dtype, M=333, d=128, r=16, rg=40, nh=2, gate_mode=2, gate_scale=0.3), dtype)
    check(C.run_k2(dtype, M=777), dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_k1_backward_recompute_path(dtype, monkeypatch):
    # default: the gated forward saves z / gelu' for the backward; this is the other path (backward recomputes from x1, x2)
    import vlpet_amd.functional as F
    monkeypatch.setattr(F, "SAVE_ACTIVATIONS", False)
    check(C.run_k1(dtype, M=1000, d=768, r=96, rg=48, nh=4), dtype)
    check(C.run_k1(dtype, M=130, gate_mode=2, gate_scale=0.3), dtype)
    check(C.run_k2(dtype, M=333, r=8, d=64, scale=4.0), dtype)
    check(C.run_k2(dtype), dtype)


def test_k1_full_size_bf16():
    # config 2 (VQA step): M = 500 * 56 rows
    check(C.run_k1(torch.bfloat16, M=28000), torch.bfloat16)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_k2(dtype):
    check(C.run_k2(dtype), dtype)
    check(C.run_k2(dtype, M=333, r=8, d=64, scale=4.0), dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("r,p", [(4, 0.0), (8, 0.0), (64, 0.0), (128, 0.1), (8, 0.1)])
def test_k3(dtype, r, p):
    check(C.run_k3(dtype, r=r, p=p), dtype)


def test_fails_loudly_on_cpu_tensor():
    import vlpet_amd.functional as F
    with pytest.raises(RuntimeError):
        F.pack_pair([torch.zeros(8, 64)], [torch.zeros(8)], torch.zeros(64, 8), torch.zeros(64), 1)


def test_k1_backward_transpose_read_wgrad_variant(monkeypatch):
    """The opt-in weight-gradient kernel built on ds_read_b64_tr_b16 (VLPET_WGRAD_TR=1) gives the same gradients."""
    monkeypatch.setenv("VLPET_WGRAD_TR", "1")
    C.run_k1(torch.bfloat16, M=1000, d=768, r=96, rg=96, nh=4)
    C.run_k1(torch.bfloat16, M=333, d=256, r=8, rg=16, nh=4)
