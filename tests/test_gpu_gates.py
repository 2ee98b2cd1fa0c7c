"""small / middleX / middleY granularity gates (csrc/rowgate.hip + the adapter-only K1 kernel) against the
reference-generated fixtures and the oracle."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import vlpet_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
FLAG = {O.GATE_SMALL: "use_encoder_adapter_gating_small_xy_cat", O.GATE_MIDDLE_X: "use_encoder_adapter_gating_middle_xy_add",
        O.GATE_MIDDLE_Y: "use_encoder_adapter_gating_middle_ia3_add"}


def _cfg(mode, r, nh, add=False, gs=1.0):
    c = SimpleNamespace(use_encoder_adapter_down_multihead=True, encoder_adapter_multihead_num_head=nh, adapter_down_dim=r,
                        use_encoder_adapter_gating_large_x_lowrank=False, use_encoder_adapter_gating_add=add,
                        use_encoder_gating_scaling=gs != 1.0, encoder_gating_scaling_factor=gs)
    setattr(c, FLAG[mode], True)
    return c


def _layer(mode, d, r, nh, add=False, gs=1.0):
    from vlpet_amd.encoder_pet import build_pet
    m = nn.Module()
    cfg = _cfg(mode, r, nh, add, gs)
    build_pet(m, cfg, d, ("ff",))
    m.ln = nn.LayerNorm(d)
    return m, cfg


def _gate_params(m, mode):
    pre = "encoder_ff_adapter_gating_"
    if mode == O.GATE_SMALL:
        lin = getattr(m, pre + "small_xy_cat"); return dict(w=lin.weight, b=lin.bias)
    if mode == O.GATE_MIDDLE_X:
        lin = getattr(m, pre + "middle_xy_add"); return dict(w=lin.weight, b=lin.bias)
    return dict(z=getattr(m, pre + "middle_ia3_add"))


def _rel(a, b):
    return float((a.detach().float().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-6))


@pytest.mark.parametrize("name,mode", [("k1_bart_small_d64_r8", O.GATE_SMALL), ("k1_bart_middlex_d64_r8", O.GATE_MIDDLE_X),
                                       ("k1_bart_middley_d64_r8", O.GATE_MIDDLE_Y)])
def test_gate_fixture_ffn_sublayer(name, mode):
    """Replay the FFN sublayer of the reference BartEncoderLayer recorded in the fixture through the product path."""
    from vlpet_amd.encoder_pet import apply_pet
    from vlpet_amd.tail import sublayer_tail
    z = np.load(os.path.join(G, name + ".npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files if z[k].dtype.kind == "f"}
    d, r, nh, rg, B, S = [int(v) for v in z["meta"]]
    m, cfg = _layer(mode, d, r, nh, bool(int(z["gating_add"])), float(z["gate_scale"]))
    rh = r // nh
    with torch.no_grad():
        for i, lin in enumerate(m.ff_adapter_multihead_down):
            lin.weight.copy_(g["ff_wd"][i * rh:(i + 1) * rh]); lin.bias.copy_(g["ff_bd"][i * rh:(i + 1) * rh])
        m.ff_adapter_multihead_up.weight.copy_(g["ff_wu"]); m.ff_adapter_multihead_up.bias.copy_(g["ff_bu"])
        gp = _gate_params(m, mode)
        if mode == O.GATE_MIDDLE_Y:
            gp["z"].copy_(g["ff_gz"])
        else:
            gp["w"].copy_(g["ff_gw"]); gp["b"].copy_(g["ff_gb"])
        m.ln.weight.copy_(g["ln2_w"]); m.ln.bias.copy_(g["ln2_b"])
    m.cuda()
    x1 = g["ff_x1"].cuda().requires_grad_(True)
    x2 = g["ff_x2"].cuda().requires_grad_(True)
    y = apply_pet(m, "ff", x1, x2, cfg)
    out = sublayer_tail(x1, y, m.ln, 0.0, False)
    assert _rel(out, g["out"]) <= 1e-4
    out.backward(g["dy"].cuda())
    tol = 2e-4
    assert _rel(x2.grad, g["ff_dx2"]) <= tol
    wd = torch.cat([l.weight.grad for l in m.ff_adapter_multihead_down]); bd = torch.cat([l.bias.grad for l in m.ff_adapter_multihead_down])
    assert _rel(wd, g["ff_dwd"]) <= tol and _rel(bd, g["ff_dbd"]) <= tol
    assert _rel(m.ff_adapter_multihead_up.weight.grad, g["ff_dwu"]) <= tol
    assert _rel(m.ff_adapter_multihead_up.bias.grad, g["ff_dbu"]) <= tol
    assert _rel(m.ln.weight.grad, g["ln2_dw"]) <= tol and _rel(m.ln.bias.grad, g["ln2_db"]) <= tol
    gp = _gate_params(m, mode)
    if mode == O.GATE_MIDDLE_Y:
        assert _rel(gp["z"].grad, g["ff_dgz"]) <= tol
    else:
        assert _rel(gp["w"].grad, g["ff_dgw"]) <= tol and _rel(gp["b"].grad, g["ff_dgb"]) <= tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("add,gs", [(False, 1.0), (True, 0.3)])
@pytest.mark.parametrize("mode", [O.GATE_SMALL, O.GATE_MIDDLE_X, O.GATE_MIDDLE_Y])
def test_gate_matches_oracle_d768(mode, add, gs, dtype):
    """BART-base width, S = 56 (20 text + 36 visual), r = 96, N_h = 4."""
    from vlpet_amd.encoder_pet import apply_pet
    torch.manual_seed(11)
    B, S, d, r, nh = 9, 56, 768, 96, 4
    m, cfg = _layer(mode, d, r, nh, add, gs)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    q = lambda t: t.to(dtype).float()
    x1, x2, dy = q(torch.randn(B, S, d)), q(torch.randn(B, S, d)), q(torch.randn(B, S, d))
    # oracle (fp32, CPU)
    P = {n: p.detach().clone().requires_grad_(True) for n, p in m.named_parameters()}
    x1r, x2r = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    dws = [P[f"ff_adapter_multihead_down.{i}.weight"] for i in range(nh)]
    dbs = [P[f"ff_adapter_multihead_down.{i}.bias"] for i in range(nh)]
    pre = "encoder_ff_adapter_gating_"
    if mode == O.GATE_SMALL:
        gate = dict(w=P[pre + "small_xy_cat.weight"], b=P[pre + "small_xy_cat.bias"])
    elif mode == O.GATE_MIDDLE_X:
        gate = dict(w=P[pre + "middle_xy_add.weight"], b=P[pre + "middle_xy_add.bias"])
    else:
        gate = dict(z=P[pre + "middle_ia3_add"])
    ref = O.encoder_adapter_gate(x1r, x2r, dws, dbs, P["ff_adapter_multihead_up.weight"], P["ff_adapter_multihead_up.bias"],
                                 gate, mode, add, 1.0, 1.0, gs)
    ref.backward(dy)
    # product
    m.cuda()
    X1, X2 = x1.cuda().to(dtype).requires_grad_(True), x2.cuda().to(dtype).requires_grad_(True)
    y = apply_pet(m, "ff", X1, X2, cfg)
    y.backward(dy.cuda().to(dtype))
    tol = 1e-3 if dtype == torch.float32 else 1e-2          # north_star tolerances
    assert _rel(y, ref.detach()) <= tol
    assert _rel(X2.grad, x2r.grad) <= tol
    if mode != O.GATE_MIDDLE_Y:
        assert _rel(X1.grad, x1r.grad) <= tol
    for n, p in m.named_parameters():
        if n.startswith("ln"):
            continue
        ref_g = P[n].grad
        if n.endswith("_xy_cat.bias") or n.endswith("_xy_add.bias"):
            # the gate's scalar bias gradient is sum_rows beta (heavy cancellation): judge it at the scale of the
            # gate's weight gradient sum_rows beta * x, which shares the same per-row factors
            scale = P[n.replace(".bias", ".weight")].grad.abs().max()
            err = float((p.grad.float().cpu() - ref_g).abs().max() / scale)
        else:
            err = _rel(p.grad, ref_g)
        assert err <= (tol if dtype == torch.float32 else 1e-2), (n, err)
