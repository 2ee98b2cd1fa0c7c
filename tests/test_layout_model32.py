"""Wavefront-level model of the v4 kernels (32-row waves, v_mfma_f32_32x32x16, 128 bytes of every row
per stage) on top of tests/packing_spec.py pack_*4 -- CPU, float64, both stage geometries."""
import numpy as np
import pytest
import torch

from oracle import vlpet_oracle as O
import packing_spec as PK
from test_layout_model import mfma32, gelu, dgelu


def c_of(ct, l, reg):
    return PK.pi_d(ct, 8 * (reg >> 2) + 4 * (l >> 5) + (reg & 3))


def down_phase(x, W, bias, r, d, NS):
    G = PK.stage_geom4(NS)
    RT = PK.pad32(r) // 32
    S = d // G["FE"]
    pk = PK.pack_down4(W, NS).reshape(S, G["KU"], RT, 64, 8)
    acc = [np.zeros((64, 16)) for _ in range(RT)]
    for s in range(S):
        for u in range(G["KU"]):
            b = np.zeros((64, 8))
            for l in range(64):
                k0 = G["FE"] * s + 16 * u + 8 * (l >> 5)
                b[l] = x[l & 31, k0:k0 + 8]
            for ct in range(RT):
                acc[ct] = mfma32(pk[s, u, ct], b, acc[ct])
    for ct in range(RT):
        for l in range(64):
            for reg in range(16):
                c = c_of(ct, l, reg)
                acc[ct][l, reg] += bias[c] if c < r else 0.0
    return acc


def to_bfrags(vals):
    return [vals[ct][:, 8 * s:8 * s + 8] for ct in range(len(vals)) for s in range(2)]


@pytest.mark.parametrize("NS,d,r,rg", [(1, 64, 8, 8), (1, 128, 40, 96), (2, 64, 24, 8)])
def test_chain32(NS, d, r, rg):
    G = PK.stage_geom4(NS)
    FE, NV, LW, E4 = G["FE"], G["NV"], G["LW"], G["E4"]
    rng = np.random.default_rng(2)
    Mr = 32
    x1 = rng.standard_normal((Mr, d)); x2 = rng.standard_normal((Mr, d)); dy = rng.standard_normal((Mr, d))
    wd = rng.standard_normal((r, d)) * 0.2; bd = rng.standard_normal(r) * 0.2
    wu = rng.standard_normal((d, r)) * 0.2; bu = rng.standard_normal(d) * 0.2
    wgd = rng.standard_normal((rg, d)) * 0.2; bgd = rng.standard_normal(rg) * 0.2
    wgu = rng.standard_normal((d, rg)) * 0.2; bgu = rng.standard_normal(d) * 0.2
    tt = torch.from_numpy
    gs, sd, s2 = 0.7, 1.5, 0.9
    y_ref, g_ref = O.k1_fwd_bwd(tt(x1), tt(x2), tt(wd), tt(bd), tt(wu), tt(bu), tt(wgd), tt(bgd), tt(wgu), tt(bgu),
                                tt(dy), gate_scale=gs, delta_scale=sd, x2_scale=s2)
    RT, RTg = PK.pad32(r) // 32, PK.pad32(rg) // 32
    KT, KTg = 2 * RT, 2 * RTg
    S = d // FE
    preA = down_phase(x2, wd, bd, r, d, NS); preG = down_phase(x1, wgd, bgd, rg, d, NS)
    zA = to_bfrags([gelu(p) for p in preA]); zG = to_bfrags([gelu(p) for p in preG])
    pu = PK.pack_up4(wu, NS).reshape(S, NV, KT, 64, 8); pgu = PK.pack_up4(wgu, NS).reshape(S, NV, KTg, 64, 8)
    put = PK.pack_up_t4(wu, NS).reshape(S, E4, RT, 64, 8); pgut = PK.pack_up_t4(wgu, NS).reshape(S, E4, RTg, 64, 8)
    y = np.zeros((Mr, d)); DH = np.zeros((Mr, d))
    dzA = [np.zeros((64, 16)) for _ in range(RT)]; dzG = [np.zeros((64, 16)) for _ in range(RTg)]
    for su in range(S):
        dh = np.zeros((64, LW)); dq = np.zeros((64, LW))
        for v in range(NV):
            aA = np.zeros((64, 16)); aG = np.zeros((64, 16))
            for ks in range(KT):
                aA = mfma32(pu[su, v, ks], zA[ks], aA)
            for ks in range(KTg):
                aG = mfma32(pgu[su, v, ks], zG[ks], aG)
            for l in range(64):
                m, h = l & 31, l >> 5
                f0 = FE * su + LW * h + 16 * v
                hv = s2 * x2[m, f0:f0 + 16] + sd * (aA[l] + bu[f0:f0 + 16])
                gt = 1 / (1 + np.exp(-(aG[l] + bgu[f0:f0 + 16])))
                y[m, f0:f0 + 16] = gs * hv * gt
                dyp = gs * dy[m, f0:f0 + 16]
                dh[l, 16 * v:16 * v + 16] = dyp * gt
                dq[l, 16 * v:16 * v + 16] = dyp * hv * gt * (1 - gt)
                DH[m, f0:f0 + 16] = dyp * gt
        for e in range(E4):
            for ct in range(RT):
                dzA[ct] = mfma32(put[su, e, ct], sd * dh[:, 8 * e:8 * e + 8], dzA[ct])
            for ct in range(RTg):
                dzG[ct] = mfma32(pgut[su, e, ct], dq[:, 8 * e:8 * e + 8], dzG[ct])
    np.testing.assert_allclose(y, y_ref.numpy(), rtol=1e-9, atol=1e-9)
    dpA = [dzA[ct] * dgelu(preA[ct]) for ct in range(RT)]; dpG = [dzG[ct] * dgelu(preG[ct]) for ct in range(RTg)]
    fA, fG = to_bfrags(dpA), to_bfrags(dpG)
    pdt = PK.pack_down_t4(wd, NS).reshape(S, NV, KT, 64, 8); pgdt = PK.pack_down_t4(wgd, NS).reshape(S, NV, KTg, 64, 8)
    dx1 = np.zeros((Mr, d)); dx2 = np.zeros((Mr, d))
    for su in range(S):
        for v in range(NV):
            a = np.zeros((64, 16)); gg = np.zeros((64, 16))
            for ks in range(KT):
                a = mfma32(pdt[su, v, ks], fA[ks], a)
            for ks in range(KTg):
                gg = mfma32(pgdt[su, v, ks], fG[ks], gg)
            for l in range(64):
                m, h = l & 31, l >> 5
                f0 = FE * su + LW * h + 16 * v
                dx2[m, f0:f0 + 16] = s2 * DH[m, f0:f0 + 16] + a[l]
                dx1[m, f0:f0 + 16] = gg[l]
    np.testing.assert_allclose(dx1, g_ref["x1"].numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(dx2, g_ref["x2"].numpy(), rtol=1e-9, atol=1e-9)


def test_row_tile_swizzle32_is_conflict_free():
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for u in range(4):
        for grp in groups:
            slots = set()
            for l in grp:
                m, h = l & 31, l >> 5
                slots.add((m * 8 + PK.row_tile_slot(m, 2 * u + h)) % 16)
            assert len(slots) == 16
