"""Wavefront-level model of the v3 kernels (16-row waves on v_mfma_f32_16x16x32) -- CPU, float64.

MFMA layout emulated: A row = lane&15, B col = lane&15, k-slot = (lane>>4, j); C/D: col = lane&15,
row = 4*(lane>>4) + reg.  Replays the index algebra of csrc/pet_fwd.hip / pet_bwd.hip on top of
tests/packing_spec.py (pack_*16): forward chain, backward chain (dz, dx) for both stage geometries
(bf16 IO: 64 features per stage, fp32 IO: 32)."""
import numpy as np
import pytest
import torch

from oracle import vlpet_oracle as O
import packing_spec as PK


def mfma16(A, B, C):
    """A,B: [64,8]; C: [64,4] -> D [64,4]."""
    Am = np.zeros((16, 32)); Bm = np.zeros((32, 16))
    for l in range(64):
        for j in range(8):
            Am[l & 15, 8 * (l >> 4) + j] = A[l, j]
            Bm[8 * (l >> 4) + j, l & 15] = B[l, j]
    Dm = Am @ Bm
    D = C.copy()
    for l in range(64):
        for reg in range(4):
            D[l, reg] += Dm[4 * (l >> 4) + reg, l & 15]
    return D


def gelu(x):
    return O.gelu_new(torch.from_numpy(np.asarray(x))).numpy()


def dgelu(x):
    t = torch.from_numpy(np.asarray(x)).requires_grad_(True)
    O.gelu_new(t).sum().backward()
    return t.grad.numpy()


def down_phase(x, W, bias, r, d, NS):
    """returns pre[K][e] as [64,4] (lane (m,g), reg rho <-> c = 32K + 8g + 4e + rho)"""
    G = PK.stage_geom(NS)
    RT = PK.pad32(r) // 32
    pk = PK.pack_down16(W, NS).reshape(d // G["FE"], G["KS"], RT, 2, 64, 8)
    acc = [[np.zeros((64, 4)) for _ in range(2)] for _ in range(RT)]
    for s in range(d // G["FE"]):
        for u in range(G["KS"]):
            b = np.zeros((64, 8))
            for l in range(64):
                m, g = l & 15, l >> 4
                k0 = G["FE"] * s + 32 * u + 8 * g
                b[l] = x[m, k0:k0 + 8]
            for K in range(RT):
                for e in range(2):
                    acc[K][e] = mfma16(pk[s, u, K, e], b, acc[K][e])
    for K in range(RT):
        for e in range(2):
            for l in range(64):
                for rho in range(4):
                    c = 32 * K + 8 * (l >> 4) + 4 * e + rho
                    acc[K][e][l, rho] += bias[c] if c < r else 0.0
    return acc


def to_bfrags(vals):
    """B fragment of k-step K: slot j = 4e + rho."""
    return [np.concatenate([vals[K][0], vals[K][1]], axis=1) for K in range(len(vals))]


@pytest.mark.parametrize("NS,d,r,rg", [(1, 64, 8, 8), (1, 128, 40, 96), (2, 64, 24, 8)])
def test_chain16(NS, d, r, rg):
    G = PK.stage_geom(NS)
    FE, NQ, LW = G["FE"], G["NQ"], G["LW"]
    rng = np.random.default_rng(1)
    Mr = 16
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
    S = d // FE
    preA = down_phase(x2, wd, bd, r, d, NS); preG = down_phase(x1, wgd, bgd, rg, d, NS)
    zA = to_bfrags([[gelu(p) for p in pe] for pe in preA]); zG = to_bfrags([[gelu(p) for p in pe] for pe in preG])
    pu = PK.pack_up16(wu, NS).reshape(S, NQ, RT, 64, 8); pgu = PK.pack_up16(wgu, NS).reshape(S, NQ, RTg, 64, 8)
    E2 = LW // 8
    put = PK.pack_up_t16(wu, NS).reshape(S, E2, RT, 2, 64, 8); pgut = PK.pack_up_t16(wgu, NS).reshape(S, E2, RTg, 2, 64, 8)
    y = np.zeros((Mr, d)); DH = np.zeros((Mr, d))
    dzA = [[np.zeros((64, 4)) for _ in range(2)] for _ in range(RT)]
    dzG = [[np.zeros((64, 4)) for _ in range(2)] for _ in range(RTg)]
    for su in range(S):
        dh = np.zeros((64, LW)); dq = np.zeros((64, LW))
        for q in range(NQ):
            aA = np.zeros((64, 4)); aG = np.zeros((64, 4))
            for K in range(RT):
                aA = mfma16(pu[su, q, K], zA[K], aA)
            for K in range(RTg):
                aG = mfma16(pgu[su, q, K], zG[K], aG)
            for l in range(64):
                m, g = l & 15, l >> 4
                f0 = FE * su + LW * g + 4 * q
                delta = aA[l] + bu[f0:f0 + 4]
                qq = aG[l] + bgu[f0:f0 + 4]
                hv = s2 * x2[m, f0:f0 + 4] + sd * delta
                gt = 1 / (1 + np.exp(-qq))
                y[m, f0:f0 + 4] = gs * hv * gt
                dyp = gs * dy[m, f0:f0 + 4]
                dh[l, 4 * q:4 * q + 4] = dyp * gt
                dq[l, 4 * q:4 * q + 4] = dyp * hv * gt * (1 - gt)
                DH[m, f0:f0 + 4] = dyp * gt
        for e2 in range(E2):
            for K in range(RT):
                for e in range(2):
                    dzA[K][e] = mfma16(put[su, e2, K, e], sd * dh[:, 8 * e2:8 * e2 + 8], dzA[K][e])
            for K in range(RTg):
                for e in range(2):
                    dzG[K][e] = mfma16(pgut[su, e2, K, e], dq[:, 8 * e2:8 * e2 + 8], dzG[K][e])
    np.testing.assert_allclose(y, y_ref.numpy(), rtol=1e-9, atol=1e-9)

    dpA = [[dzA[K][e] * dgelu(preA[K][e]) for e in range(2)] for K in range(RT)]
    dpG = [[dzG[K][e] * dgelu(preG[K][e]) for e in range(2)] for K in range(RTg)]
    fA, fG = to_bfrags(dpA), to_bfrags(dpG)
    pdt = PK.pack_down_t16(wd, NS).reshape(S, NQ, RT, 64, 8); pgdt = PK.pack_down_t16(wgd, NS).reshape(S, NQ, RTg, 64, 8)
    dx1 = np.zeros((Mr, d)); dx2 = np.zeros((Mr, d))
    for su in range(S):
        for q in range(NQ):
            a = np.zeros((64, 4)); gg = np.zeros((64, 4))
            for K in range(RT):
                a = mfma16(pdt[su, q, K], fA[K], a)
            for K in range(RTg):
                gg = mfma16(pgdt[su, q, K], fG[K], gg)
            for l in range(64):
                m, g = l & 15, l >> 4
                f0 = FE * su + LW * g + 4 * q
                dx2[m, f0:f0 + 4] = s2 * DH[m, f0:f0 + 4] + a[l]
                dx1[m, f0:f0 + 4] = gg[l]
    np.testing.assert_allclose(dx1, g_ref["x1"].numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(dx2, g_ref["x2"].numpy(), rtol=1e-9, atol=1e-9)

    # row-major side products (lane (m,g) of tile (K,e) writes 4 contiguous c at 32K + 8g + 4e)
    def rowmajor(vals, rr):
        out = np.zeros((Mr, PK.pad32(rr)))
        for K in range(len(vals)):
            for e in range(2):
                for l in range(64):
                    c0 = 32 * K + 8 * (l >> 4) + 4 * e
                    out[l & 15, c0:c0 + 4] = vals[K][e][l]
        return out
    DPA = rowmajor(dpA, r)
    np.testing.assert_allclose((DPA.T @ x2)[:r], g_ref["wd"].numpy(), rtol=1e-9, atol=1e-9)
    Z = rowmajor([[gelu(p) for p in pe] for pe in preA], r)
    np.testing.assert_allclose(sd * (DH.T @ Z)[:, :r], g_ref["wu"].numpy(), rtol=1e-9, atol=1e-9)


def test_row_tile_swizzle_is_conflict_free():
    """ds_read_b128 services a wave in four 16-lane groups; within each, the 16-byte slots (mod 16 slots =
    one 256-byte bank row) must be distinct for the B-fragment read pattern (lane (m,g) reads piece 4u+g)."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for u in range(2):
        for grp in groups:
            slots = set()
            for l in grp:
                m, g = l & 15, l >> 4
                unit = m * 8 + PK.row_tile_slot(m, 4 * u + g)       # 16-byte unit index in the tile
                slots.add(unit % 16)
            assert len(slots) == 16
