"""Wavefront-level model of the HIP kernels' data flow (CPU).

Emulates v_mfma_f32_32x32x16 per its documented register layout (A row = lane&31, B col =
lane&31, k-slot = (lane>>5, j); D: col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)) and
replays, lane by lane, exactly the index algebra csrc/ uses on top of tests/packing_spec.py:
forward chain, backward chain (dz, dx) and the MFMA-transpose weight-gradient.  Values are kept
in float64 so the comparison with the oracle is about layout only."""
import numpy as np
import pytest
import torch

from oracle import vlpet_oracle as O
import packing_spec as PK

LANE = np.arange(64)
M_ = LANE & 31
H_ = LANE >> 5


def mfma32(A, B, C):
    """A,B: [64,8]; C: [64,16] -> D [64,16]."""
    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for l in range(64):
        for j in range(8):
            Am[l & 31, 8 * (l >> 5) + j] = A[l, j]
            Bm[8 * (l >> 5) + j, l & 31] = B[l, j]
    Dm = Am @ Bm
    D = C.copy()
    for l in range(64):
        for reg in range(16):
            D[l, reg] += Dm[(reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5), l & 31]
    return D


def gelu(x):
    return O.gelu_new(torch.from_numpy(np.asarray(x))).numpy()


def dgelu(x):
    t = torch.from_numpy(np.asarray(x)).requires_grad_(True)
    O.gelu_new(t).sum().backward()
    return t.grad.numpy()


def x_frag(x, row0, t, u):
    """B fragment of super-step t, piece u: lane (m,h) holds x[row0+m][64t+32h+8u .. +8]."""
    out = np.zeros((64, 8))
    for l in range(64):
        m, h = l & 31, l >> 5
        out[l] = x[row0 + m, 64 * t + 32 * h + 8 * u: 64 * t + 32 * h + 8 * u + 8]
    return out


def c_of(ct, l, reg):
    """bottleneck index held by (lane, reg) of c-tile ct after the down projection."""
    return PK.pi_d(ct, 8 * (reg >> 2) + 4 * (l >> 5) + (reg & 3))


def down_phase(x, row0, pack, bias, r, d):
    RT, T = PK.pad32(r) // 32, d // 64
    pk = pack.reshape(T, 4, RT, 64, 8)
    acc = [np.zeros((64, 16)) for _ in range(RT)]
    for t in range(T):
        for u in range(4):
            b = x_frag(x, row0, t, u)
            for ct in range(RT):
                acc[ct] = mfma32(pk[t, u, ct], b, acc[ct])
    # bias (lane dependent)
    pre = []
    for ct in range(RT):
        p = acc[ct].copy()
        for l in range(64):
            for reg in range(16):
                c = c_of(ct, l, reg)
                p[l, reg] += bias[c] if c < r else 0.0
        pre.append(p)
    return pre


def to_bfrags(vals):
    """[RT] x [64,16] (D layout of the down phase) -> B fragments per k-step of 16:
    frag[2ct+s][lane][j] = vals[ct][lane][4*(2s + (j>>2)) + (j&3)]."""
    frags = []
    for ct in range(len(vals)):
        for s in range(2):
            f = np.zeros((64, 8))
            for j in range(8):
                f[:, j] = vals[ct][:, 4 * (2 * s + (j >> 2)) + (j & 3)]
            frags.append(f)
    return frags


def up_tile(zf, pack_up, nt, KT):
    pk = pack_up.reshape(-1, KT, 64, 8)
    acc = np.zeros((64, 16))
    for ks in range(KT):
        acc = mfma32(pk[nt, ks], zf[ks], acc)
    return acc


def feat0(nt, l):
    return 64 * (nt >> 1) + 32 * (l >> 5) + 16 * (nt & 1)


@pytest.mark.parametrize("d,r,rg", [(64, 8, 8), (128, 40, 96)])
def test_forward_and_backward_chain(d, r, rg):
    rng = np.random.default_rng(0)
    Mrows = 32
    x1 = rng.standard_normal((Mrows, d)); x2 = rng.standard_normal((Mrows, d))
    wd = rng.standard_normal((r, d)) * 0.2; bd = rng.standard_normal(r) * 0.2
    wu = rng.standard_normal((d, r)) * 0.2; bu = rng.standard_normal(d) * 0.2
    wgd = rng.standard_normal((rg, d)) * 0.2; bgd = rng.standard_normal(rg) * 0.2
    wgu = rng.standard_normal((d, rg)) * 0.2; bgu = rng.standard_normal(d) * 0.2
    dy = rng.standard_normal((Mrows, d))
    tt = lambda a: torch.from_numpy(a)
    y_ref, g_ref = O.k1_fwd_bwd(tt(x1), tt(x2), tt(wd), tt(bd), tt(wu), tt(bu), tt(wgd), tt(bgd), tt(wgu),
                                tt(bgu), tt(dy), gate_scale=0.7, delta_scale=1.5, x2_scale=0.9)
    gs, sd, s2 = 0.7, 1.5, 0.9
    KT, KTg, NT = PK.pad32(r) // 16, PK.pad32(rg) // 16, d // 32
    RT, RTg = PK.pad32(r) // 32, PK.pad32(rg) // 32

    # ---- forward
    preA = down_phase(x2, 0, PK.pack_down(wd), bd, r, d)
    preG = down_phase(x1, 0, PK.pack_down(wgd), bgd, rg, d)
    zA = to_bfrags([gelu(p) for p in preA]); zG = to_bfrags([gelu(p) for p in preG])
    pu, pgu = PK.pack_up(wu), PK.pack_up(wgu)
    y = np.zeros((Mrows, d)); DH = np.zeros((Mrows, d)); DQ = np.zeros((Mrows, d))
    dzA = [np.zeros((64, 16)) for _ in range(RT)]; dzG = [np.zeros((64, 16)) for _ in range(RTg)]
    put, pgut = PK.pack_up_t(wu).reshape(NT, 2, RT, 64, 8), PK.pack_up_t(wgu).reshape(NT, 2, RTg, 64, 8)
    for nt in range(NT):
        aA = up_tile(zA, pu, nt, KT); aG = up_tile(zG, pgu, nt, KTg)
        dh = np.zeros((64, 16)); dq = np.zeros((64, 16))
        for l in range(64):
            f0 = feat0(nt, l); m = l & 31
            delta = aA[l] + bu[f0:f0 + 16]
            q = aG[l] + bgu[f0:f0 + 16]
            hh = s2 * x2[m, f0:f0 + 16] + sd * delta
            g = 1 / (1 + np.exp(-q))
            y[m, f0:f0 + 16] = gs * hh * g
            dyp = gs * dy[m, f0:f0 + 16]
            dh[l] = dyp * g
            dq[l] = dyp * hh * g * (1 - g)
            DH[m, f0:f0 + 16] = dh[l]; DQ[m, f0:f0 + 16] = dq[l]
        # contraction over features: B fragments e=0,1 are regs [8e, 8e+8)
        for e in range(2):
            for ct in range(RT):
                dzA[ct] = mfma32(put[nt, e, ct], sd * dh[:, 8 * e:8 * e + 8], dzA[ct])
            for ct in range(RTg):
                dzG[ct] = mfma32(pgut[nt, e, ct], dq[:, 8 * e:8 * e + 8], dzG[ct])
    np.testing.assert_allclose(y, y_ref.numpy(), rtol=1e-9, atol=1e-9)

    # ---- backward: dpre = dz * gelu'(pre), dx tiles
    dpA = [dzA[ct] * dgelu(preA[ct]) for ct in range(RT)]
    dpG = [dzG[ct] * dgelu(preG[ct]) for ct in range(RTg)]
    fA, fG = to_bfrags(dpA), to_bfrags(dpG)
    pdt, pgdt = PK.pack_down_t(wd), PK.pack_down_t(wgd)
    dx1 = np.zeros((Mrows, d)); dx2 = np.zeros((Mrows, d))
    for nt in range(NT):
        a = up_tile(fA, pdt, nt, KT); g = up_tile(fG, pgdt, nt, KTg)
        for l in range(64):
            f0 = feat0(nt, l); m = l & 31
            dx2[m, f0:f0 + 16] = s2 * DH[m, f0:f0 + 16] + a[l]
            dx1[m, f0:f0 + 16] = g[l]
    np.testing.assert_allclose(dx1, g_ref["x1"].numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(dx2, g_ref["x2"].numpy(), rtol=1e-9, atol=1e-9)

    # ---- row-major side products written for the weight-gradient kernel
    def rowmajor(vals, rr):
        out = np.zeros((Mrows, PK.pad32(rr)))
        for ct in range(len(vals)):
            for l in range(64):
                for reg in range(16):
                    out[l & 31, c_of(ct, l, reg)] = vals[ct][l, reg]
        return out
    Z, ZG = rowmajor([gelu(p) for p in preA], r), rowmajor([gelu(p) for p in preG], rg)
    DPA, DPG = rowmajor(dpA, r), rowmajor(dpG, rg)
    np.testing.assert_allclose((DPA.T @ x2)[:r], g_ref["wd"].numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(DPA.sum(0)[:r], g_ref["bd"].numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(sd * (DH.T @ Z)[:, :r], g_ref["wu"].numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(sd * DH.sum(0), g_ref["bu"].numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose((DPG.T @ x1)[:rg], g_ref["wgd"].numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose((DQ.T @ ZG)[:, :rg], g_ref["wgu"].numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(DQ.sum(0), g_ref["bgu"].numpy(), rtol=1e-9, atol=1e-9)

    # ---- MFMA-transpose weight gradient:  Out[c, n] = sum_m P[m,c] X[m,n]
    def nat_frag(A, ks):      # natural A fragment: lane (m,hh) holds A[m][16ks+8hh .. +8]
        f = np.zeros((64, 8))
        for l in range(64):
            f[l] = A[l & 31, 16 * ks + 8 * (l >> 5): 16 * ks + 8 * (l >> 5) + 8]
        return f
    I0 = np.zeros((64, 8)); I1 = np.zeros((64, 8))
    for l in range(64):
        for j in range(8):
            I0[l, j] = 1.0 if (l & 31) == 8 * (l >> 5) + j else 0.0
            I1[l, j] = 1.0 if (l & 31) == 16 + 8 * (l >> 5) + j else 0.0

    def transpose_tile(A, tile):   # -> [64,16]: lane (col c', h) regs <-> rows m = 8b+4h+a
        T = mfma32(nat_frag(A, 2 * tile), I0, np.zeros((64, 16)))
        return mfma32(nat_frag(A, 2 * tile + 1), I1, T)
    P, X = DPA, x2
    out = np.zeros((PK.pad32(r), d))
    for ct in range(RT):
        PT = transpose_tile(P, ct)
        for nt in range(d // 32):
            XT = transpose_tile(X, nt)
            acc = np.zeros((64, 16))
            for e in range(2):
                acc = mfma32(PT[:, 8 * e:8 * e + 8], XT[:, 8 * e:8 * e + 8], acc)
            for l in range(64):
                for reg in range(16):
                    out[32 * ct + (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5), 32 * nt + (l & 31)] = acc[l, reg]
    np.testing.assert_allclose(out[:r], g_ref["wd"].numpy(), rtol=1e-9, atol=1e-9)


def test_pack_sizes_and_padding():
    wd = np.arange(8 * 64, dtype=np.float32).reshape(8, 64)
    p = PK.pack_down(wd)
    assert p.shape == (PK.packed_sizes(8, 64)["down"], 512)
    # every weight appears exactly once, the rest is zero padding
    assert np.count_nonzero(p) == np.count_nonzero(wd)
    assert np.isclose(p.sum(), wd.sum())
    wu = np.arange(64 * 8, dtype=np.float32).reshape(64, 8) + 1
    for f in (PK.pack_up, PK.pack_up_t):
        q = f(wu)
        assert np.count_nonzero(q) == wu.size and np.isclose(q.sum(), wu.sum())
    q = PK.pack_down_t(wd + 1)
    assert np.count_nonzero(q) == wd.size


def test_bf16_split():
    x = np.random.default_rng(1).standard_normal(1000).astype(np.float32)
    hi, lo = PK.split_hi_lo(x)
    rec = PK.bf16_bits_to_f32(hi) + PK.bf16_bits_to_f32(lo)
    assert np.max(np.abs(rec - x) / np.abs(x)) < 2.0 ** -15
    t = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(hi, t)
