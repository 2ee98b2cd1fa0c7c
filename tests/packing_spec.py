"""MFMA-fragment packing of the (tiny, trainable) PET weights -- host-side specification.

The HIP kernels (csrc/) run every contraction in *swapped* form on
``v_mfma_f32_32x32x16_bf16``: the weight matrix is the A operand, the activation rows are the
B operand, so a lane (m = lane & 31, h = lane >> 5) always owns activation row ``m`` and the
MFMA result lands as "16 values of row m per lane".  Choosing which weight row each MFMA row
``i`` stands for (the permutations below) makes

  * the down-projection result, after bias + gelu_new + bf16 rounding, *already* the B operand
    of the up-projection (no LDS round trip, no cross-lane shuffle), and
  * the up-projection result 16 *contiguous* output features of row m per lane
    (32-byte bf16 stores, and the residual x2 is read in the same shape).

Because the weights are ~0.6 MB and change once per optimizer step, they are re-packed into
that fragment order (and cast to bf16, or split into bf16 hi/lo planes for the fp32 path) by a
small HIP kernel; this module is the numpy statement of the same layout.  It is used by the CPU
tests (tests/test_layout_model.py emulates a wavefront on top of it) and the GPU test compares the
HIP pack kernel's bytes with it.

Fragment = 64 lanes x 8 bf16 (1 KiB); element (lane, j) is at ``lane*8 + j``.
Lane (i = lane & 31, hh = lane >> 5) of an A fragment holds MFMA row ``i``, k-slots (hh, j).
"""
from __future__ import annotations

import numpy as np

FRAG = 512  # elements per fragment


def pad32(r: int) -> int:
    return (r + 31) // 32 * 32


def pi_d(ct, i):
    """bottleneck index c that MFMA row i of c-tile ct stands for (down-projection output)."""
    b, hp, a = i >> 3, (i >> 2) & 1, i & 3
    return 32 * ct + 16 * (b >> 1) + 8 * hp + 4 * (b & 1) + a


def pi_u(nt, i):
    """feature index f that MFMA row i of n-tile nt stands for (up-projection output)."""
    b, hp, a = i >> 3, (i >> 2) & 1, i & 3
    return 64 * (nt >> 1) + 32 * hp + 16 * (nt & 1) + 4 * b + a


def _lanes():
    lane = np.arange(64)
    return lane & 31, lane >> 5  # i, hh


def _gather(W, rows, cols, rmax, cmax):
    """W[rows, cols] with zeros where rows>=rmax or cols>=cmax (rows/cols broadcastable)."""
    rows, cols = np.broadcast_arrays(rows, cols)
    ok = (rows < rmax) & (cols < cmax)
    out = np.zeros(rows.shape, dtype=W.dtype)
    out[ok] = W[rows[ok], cols[ok]]
    return out


def pack_down(Wd: np.ndarray) -> np.ndarray:
    """Wd [r, d] -> fragments ordered (t, u, ct); slot (i,hh,j) = Wd[pi_d(ct,i)][64t+32hh+8u+j].

    Feeds  zT[c, m] = sum_k Wd[c,k] x[m,k]  where lane (m,h) holds x[m][64t+32h .. +32] of
    super-step t as four 8-element pieces u."""
    r, d = Wd.shape
    RT, T = pad32(r) // 32, d // 64
    i, hh = _lanes()
    j = np.arange(8)
    out = np.zeros((T, 4, RT, 64, 8), dtype=Wd.dtype)
    for t in range(T):
        for u in range(4):
            for ct in range(RT):
                rows = pi_d(ct, i)[:, None]
                cols = (64 * t + 32 * hh[:, None] + 8 * u + j[None, :])
                out[t, u, ct] = _gather(Wd, rows, cols, r, d)
    return out.reshape(-1, FRAG)


def pack_up(Wu: np.ndarray) -> np.ndarray:
    """Wu [d, r] -> fragments ordered (nt, ks); slot = Wu[pi_u(nt,i)][16ks+8hh+j]."""
    d, r = Wu.shape
    KT, NT = pad32(r) // 16, d // 32
    i, hh = _lanes()
    j = np.arange(8)
    out = np.zeros((NT, KT, 64, 8), dtype=Wu.dtype)
    for nt in range(NT):
        for ks in range(KT):
            rows = pi_u(nt, i)[:, None]
            cols = 16 * ks + 8 * hh[:, None] + j[None, :]
            out[nt, ks] = _gather(Wu, rows, cols, d, r)
    return out.reshape(-1, FRAG)


def pack_up_t(Wu: np.ndarray) -> np.ndarray:
    """Wu [d, r] -> fragments ordered (nt, e, ct); slot = Wu[64t+32hh+16v+8e+j][pi_d(ct,i)], nt=2t+v.

    Feeds the backward contraction over features  dzT[c, m] = sum_f Wu[f,c] dDelta[m,f]."""
    d, r = Wu.shape
    RT, NT = pad32(r) // 32, d // 32
    i, hh = _lanes()
    j = np.arange(8)
    out = np.zeros((NT, 2, RT, 64, 8), dtype=Wu.dtype)
    for nt in range(NT):
        t, v = nt >> 1, nt & 1
        for e in range(2):
            for ct in range(RT):
                rows = 64 * t + 32 * hh[:, None] + 16 * v + 8 * e + j[None, :]
                cols = pi_d(ct, i)[:, None]
                out[nt, e, ct] = _gather(Wu, rows, cols, d, r)
    return out.reshape(-1, FRAG)


def pack_down_t(Wd: np.ndarray) -> np.ndarray:
    """Wd [r, d] -> fragments ordered (nt, ks); slot = Wd[16ks+8hh+j][pi_u(nt,i)].

    Feeds  dxT[k, m] = sum_c Wd[c,k] dpre[m,c]."""
    r, d = Wd.shape
    KT, NT = pad32(r) // 16, d // 32
    i, hh = _lanes()
    j = np.arange(8)
    out = np.zeros((NT, KT, 64, 8), dtype=Wd.dtype)
    for nt in range(NT):
        for ks in range(KT):
            rows = 16 * ks + 8 * hh[:, None] + j[None, :]
            cols = pi_u(nt, i)[:, None]
            out[nt, ks] = _gather(Wd, rows, cols, r, d)
    return out.reshape(-1, FRAG)


# --------------------------------------------------------------------------- bf16 helpers
def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """round-to-nearest-even fp32 -> bf16 bit pattern (uint16)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    rounding = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + rounding) >> 16).astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def split_hi_lo(x: np.ndarray):
    """x ~= hi + lo with hi, lo bf16: the two planes the fp32 path feeds to the bf16 MFMA
    (hi*hi + hi*lo + lo*hi, fp32 accumulate)."""
    hi = f32_to_bf16_bits(x)
    lo = f32_to_bf16_bits(np.asarray(x, np.float32) - bf16_bits_to_f32(hi))
    return hi, lo


def packed_sizes(r: int, d: int):
    """number of fragments in each of the four packs of one (down [r,d], up [d,r]) pair."""
    RT, KT, T, NT = pad32(r) // 32, pad32(r) // 16, d // 64, d // 32
    return dict(down=T * 4 * RT, up=NT * KT, up_t=NT * 2 * RT, down_t=NT * KT)


# ======================================================================================= v3
# 16-row waves on v_mfma_f32_16x16x32_bf16 (8 waves per workgroup = 2 per SIMD, so MFMA, VALU and
# memory work of different waves overlap).  A fragment: lane (i = lane & 15, g = lane >> 4) holds
# MFMA row i, k-slots (g, j), j < 8.  C/D: register rho of lane (m = lane & 15, g) is row 4g + rho.
# A *stage* always moves 128 bytes of every activation row: FE = 64 features (bf16 IO, NS = 1) or
# 32 features (fp32 IO, NS = 2); KS = FE/32 MFMA k-steps (down phase), NQ = FE/16 n-tiles (up phase),
# LW = FE/4 contiguous output features per lane.
def stage_geom(NS: int):
    FE = 64 // NS
    return dict(FE=FE, KS=FE // 32, NQ=FE // 16, LW=FE // 4)


def c_of16(K, e, i):
    """bottleneck index of MFMA row i of c-tile (K, e): after the down projection lane (m, g) holds
    c = 32K + 8g + 4e + rho, i.e. exactly the B operand (k-slot (g, j = 4e + rho)) of up-projection k-step K."""
    return 32 * K + 8 * (i >> 2) + 4 * e + (i & 3)


def f_of16(su, q, i, NS):
    """output feature of MFMA row i of n-tile q of stage su: lane (m, g) ends with LW contiguous features."""
    G = stage_geom(NS)
    return G["FE"] * su + G["LW"] * (i >> 2) + 4 * q + (i & 3)


def _lanes16():
    lane = np.arange(64)
    return lane & 15, lane >> 4


def pack_down16(Wd: np.ndarray, NS: int = 1) -> np.ndarray:
    """Wd [r, d] -> fragments (stage, u, K, e): slot = Wd[c_of16(K,e,i)][FE*stage + 32u + 8g + j]."""
    r, d = Wd.shape
    RT = pad32(r) // 32
    G = stage_geom(NS)
    S = d // G["FE"]
    i, g = _lanes16()
    j = np.arange(8)
    out = np.zeros((S, G["KS"], RT, 2, 64, 8), dtype=Wd.dtype)
    for s in range(S):
        for u in range(G["KS"]):
            for K in range(RT):
                for e in range(2):
                    rows = c_of16(K, e, i)[:, None]
                    cols = G["FE"] * s + 32 * u + 8 * g[:, None] + j[None, :]
                    out[s, u, K, e] = _gather(Wd, rows, cols, r, d)
    return out.reshape(-1, FRAG)


def pack_up16(Wu: np.ndarray, NS: int = 1) -> np.ndarray:
    """Wu [d, r] -> fragments (stage, q, K): slot = Wu[f_of16(stage,q,i)][32K + 8g + j]."""
    d, r = Wu.shape
    RT = pad32(r) // 32
    G = stage_geom(NS)
    S = d // G["FE"]
    i, g = _lanes16()
    j = np.arange(8)
    out = np.zeros((S, G["NQ"], RT, 64, 8), dtype=Wu.dtype)
    for s in range(S):
        for q in range(G["NQ"]):
            for K in range(RT):
                rows = f_of16(s, q, i, NS)[:, None]
                cols = 32 * K + 8 * g[:, None] + j[None, :]
                out[s, q, K] = _gather(Wu, rows, cols, d, r)
    return out.reshape(-1, FRAG)


def pack_up_t16(Wu: np.ndarray, NS: int = 1) -> np.ndarray:
    """Wu [d, r] -> fragments (stage, e2, K, e): slot = Wu[FE*stage + LW*g + 8*e2 + j][c_of16(K,e,i)]."""
    d, r = Wu.shape
    RT = pad32(r) // 32
    G = stage_geom(NS)
    S = d // G["FE"]
    i, g = _lanes16()
    j = np.arange(8)
    E2 = G["LW"] // 8
    out = np.zeros((S, E2, RT, 2, 64, 8), dtype=Wu.dtype)
    for s in range(S):
        for e2 in range(E2):
            for K in range(RT):
                for e in range(2):
                    rows = G["FE"] * s + G["LW"] * g[:, None] + 8 * e2 + j[None, :]
                    cols = c_of16(K, e, i)[:, None]
                    out[s, e2, K, e] = _gather(Wu, rows, cols, d, r)
    return out.reshape(-1, FRAG)


def pack_down_t16(Wd: np.ndarray, NS: int = 1) -> np.ndarray:
    """Wd [r, d] -> fragments (stage, q, K): slot = Wd[32K + 8g + j][f_of16(stage,q,i)]."""
    r, d = Wd.shape
    RT = pad32(r) // 32
    G = stage_geom(NS)
    S = d // G["FE"]
    i, g = _lanes16()
    j = np.arange(8)
    out = np.zeros((S, G["NQ"], RT, 64, 8), dtype=Wd.dtype)
    for s in range(S):
        for q in range(G["NQ"]):
            for K in range(RT):
                rows = 32 * K + 8 * g[:, None] + j[None, :]
                cols = f_of16(s, q, i, NS)[:, None]
                out[s, q, K] = _gather(Wd, rows, cols, r, d)
    return out.reshape(-1, FRAG)


def row_tile_slot(row_in_tile, piece):
    """LDS slot (16-byte unit inside the 128-byte row segment) that holds `piece` of a staged row:
    XOR swizzle so that the 16x16x32 B-fragment reads (16 rows x same piece) are bank-conflict free."""
    return piece ^ ((row_in_tile >> 1) & 7)


# ======================================================================================= v4
# 32-row waves on v_mfma_f32_32x32x16_bf16 (half the LDS fragment traffic per flop of the 16x16x32
# form) with the v3 memory system (global_load_lds rings, 128 bytes of every row per stage).
# Lane (m = lane & 31, h = lane >> 5); FE = 64 / NS features per stage; MFMA k-slot (h, j) of k-step u
# is stage-local feature 16u + 8h + j; after the up phase a lane holds LW = FE/2 contiguous features.
def stage_geom4(NS: int):
    FE = 64 // NS
    return dict(FE=FE, KU=FE // 16, NV=FE // 32, LW=FE // 2, E4=FE // 16)


def f_of4(su, v, i, NS):
    G = stage_geom4(NS)
    b, hp, a = i >> 3, (i >> 2) & 1, i & 3
    return G["FE"] * su + G["LW"] * hp + 16 * v + 4 * b + a


def pack_down4(Wd: np.ndarray, NS: int = 1) -> np.ndarray:
    """Wd [r, d] -> fragments (stage, u, ct): slot = Wd[pi_d(ct,i)][FE*stage + 16u + 8hh + j]."""
    r, d = Wd.shape
    RT = pad32(r) // 32
    G = stage_geom4(NS)
    S = d // G["FE"]
    i, hh = _lanes()
    j = np.arange(8)
    out = np.zeros((S, G["KU"], RT, 64, 8), dtype=Wd.dtype)
    for s in range(S):
        for u in range(G["KU"]):
            for ct in range(RT):
                out[s, u, ct] = _gather(Wd, pi_d(ct, i)[:, None], G["FE"] * s + 16 * u + 8 * hh[:, None] + j[None, :], r, d)
    return out.reshape(-1, FRAG)


def pack_up4(Wu: np.ndarray, NS: int = 1) -> np.ndarray:
    """Wu [d, r] -> fragments (stage, v, ks): slot = Wu[f_of4(stage,v,i)][16ks + 8hh + j]."""
    d, r = Wu.shape
    KT = pad32(r) // 16
    G = stage_geom4(NS)
    S = d // G["FE"]
    i, hh = _lanes()
    j = np.arange(8)
    out = np.zeros((S, G["NV"], KT, 64, 8), dtype=Wu.dtype)
    for s in range(S):
        for v in range(G["NV"]):
            for ks in range(KT):
                out[s, v, ks] = _gather(Wu, f_of4(s, v, i, NS)[:, None], 16 * ks + 8 * hh[:, None] + j[None, :], d, r)
    return out.reshape(-1, FRAG)


def pack_up_t4(Wu: np.ndarray, NS: int = 1) -> np.ndarray:
    """Wu [d, r] -> fragments (stage, e, ct): slot = Wu[FE*stage + LW*hh + 8e + j][pi_d(ct,i)], e < LW/8."""
    d, r = Wu.shape
    RT = pad32(r) // 32
    G = stage_geom4(NS)
    S = d // G["FE"]
    i, hh = _lanes()
    j = np.arange(8)
    out = np.zeros((S, G["E4"], RT, 64, 8), dtype=Wu.dtype)
    for s in range(S):
        for e in range(G["E4"]):
            for ct in range(RT):
                out[s, e, ct] = _gather(Wu, G["FE"] * s + G["LW"] * hh[:, None] + 8 * e + j[None, :], pi_d(ct, i)[:, None], d, r)
    return out.reshape(-1, FRAG)


def pack_down_t4(Wd: np.ndarray, NS: int = 1) -> np.ndarray:
    """Wd [r, d] -> fragments (stage, v, ks): slot = Wd[16ks + 8hh + j][f_of4(stage,v,i)]."""
    r, d = Wd.shape
    KT = pad32(r) // 16
    G = stage_geom4(NS)
    S = d // G["FE"]
    i, hh = _lanes()
    j = np.arange(8)
    out = np.zeros((S, G["NV"], KT, 64, 8), dtype=Wd.dtype)
    for s in range(S):
        for v in range(G["NV"]):
            for ks in range(KT):
                out[s, v, ks] = _gather(Wd, 16 * ks + 8 * hh[:, None] + j[None, :], f_of4(s, v, i, NS)[:, None], r, d)
    return out.reshape(-1, FRAG)
