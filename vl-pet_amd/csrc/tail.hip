// K5: the sublayer tail that follows every PET op on the BART path,
//
//     out = LayerNorm( x1 + dropout(y) )            (my_transformers/modeling_bart.py:1259-1261, 1375-1377;
//                                                    decoder: :1489-1491, 1513-1515, 1527-1529)
//     out = x1 + dropout(y)                         (T5: my_transformers/modeling_t5.py:408, 824)  [norm = 0]
//
// as ONE pass over [M, d] (read y, read x1, write out [+ h for the backward]) instead of the three eager
// passes (dropout, add, LayerNorm); the backward is one pass too (read dout, h; write dx1 [, dy]) and
// regenerates the dropout mask from the same counter-based generator, so no mask is stored.
//
// HBM-bound row kernel: one wave per row, a lane owns 16-byte pieces lane, lane+64, ... of the row (whole
// 128-byte lines per 8 lanes), row statistics by two wave reductions (mean, then centred variance),
// many rows in flight per CU through occupancy (<= 64 VGPRs at d = 768 bf16).
// Dropout: Philox-4x32 (7 rounds; Salmon et al., "Parallel random numbers: as easy as 1, 2, 3") keyed by
// the call's 64-bit seed, counter = index of the 8-element group; element j of the group keeps iff its
// 16-bit lane >= round(p * 65536).  The mask depends on (seed, element index) only -- not on the IO dtype.
#include "common.h"
#include "kernels.h"
#include "rowops.h"
#include "rng.h"
#include <cstdlib>

constexpr int TAIL_WAVES = 4;

// Keep flags of this lane's NP pieces (piece k = elements row_e0 + E (lane + 64 k) ..) of one row; bit j of kb[k] = element j kept.
// E = 8: a piece is one generator group.  E = 4 (fp32 IO, or the 8-byte bf16 pieces): a piece is HALF a group and lanes 2j, 2j + 1
// own the two halves of the same groups -- instead of one generator call per piece in both lanes (NP calls), the even lane draws
// the groups of pieces k = 0, 2, .. and the odd lane those of k = 1, 3, .. (ceil(NP / 2) calls per lane) and each fetches the other's
// word with a quad-permute.  The mask is the same function of (seed, element index) either way.
template <int NP, int E>
__device__ __forceinline__ void row_keep_bits(int64_t row_e0, int lane, int pieces, uint64_t seed, uint32_t thr, uint32_t (&kb)[NP]) {
    if constexpr (E == 8) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = lane + 64 * k;
            kb[k] = p < pieces ? keep8((row_e0 + (int64_t)p * 8) >> 3, seed, thr) : 0u;
        }
    } else {
        static_assert(E == 4, "pieces of 4 or 8 elements");
        const int par = lane & 1, half = 4 * par;
#pragma unroll
        for (int c = 0; c < (NP + 1) / 2; ++c) {
            int k = 2 * c + par;
            if (k >= NP) k = NP - 1;                               // (odd NP: the odd lane repeats the last even piece's group)
            const int p = (lane & ~1) + 64 * k;                    // even piece of the pair = first element of the group
            const uint32_t mine = keep8((row_e0 + (int64_t)p * 4) >> 3, seed, thr);
            const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);     // lane ^ 1 (quad_perm [1,0,3,2])
            // piece 2c belongs to the even lane's call, piece 2c + 1 to the odd lane's
            kb[2 * c] = ((par == 0 ? mine : other) >> half) & 0xfu;
            if (2 * c + 1 < NP) kb[2 * c + 1] = ((par == 1 ? mine : other) >> half) & 0xfu;
        }
    }
}

// POST (NORM only): out = LayerNorm(dropout(y)) + x1 -- the residual joins AFTER the norm (visual projectors:
// src/modeling_bart.py:298-299 then :324-325); statistics and the saved pre-norm tensor are those of dropout(y) alone.
// RMS2 (plain residual tail only, round 5): the wave that has just formed the row x1 + dropout(y) also applies the NEXT sublayer's RMS norm
// to it and writes both -- T5's pre-norm stream otherwise reads the sum back in a second launch (2 of that launch's 2 units + the launch).
template <int E> __device__ __forceinline__ void load_f32_vec(const float* src, float* v) {      // src 16-byte aligned, E % 4 == 0
#pragma unroll
    for (int j = 0; j < E; j += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + j);
        v[j] = t[0]; v[j + 1] = t[1]; v[j + 2] = t[2]; v[j + 3] = t[3];
    }
}

// FULL (round 5): the row is exactly NP x 64 pieces (d = 768 with 8-byte pieces, 1024 with either): every "is this piece inside the row"
// test folds away and -- what matters -- the row loads become UNCONDITIONAL.  With loads under a branch hipcc cannot count how many
// requests are younger than the row it is about to use, so it guarded that use with s_waitcnt vmcnt(0): the row requested one ahead
// was awaited before the current one was reduced, i.e. the prefetch hid nothing (found in the ISA, tools/isa_waits.py).
template <typename IO, int NP, bool NORM, bool POST = false, int PB = 16, bool RMS2 = false, bool FULL = false>
__global__ __launch_bounds__(TAIL_WAVES * 64) void tail_fwd_kernel(TailArgs a) {
    static_assert(!POST || NORM, "post-norm residual needs the norm");
    static_assert(!RMS2 || !NORM, "the second (normalised) output belongs to the plain residual tail");
    using P = Piece<IO, PB>;
    using Raw = typename P::Raw;
    constexpr int E = P::E;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int d = FULL ? NP * 64 * E : a.d, pieces = FULL ? NP * 64 : d / E;
    const uint32_t thr = a.thr;
    const uint64_t seed = thr ? vlpet_eff_seed(a.seed, a.seed_ctr) : 0;      // (one scalar load, before the row loop)
    const float scale = a.keep_scale;
    float gam[NORM ? NP : 1][E], bet[NORM ? NP : 1][E];
    if constexpr (NORM) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = lane + 64 * k;
            if constexpr (FULL) {       // 16-byte vectors, no branch (the launcher checked the alignment); element by element they were 24 dword loads
                const float bm = a.beta ? 1.f : 0.f;
                const float* bsrc = a.beta ? a.beta : a.gamma;
                load_f32_vec<E>(a.gamma + p * E, gam[k]);
                load_f32_vec<E>(bsrc + p * E, bet[k]);
#pragma unroll
                for (int j = 0; j < E; ++j) bet[k][j] *= bm;
            } else {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    gam[k][j] = p < pieces ? a.gamma[p * E + j] : 0.f;
                    bet[k][j] = (p < pieces && a.beta) ? a.beta[p * E + j] : 0.f;
                }
            }
        }
    }
    float gam2[RMS2 ? NP : 1][E];
    if constexpr (RMS2) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = lane + 64 * k;
            if constexpr (FULL) load_f32_vec<E>(a.gamma2 + p * E, gam2[k]);
            else {
#pragma unroll
                for (int j = 0; j < E; ++j) gam2[k][j] = p < pieces ? a.gamma2[p * E + j] : 0.f;
            }
        }
    }
    const uint8_t* y = reinterpret_cast<const uint8_t*>(a.y);
    const uint8_t* x1 = reinterpret_cast<const uint8_t*>(a.x1);
    const uint8_t* ysrc = y ? y : x1;               // (no y: the request still goes out -- to x1's lines -- and its value is not used)
    uint8_t* out = reinterpret_cast<uint8_t*>(a.out);
    uint8_t* out2 = reinterpret_cast<uint8_t*>(a.out2);
    uint8_t* hs = reinterpret_cast<uint8_t*>(a.h);
    const float inv_d = 1.0f / (float)d;
    // the next row of this wave is requested before the current one is reduced: a wave that loads, reduces, stores
    // and only then loads again has nothing in flight half of the time (2.8-3.3 TB/s before)
    const int64_t rstride = (int64_t)gridDim.x * TAIL_WAVES;
    Raw cy[NP], cx[NP];
    auto load_row = [&](int64_t r, Raw (&ry)[NP], Raw (&rx)[NP]) {
        if (r >= a.M) r = a.M - 1;                  // unconditional (clamped) loads keep the vmcnt waits counted
        const int64_t o = r * d * (int64_t)sizeof(IO);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = lane + 64 * k;
            if (p < pieces) { ry[k] = P::load_raw_nt(ysrc + o + p * PB); rx[k] = P::load_raw_nt(x1 + o + p * PB); }
        }
    };
    {
        const int64_t r0 = (int64_t)blockIdx.x * TAIL_WAVES + wave;
        if (r0 < a.M) load_row(r0, cy, cx);
    }
    for (int64_t row = (int64_t)blockIdx.x * TAIL_WAVES + wave; row < a.M; row += rstride) {
        const int64_t rb = row * d * (int64_t)sizeof(IO);
        Raw ny[NP], nx[NP];
        load_row(row + rstride, ny, nx);
        float h[NP][E];
        float xpost[POST ? NP : 1][E];
        float s = 0.f;
        uint32_t kb[NP];
        if (thr) row_keep_bits<NP, E>(row * d, lane, pieces, seed, thr, kb);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = lane + 64 * k;
            if (p < pieces) {
                float vy[E], vx[E];
                if (y) P::from_raw(cy[k], vy);
                else {
#pragma unroll
                    for (int j = 0; j < E; ++j) vy[j] = 0.f;
                }
                P::from_raw(cx[k], vx);
                if constexpr (POST) {
#pragma unroll
                    for (int j = 0; j < E; ++j) { xpost[k][j] = vx[j]; vx[j] = 0.f; }
                }
                const uint32_t bits = thr ? kb[k] : 0xffu;
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    h[k][j] = vx[j] + (((bits >> j) & 1u) ? vy[j] * scale : 0.f);
                    s += h[k][j];
                }
                if (a.keep_out) {
#pragma unroll
                    for (int j = 0; j < E; ++j) a.keep_out[row * d + p * E + j] = (bits >> j) & 1u;
                }
            } else {
#pragma unroll
                for (int j = 0; j < E; ++j) h[k][j] = 0.f;
            }
        }
        if constexpr (NORM) {
            const float mean = a.rms ? 0.f : wave_sum(s) * inv_d;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if (lane + 64 * k < pieces) {
#pragma unroll
                    for (int j = 0; j < E; ++j) { const float c = h[k][j] - mean; q += c * c; }
                }
            }
            const float rstd = rsqrtf(wave_sum(q) * inv_d + a.eps);
            if (lane == 0) { if (a.mean) a.mean[row] = mean; a.rstd[row] = rstd; }
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = lane + 64 * k;
                if (p < pieces) {
                    if (hs) P::store(hs + rb + p * PB, h[k]);
                    float o[E];
#pragma unroll
                    for (int j = 0; j < E; ++j) o[j] = (h[k][j] - mean) * rstd * gam[k][j] + bet[k][j];
                    if constexpr (POST) {
#pragma unroll
                        for (int j = 0; j < E; ++j) o[j] += xpost[k][j];
                    }
                    P::store(out + rb + p * PB, o);
                }
            }
        } else {
            float rstd2 = 0.f;
            if constexpr (RMS2) {
                float q = 0.f;
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    if (lane + 64 * k < pieces) {
#pragma unroll
                        for (int j = 0; j < E; ++j) {
                            // the norm sees the sum as the next launch would have read it: rounded to the IO dtype
                            const float hr = sizeof(IO) == 2 ? (float)(__bf16)h[k][j] : h[k][j];
                            q += hr * hr;
                        }
                    }
                }
                rstd2 = rsqrtf(wave_sum(q) * inv_d + a.eps);
                if (lane == 0) a.rstd[row] = rstd2;
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = lane + 64 * k;
                if (p < pieces) {
                    P::store(out + rb + p * PB, h[k]);
                    if constexpr (RMS2) {
                        float o2[E];
#pragma unroll
                        for (int j = 0; j < E; ++j) {
                            const float hr = sizeof(IO) == 2 ? (float)(__bf16)h[k][j] : h[k][j];
                            o2[j] = hr * rstd2 * gam2[k][j];
                        }
                        P::store(out2 + rb + p * PB, o2);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) { cy[k] = ny[k]; cx[k] = nx[k]; }
    }
}

// backward: dh = LN'(dout) (or dout when NORM == 0); dx1 = dh; dy = dh * keep / (1-p) (only written when
// dropout is on -- with p = 0 the caller aliases dy to dx1); per-workgroup partial sums of dgamma / dbeta.
// HOUT (NORM only, round 4): `h` holds the LayerNorm OUTPUT rows (what the next sublayer keeps anyway) instead of the pre-norm sum,
// and the normalised rows are recovered as xhat = (out - beta) / gamma (0 where gamma is 0): the forward then writes ONE row tensor
// (out) instead of two (out and h) -- 3 units of traffic instead of 4.
// Round 5: the per-column constants (gamma; HOUT: 1 / gamma and beta / gamma too) live in LDS, one piece-sized vector per lane and
// piece, and are fetched where a sweep uses them; what stays in registers for the whole launch are the dgamma / dbeta sums only.
// With 8-byte pieces at d = 768 (three per lane, every lane busy) the HOUT form went from 196 registers / 2 waves per SIMD to <= 128 / 4.
// The row is swept twice from its raw registers (sums first, then the gradient: g and xhat are recomputed per piece instead of being
// held across the wave reduction -- 24 registers at d = 768), DRES (the parked gradient of T5's norm link) is a template flag.
template <typename IO, int NP, bool NORM, bool HOUT = false, int PB = 16, bool DRES = false, bool FULL = false>
__global__ __launch_bounds__(TAIL_WAVES * 64) void tail_bwd_kernel(TailArgs a) {
    static_assert(!HOUT || NORM, "recovering xhat from the output needs the norm");
    using P = Piece<IO, PB>;
    using Raw = typename P::Raw;
    constexpr int E = P::E;
    constexpr int NPRM = NORM ? (HOUT ? 3 : 1) : 0;                 // gamma [, 1 / gamma, beta / gamma]
    constexpr int PRM_F = NPRM * NP * 64 * E, ACC_F = TAIL_WAVES * 2 * 64 * E;
    __shared__ __attribute__((aligned(16))) float sm[(PRM_F > ACC_F ? PRM_F : ACC_F) > 0 ? (PRM_F > ACC_F ? PRM_F : ACC_F) : 4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int d = FULL ? NP * 64 * E : a.d, pieces = FULL ? NP * 64 : d / E;      // (FULL: see the forward)
    const uint32_t thr = a.thr;
    const uint64_t seed = thr ? vlpet_eff_seed(a.seed, a.seed_ctr) : 0;      // (one scalar load, before the row loop)
    const float scale = a.keep_scale;
    float dg[NORM ? NP : 1][E], db[NORM ? NP : 1][E];
    if constexpr (NORM) {
        // slot (arr, k, lane) of the parameter area holds the E values of piece lane + 64 k
        for (int q = threadIdx.x; q < NP * 64; q += TAIL_WAVES * 64) {
            const int p = q;                                        // piece index = lane' + 64 k'  (q = 64 k' + lane')
            float gv[E], bv[E];
            if constexpr (FULL) {       // (vector loads, no branch: see the forward)
                const float bm = a.beta ? 1.f : 0.f;
                const float* bsrc = a.beta ? a.beta : a.gamma;
                load_f32_vec<E>(a.gamma + p * E, gv);
                if constexpr (HOUT) {
                    load_f32_vec<E>(bsrc + p * E, bv);
#pragma unroll
                    for (int j = 0; j < E; ++j) bv[j] *= bm;
                }
            } else {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    gv[j] = p < pieces ? a.gamma[p * E + j] : 0.f;
                    if constexpr (HOUT) bv[j] = (p < pieces && a.beta) ? a.beta[p * E + j] : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const float gm = gv[j];
                sm[q * E + j] = gm;
                if constexpr (HOUT) {
                    const float gi = gm != 0.f ? 1.0f / gm : 0.f;
                    sm[(NP * 64 + q) * E + j] = gi;
                    sm[(2 * NP * 64 + q) * E + j] = bv[j] * gi;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
            for (int j = 0; j < E; ++j) { dg[k][j] = 0.f; db[k][j] = 0.f; }
        __syncthreads();
    }
    auto prm = [&](int arr, int k, float* v) {                      // this lane's piece k of parameter vector `arr`
        const float* src = sm + ((arr * NP + k) * 64 + lane) * E;
#pragma unroll
        for (int j = 0; j < E; j += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(src + j);
            v[j] = t[0]; v[j + 1] = t[1]; v[j + 2] = t[2]; v[j + 3] = t[3];
        }
    };
    const uint8_t* dout = reinterpret_cast<const uint8_t*>(a.out);     // `out` carries dout in the backward
    const uint8_t* hs = reinterpret_cast<const uint8_t*>(a.h);
    uint8_t* dx1 = reinterpret_cast<uint8_t*>(const_cast<void*>(a.x1));
    uint8_t* dy = reinterpret_cast<uint8_t*>(const_cast<void*>(a.y));
    const float inv_d = 1.0f / (float)d;
    const int64_t rstride = (int64_t)gridDim.x * TAIL_WAVES;
    const uint8_t* dres = reinterpret_cast<const uint8_t*>(a.dres);
    Raw cd[NP], ch[NORM ? NP : 1], cr[DRES ? NP : 1];
    float cmean = 0.f, crstd = 1.f;
    auto load_row = [&](int64_t r, Raw (&rd)[NP], Raw (&rh)[NORM ? NP : 1], Raw (&rr)[DRES ? NP : 1], float& mu, float& rs) {
        if (r >= a.M) r = a.M - 1;                  // next row of this wave, requested one row ahead (see the forward)
        const int64_t o = r * d * (int64_t)sizeof(IO);
        if constexpr (NORM) { mu = (HOUT || a.h_xhat || a.rms) ? 0.f : a.mean[r]; rs = a.rstd[r]; }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = lane + 64 * k;
            if (p < pieces) {
                rd[k] = P::load_raw_nt(dout + o + p * PB);
                if constexpr (NORM) rh[k] = P::load_raw_nt(hs + o + p * PB);
                if constexpr (DRES) rr[k] = P::load_raw_nt(dres + o + p * PB);
            }
        }
    };
    {
        const int64_t r0 = (int64_t)blockIdx.x * TAIL_WAVES + wave;
        if (r0 < a.M) load_row(r0, cd, ch, cr, cmean, crstd);
    }
    for (int64_t row = (int64_t)blockIdx.x * TAIL_WAVES + wave; row < a.M; row += rstride) {
        const int64_t rb = row * d * (int64_t)sizeof(IO);
        Raw nd[NP], nh[NORM ? NP : 1], nr[DRES ? NP : 1];
        float nmean = 0.f, nrstd = 1.f;
        load_row(row + rstride, nd, nh, nr, nmean, nrstd);
        float s1 = 0.f, s2 = 0.f;
        const float mean = cmean, rstd = crstd;
        const float rin = a.h_xhat ? 1.f : rstd;            // (rows already normalised: K4's saved xhat)
        // g = dout * gamma and xhat of piece k, from the raw registers
        auto piece_gx = [&](int k, float* gq, float* xq, float* vd) {
            float vh[E], gm[E];
            P::from_raw(cd[k], vd);
            P::from_raw(ch[k], vh);
            prm(0, k, gm);
            if constexpr (HOUT) {
                float gi[E], bg[E];
                prm(1, k, gi); prm(2, k, bg);
#pragma unroll
                for (int j = 0; j < E; ++j) xq[j] = vh[j] * gi[j] - bg[j];
            } else {
#pragma unroll
                for (int j = 0; j < E; ++j) xq[j] = (vh[j] - mean) * rin;
            }
#pragma unroll
            for (int j = 0; j < E; ++j) gq[j] = vd[j] * gm[j];
        };
        float c1 = 0.f, c2 = 0.f;
        if constexpr (NORM) {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if (lane + 64 * k < pieces) {
                    float gq[E], xq[E], vd[E];
                    piece_gx(k, gq, xq, vd);
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        s1 += gq[j];
                        s2 += gq[j] * xq[j];
                        dg[k][j] += vd[j] * xq[j];
                        db[k][j] += vd[j];
                    }
                }
            }
            wave_sum2(s1, s2, c1, c2);
            c1 = a.rms ? 0.f : c1 * inv_d; c2 *= inv_d;
        }
        uint32_t kb[NP];
        if (thr) row_keep_bits<NP, E>(row * d, lane, pieces, seed, thr, kb);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = lane + 64 * k;
            if (p < pieces) {
                float g[E];
                if constexpr (NORM) {
                    float xq[E], vd[E];
                    piece_gx(k, g, xq, vd);
#pragma unroll
                    for (int j = 0; j < E; ++j) g[j] = (g[j] - c1 - xq[j] * c2) * rstd;
                } else {
                    P::from_raw(cd[k], g);
                }
                if constexpr (DRES) {                    // the parked gradient of the other reader of this input
                    float vr[E];
                    P::from_raw(cr[k], vr);
#pragma unroll
                    for (int j = 0; j < E; ++j) g[j] += vr[j];
                }
                if (NORM || DRES || dx1) P::store(dx1 + rb + p * PB, g);      // (plain residual tail: dx1 == dout, the caller may alias it and pass no dx1)
                if (thr) {
                    const uint32_t bits = kb[k];
                    float o[E];
#pragma unroll
                    for (int j = 0; j < E; ++j) o[j] = ((bits >> j) & 1u) ? g[j] * scale : 0.f;
                    P::store(dy + rb + p * PB, o);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) { cd[k] = nd[k]; if constexpr (NORM) ch[k] = nh[k]; if constexpr (DRES) cr[k] = nr[k]; }
        cmean = nmean; crstd = nrstd;
    }
    if constexpr (NORM) {
        if (a.dgb) {
            // partial sums of this workgroup: [blockIdx][2][d]; waves combined through LDS one piece at a time (the parameter
            // area is dead by now: every wave is past its last row)
            __syncthreads();
            float* acc = sm;                                        // [TAIL_WAVES][2][64 * E]
            float* dst = a.dgb + (size_t)blockIdx.x * 2 * d;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = lane + 64 * k;
#pragma unroll
                for (int j = 0; j < E; ++j) { acc[(wave * 2 + 0) * 64 * E + lane * E + j] = dg[k][j]; acc[(wave * 2 + 1) * 64 * E + lane * E + j] = db[k][j]; }
                __syncthreads();
                if (wave == 0 && p < pieces) {
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        float sg = 0.f, sb = 0.f;
#pragma unroll
                        for (int w = 0; w < TAIL_WAVES; ++w) { sg += acc[(w * 2 + 0) * 64 * E + lane * E + j]; sb += acc[(w * 2 + 1) * 64 * E + lane * E + j]; }
                        dst[p * E + j] = sg; dst[d + p * E + j] = sb;
                    }
                }
                __syncthreads();
            }
        }
    }
}

// Sum of the backward's per-workgroup partials [nb][2][d] into dgamma [d] / dbeta [d] (OVERWRITTEN; either may be null):
// a workgroup owns 64 consecutive columns of the [2 d] row, its 16 waves take every 16th partial row (a wave reads 256
// contiguous bytes per row), LDS combines the waves.  One launch instead of a library reduction plus autograd's adds.
__global__ __launch_bounds__(1024) void tail_reduce_kernel(const float* __restrict__ part, int nb, int d,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float acc[16][64];
    const int c = threadIdx.x & 63, s = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const int w = 2 * d;
    float v = 0.f;
    if (col < w) {
        const float* src = part + col;
        int r = s;
        for (; r + 48 < nb; r += 64) {
            const float a0 = src[(size_t)r * w], a1 = src[(size_t)(r + 16) * w], a2 = src[(size_t)(r + 32) * w],
                        a3 = src[(size_t)(r + 48) * w];
            v += (a0 + a1) + (a2 + a3);
        }
        for (; r < nb; r += 16) v += src[(size_t)r * w];
    }
    acc[s][c] = v;
    __syncthreads();
    if (s == 0 && col < w) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += acc[k][c];
        if (col < d) { if (dgamma) dgamma[col] = t; }
        else if (dbeta) dbeta[col - d] = t;
    }
}

// Column sums of x [M, n] (the gradient of a trainable Linear bias: autograd's dy.sum(0), the LoRA runs train every bias,
// my_transformers/modeling_bart.py:791-811 with lora/controller.py): pass 1 = per-workgroup partial sums [blocks][n] (a wave owns
// a row, lanes own 16-byte pieces, the next row requested before the current one is added), pass 2 = tail_reduce_kernel over
// the partials viewed as [blocks][2 * (n / 2)].
template <typename IO, int NP>
__global__ __launch_bounds__(TAIL_WAVES * 64) void colsum_partial_kernel(const void* __restrict__ xin, int64_t M, int n,
                                                                         float* __restrict__ part) {
    using P = Piece<IO>;
    constexpr int E = P::E;
    __shared__ float acc[TAIL_WAVES][64 * 8];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pieces = n / E;
    const uint8_t* x = reinterpret_cast<const uint8_t*>(xin);
    const int64_t rstride = (int64_t)gridDim.x * TAIL_WAVES;
    float s[NP][E];
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int j = 0; j < E; ++j) s[k][j] = 0.f;
    u32x4 cur[NP];
    auto load_row = [&](int64_t r, u32x4 (&rd)[NP]) {
        if (r >= M) r = M - 1;
        const int64_t o = r * n * (int64_t)sizeof(IO);
#pragma unroll
        for (int k = 0; k < NP; ++k) { const int p = lane + 64 * k; if (p < pieces) rd[k] = P::load_raw_nt(x + o + p * 16); }
    };
    const int64_t r0 = (int64_t)blockIdx.x * TAIL_WAVES + wave;
    if (r0 < M) load_row(r0, cur);
    for (int64_t row = r0; row < M; row += rstride) {
        u32x4 nxt[NP];
        load_row(row + rstride, nxt);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (lane + 64 * k < pieces) {
                float v[E];
                P::from_raw(cur[k], v);
#pragma unroll
                for (int j = 0; j < E; ++j) s[k][j] += v[j];
            }
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) cur[k] = nxt[k];
    }
    float* dst = part + (size_t)blockIdx.x * n;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int p = lane + 64 * k;
#pragma unroll
        for (int j = 0; j < E; ++j) acc[wave][lane * 8 + j] = s[k][j];
        __syncthreads();
        if (wave == 0 && p < pieces) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < TAIL_WAVES; ++w) t += acc[w][lane * 8 + j];
                dst[p * E + j] = t;
            }
        }
        __syncthreads();
    }
}

template <typename IO>
static hipError_t launch_colsum_io(const void* x, int64_t M, int n, float* part, int blocks, hipStream_t stream) {
    const int np = (n / Piece<IO>::E + 63) / 64;
    const dim3 g(blocks), b(TAIL_WAVES * 64);
    if (np <= 1) hipLaunchKernelGGL((colsum_partial_kernel<IO, 1>), g, b, 0, stream, x, M, n, part);
    else if (np <= 2) hipLaunchKernelGGL((colsum_partial_kernel<IO, 2>), g, b, 0, stream, x, M, n, part);
    else if (np <= 4) hipLaunchKernelGGL((colsum_partial_kernel<IO, 4>), g, b, 0, stream, x, M, n, part);
    else if (np <= 8) hipLaunchKernelGGL((colsum_partial_kernel<IO, 8>), g, b, 0, stream, x, M, n, part);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// out [n] (fp32, OVERWRITTEN) = column sums of x [M, n]; part = workspace of tail_blocks(M) * n floats
hipError_t launch_colsum(const void* x, int64_t M, int n, float* part, float* out, int io_fp32, hipStream_t stream) {
    const int blocks = tail_blocks(M);
    hipError_t e = io_fp32 ? launch_colsum_io<float>(x, M, n, part, blocks, stream) : launch_colsum_io<__bf16>(x, M, n, part, blocks, stream);
    if (e != hipSuccess) return e;
    return launch_tail_reduce(part, blocks, n / 2, out, out + n / 2, stream);
}

// Several such reductions in ONE launch (blockIdx.y = job): the LoRA runs train every bias, 96 column-sum reductions + 30 LayerNorm ones
// per step, each a 5.6 us launch of a few dozen workgroups (0.7 ms of a 23 ms step).  A trainer queues them during the backward and
// flushes once (functional.flush_reduces): nobody reads a parameter gradient before the optimizer.
__global__ __launch_bounds__(1024) void tail_reduce_batch_kernel(ReduceBatch b) {
    __shared__ float acc[16][64];
    const ReduceJob J = b.j[blockIdx.y];
    const int c = threadIdx.x & 63, s = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const int w = 2 * J.d, nb = J.nb;
    if ((int)blockIdx.x * 64 >= w) return;            // (jobs narrower than the widest one of the batch)
    float v = 0.f;
    if (col < w) {
        const float* src = J.part + col;
        int r = s;
        for (; r + 48 < nb; r += 64) {
            const float a0 = src[(size_t)r * w], a1 = src[(size_t)(r + 16) * w], a2 = src[(size_t)(r + 32) * w],
                        a3 = src[(size_t)(r + 48) * w];
            v += (a0 + a1) + (a2 + a3);
        }
        for (; r < nb; r += 16) v += src[(size_t)r * w];
    }
    acc[s][c] = v;
    __syncthreads();
    if (s == 0 && col < w) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += acc[k][c];
        if (col < J.d) { if (J.out0) J.out0[col] = t; }
        else if (J.out1) J.out1[col - J.d] = t;
    }
}

hipError_t launch_tail_reduce_batch(const ReduceBatch& b, int max_d, hipStream_t stream) {
    hipLaunchKernelGGL(tail_reduce_batch_kernel, dim3((2 * max_d + 63) / 64, b.n), dim3(1024), 0, stream, b);
    return hipGetLastError();
}

// pass 1 of launch_colsum alone (the reduction is queued by the caller)
hipError_t launch_colsum_partial(const void* x, int64_t M, int n, float* part, int io_fp32, hipStream_t stream) {
    const int blocks = tail_blocks(M);
    return io_fp32 ? launch_colsum_io<float>(x, M, n, part, blocks, stream) : launch_colsum_io<__bf16>(x, M, n, part, blocks, stream);
}

hipError_t launch_tail_reduce(const float* part, int nb, int d, float* dgamma, float* dbeta, hipStream_t stream) {
    hipLaunchKernelGGL(tail_reduce_kernel, dim3((2 * d + 63) / 64), dim3(1024), 0, stream, part, nb, d, dgamma, dbeta);
    return hipGetLastError();
}

// Workgroups of the row kernels: a wave walks rows with the next one prefetched, and its prologue (gamma / beta, 2 us) and the
// partial sums it leaves are per workgroup, so few rows per wave is NOT what makes a short launch short.  Round-5 sweep of the cap with
// the 118-register kernels (four workgroups per CU; profiles/r05_k5abi_small_m_cap.txt, debug build, VLPET_DBG = cap): a short launch is
// a ~5 us floor plus ~1.3 us per 1,000 rows whatever the cap between 256 and 1,024; one workgroup per CU is best up to ~2,500 rows
// (7.2 vs 7.8 us at 2,000), two from there to ~8,500 (9.9 vs 10.4 / 11.3 us at 3,500 rows with 256 / 875 workgroups; 10.8 vs 12.7 / 12.0
// at 5,000), four above (profiles/r05_k5abi_ab.txt: 768 -> 1,024 is 2-4 % at 28,000+ rows).
int tail_blocks(int64_t M) {
    const int64_t need = (M + TAIL_WAVES - 1) / TAIL_WAVES;
    int64_t cap = M <= 2500 ? 256 : M <= 8500 ? 512 : 1024;
    if (VLPET_IS_DEBUG_BUILD && vlpet_tuning().dbg >= 64) cap = vlpet_tuning().dbg;      // (diagnosis: VLPET_DBG = cap)
    return (int)(need < cap ? need : cap);
}

template <typename IO, int NP, bool NORM, int PB, bool FULL>
static hipError_t launch_np_full(const TailArgs& a, bool bwd, hipStream_t stream) {
    const int blocks = tail_blocks(a.M);
    const dim3 g(blocks), t(TAIL_WAVES * 64);
    if (bwd && NORM && a.h_out) {
        if constexpr (NORM) hipLaunchKernelGGL((tail_bwd_kernel<IO, NP, true, true, PB, false, FULL>), g, t, 0, stream, a);
    }
    else if (bwd && a.dres) hipLaunchKernelGGL((tail_bwd_kernel<IO, NP, NORM, false, PB, true, FULL>), g, t, 0, stream, a);
    else if (bwd) hipLaunchKernelGGL((tail_bwd_kernel<IO, NP, NORM, false, PB, false, FULL>), g, t, 0, stream, a);
    else if (NORM && a.post) {
        if constexpr (NORM) hipLaunchKernelGGL((tail_fwd_kernel<IO, NP, true, true, PB, false, FULL>), g, t, 0, stream, a);
    }
    else if (!NORM && a.out2) {
        if constexpr (!NORM) hipLaunchKernelGGL((tail_fwd_kernel<IO, NP, false, false, PB, true, FULL>), g, t, 0, stream, a);
    }
    else hipLaunchKernelGGL((tail_fwd_kernel<IO, NP, NORM, false, PB, false, FULL>), g, t, 0, stream, a);
    return hipGetLastError();
}

template <typename IO, int NP, bool NORM, int PB>
static hipError_t launch_np(const TailArgs& a, bool bwd, hipStream_t stream) {
    // rows of exactly NP x 64 pieces (768 / 8-byte pieces, 1024, 2048 ...): the branch-free instantiation (bf16 only: the fp32 rows of the
    // tests are not worth a second copy of every kernel); the per-column vectors must be 16-byte aligned for its vector loads (a view
    // into a flat parameter buffer may not be)
    const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (sizeof(IO) == 2 && NP <= 4 && a.d == NP * 64 * (PB / (int)sizeof(IO)) && al16(a.gamma) && al16(a.beta) && al16(a.gamma2)) {
        if constexpr (sizeof(IO) == 2 && NP <= 4) return launch_np_full<IO, NP, NORM, PB, true>(a, bwd, stream);      // (eight pieces per lane: two rows in flight do not fit the registers)
    }
    return launch_np_full<IO, NP, NORM, PB, false>(a, bwd, stream);
}

// Piece size of a bf16 row: 8-byte pieces when they leave fewer per-lane element slots than 16-byte ones (d = 768: 3 x 4 = 12
// against 2 x 8 = 16 with half the wave idle on the second piece); ties go to the wider loads.
static bool tail_pieces8(int d) {
    if (d % 4 != 0) return false;
    const int np16 = (d / 8 + 63) / 64, np8 = (d / 4 + 63) / 64;
    return d % 8 == 0 && np8 <= 4 && np8 * 4 < np16 * 8;
}

template <typename IO, bool NORM>
static hipError_t launch_io(const TailArgs& a, bool bwd, hipStream_t stream) {
    if constexpr (sizeof(IO) == 2) {
        if (tail_pieces8(a.d)) {
            const int np = (a.d / 4 + 63) / 64;
            if (np <= 1) return launch_np<IO, 1, NORM, 8>(a, bwd, stream);
            if (np <= 2) return launch_np<IO, 2, NORM, 8>(a, bwd, stream);
            if (np <= 3) return launch_np<IO, 3, NORM, 8>(a, bwd, stream);
            return launch_np<IO, 4, NORM, 8>(a, bwd, stream);
        }
    }
    const int pieces = a.d / Piece<IO>::E;
    const int np = (pieces + 63) / 64;
    if (np <= 1) return launch_np<IO, 1, NORM, 16>(a, bwd, stream);
    if (np <= 2) return launch_np<IO, 2, NORM, 16>(a, bwd, stream);
    if (np <= 3) return launch_np<IO, 3, NORM, 16>(a, bwd, stream);
    if (np <= 4) return launch_np<IO, 4, NORM, 16>(a, bwd, stream);
    if (np <= 8) return launch_np<IO, 8, NORM, 16>(a, bwd, stream);
    return hipErrorInvalidValue;
}

hipError_t launch_tail(const TailArgs& a, int io_fp32, bool bwd, hipStream_t stream) {
    if (a.post && (!a.norm || bwd)) return hipErrorInvalidValue;      // forward-only form of the norm path
    if (a.out2 && (a.norm || bwd || !a.gamma2 || !a.rstd)) return hipErrorInvalidValue;
    if (a.h_out && (!a.norm || !bwd || a.rms || a.h_xhat || a.dres)) return hipErrorInvalidValue;
    if (a.norm) return io_fp32 ? launch_io<float, true>(a, bwd, stream) : launch_io<__bf16, true>(a, bwd, stream);
    return io_fp32 ? launch_io<float, false>(a, bwd, stream) : launch_io<__bf16, false>(a, bwd, stream);
}
