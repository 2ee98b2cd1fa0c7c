// Short-sequence multi-head attention of the frozen backbone (head dim 64, at most 128 keys and 128 queries per
// sequence), forward and backward, for the shapes VL-PET trains on: S = 20-40 text + 36 / 72 visual tokens per sample and
// hundreds of samples per step (encoder self-attention B x 12 heads x 56..92 tokens; decoder self / cross attention with
// 2-20 queries).  my_transformers/modeling_bart.py:283-566 (BartAttention.forward: q k^T / sqrt(d), + mask, softmax,
// dropout(p = attention_dropout), . v) computes this with bmm + softmax + dropout + bmm; the generic flash kernels the
// library path dispatches to are built for long sequences (one 56-token sequence fills less than half of one of their
// tiles) and took 22 % of the train step at configs[1].
//
// Here a workgroup owns one (batch, head) pair and everything stays on chip: K^T Q and V^T P as 32x32x16 MFMAs in the
// "swapped" form (a lane owns a QUERY: its 16 accumulator registers of a tile are 16 keys, so the softmax over keys is a
// reduction inside the lane plus one exchange with lane ^ 32), probabilities go straight from the accumulator registers
// into the next MFMA's B operand, and the V^T / K^T / Q^T / dO^T operands come from row-major LDS images through
// ds_read_b64_tr_b16.  HBM traffic = the algorithmic minimum: forward reads q, k, v once and writes o (+ 4 B per row of
// log-sum-exp); backward reads q, k, v, o, do and writes dq, dk, dv.
//
// Backward, two phases per workgroup with everything recomputed from q, k, v and the saved log-sum-exp:
//   phase K (a wave owns a 32-key tile, loops over the query blocks):  S = Q K^T and dP = dO V^T in the UNSWAPPED form (lane
//     = key, registers = queries), P and dS elementwise, then dV^T += dO^T P, dK^T += Q^T dS (contraction over queries:
//     B operand = the registers, A operand = transpose-read of the dO / Q image);
//   phase Q (a wave owns a 32-query block, loops over the key tiles):  the swapped form again, dQ^T += K^T dS.
// Dropout keeps element (b, h, i, j) iff hash32(row_key(b, h, i) + j * golden) >= p * 2^32: a function of the element index
// and the call's seed only, so the three places that need the mask regenerate it.
#include <cstdlib>
#include "common.h"
#include "kernels.h"

#define AT_LD 72                       // bf16 elements per LDS row: 64 + 8 of padding (144 B: 16-byte aligned, conflict-light)
#define AT_ROW (AT_LD * 2)             // bytes
#define AT_NW 4                        // waves per workgroup
#define LOG2E 1.4426950408889634f

typedef short v4s16_t __attribute__((ext_vector_type(4)));
typedef short v8s16_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32: arguments here are <= 0 or -inf

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// per-row key of the dropout mask: rows are (b, h, i) triples
__device__ __forceinline__ uint32_t row_key(uint64_t seed, int64_t row) {
    return hash32((uint32_t)seed ^ (uint32_t)row) ^ hash32((uint32_t)(seed >> 32) + (uint32_t)((uint64_t)row >> 32));
}
// element (row, j): one multiply-xorshift round over the row's key (a full hash32 of the seed and the row index) plus a Weyl step per
// column -- four VALU operations per element instead of nine (the backward kernels are bound by their instruction count)
__device__ __forceinline__ uint32_t hash_elem(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; return x; }
__device__ __forceinline__ bool keep_elem(uint32_t rk, int j, uint32_t thr) { return hash_elem(rk + (uint32_t)j * 0x9E3779B9U) >= thr; }

// A operand from a row-major LDS image: lane (c = lane & 31, hh = lane >> 5) gets column cb + c of rows
// kb + 4 hh + {0..3} (slots 0..3) and kb + 8 + 4 hh + {0..3} (slots 4..7) -- the row order in which a 32x32 accumulator
// tile hands its registers 8u .. 8u+7 to the next MFMA (accumulator register r of lane half hh is row (r & 3) + 8 (r >> 2) + 4 hh).
__device__ __forceinline__ bf16x8 tr_acc_order(const uint8_t* img, int kb, int cb, int lane) {
    const int g = lane >> 4, sl = lane & 15;
    const uint8_t* p = img + (size_t)(kb + 4 * (g >> 1) + (sl >> 2)) * AT_ROW + (cb + 16 * (g & 1) + 4 * (sl & 3)) * 2;
    typedef __attribute__((address_space(3))) v4s16_t lds_v4;
    const v4s16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p));
    const v4s16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p + 8 * AT_ROW));
    const v8s16_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}
// natural operand: row `row` of the image, 8 consecutive columns at 16 ks + 8 hh
__device__ __forceinline__ bf16x8 nat_frag(const uint8_t* img, int row, int ks, int hh) {
    return *reinterpret_cast<const bf16x8*>(img + (size_t)row * AT_ROW + (16 * ks + 8 * hh) * 2);
}
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& t, int u) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (__bf16)t[8 * u + j];
    return f;
}
// [rows_pad][64] bf16 rows of one (b, h) -> LDS image, rows >= n_rows zero
__device__ __forceinline__ void stage_image(uint8_t* img, const __bf16* src, int64_t rs, int n_rows, int rows_pad, int tid, int nthreads) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int idx = tid; idx < rows_pad * 8; idx += nthreads) {
        const int row = idx >> 3, pc = idx & 7;
        u32x4 v = z;
        if (row < n_rows) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + (int64_t)row * rs + pc * 8));      // (read once per launch: nt)
        *reinterpret_cast<u32x4*>(img + (size_t)row * AT_ROW + pc * 16) = v;
    }
}
// the same for several images at once with every global load in flight before the first LDS store (a plain per-piece
// loop compiles to load -> wait -> store per trip: one memory latency per 4 KiB of image)
template <int NIMG, int MAXP>
__device__ __forceinline__ void stage_images(uint8_t* const* img, const __bf16* const* src, const int* n_rows, const int* rows_pad,
                                             const int64_t* rs, int tid, int nthreads) {
    u32x4 v[NIMG][MAXP];
#pragma unroll
    for (int g = 0; g < NIMG; ++g)
#pragma unroll
        for (int c = 0; c < MAXP; ++c) {
            const int idx = tid + c * nthreads, row = idx >> 3, pc = idx & 7;
            const int rr = row < n_rows[g] ? row : n_rows[g] - 1;
            v[g][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src[g] + (int64_t)rr * rs[g] + pc * 8));
        }
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int g = 0; g < NIMG; ++g)
#pragma unroll
        for (int c = 0; c < MAXP; ++c) {
            const int idx = tid + c * nthreads, row = idx >> 3, pc = idx & 7;
            if (row < rows_pad[g]) *reinterpret_cast<u32x4*>(img[g] + (size_t)row * AT_ROW + pc * 16) = row < n_rows[g] ? v[g][c] : z;
        }
}
// accumulator pair D[d][row] (two 32-wide d tiles; lane = row, registers = d) -> global rows through the wave's 16 x 128 B staging
// tile (144-byte rows), sixteen rows at a time with both d tiles, so that every store instruction writes eight WHOLE 128-byte rows.
// (The first version staged one d tile at a time and stored 64 bytes per row: the backward spent 56 of its 118 us at B = 500, S = 56
// writing 129 MB as half lines -- profiles/r04_attnbwd_ablation.txt.  Sixteen rows, not 32: the staging tiles decide how many
// workgroups fit a CU's LDS.)
#define AT_SROW 144
#define AT_STG (16 * AT_SROW)
__device__ __forceinline__ void store_rows_T(uint8_t* stg, const f32x16& t0, const f32x16& t1, __bf16* dst, int64_t rs,
                                             int row0, int n_rows, int lane) {
    const int m = lane & 31, hh = lane >> 5;
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if ((m >> 4) == half) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const f32x16& t = dt ? t1 : t0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bf16x4_t w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = (__bf16)t[4 * q + e];
                    *reinterpret_cast<bf16x4_t*>(stg + (size_t)(m & 15) * AT_SROW + (32 * dt + 8 * q + 4 * hh) * 2) = w;
                }
            }
        }
        // (same-wave LDS accesses are ordered in hardware: no barrier.  The compiler is told: the tile is written as bf16x4 and read
        // as u32x4, which type-based alias analysis would otherwise let it reorder across the two halves.)
        asm volatile("" ::: "memory");
        u32x4 v[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int idx = lane + 64 * c, row = idx >> 3, pc = idx & 7;
            v[c] = *reinterpret_cast<const u32x4*>(stg + (size_t)row * AT_SROW + pc * 16);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int idx = lane + 64 * c, row = 16 * half + (idx >> 3), pc = idx & 7;
            if (row0 + row < n_rows) *reinterpret_cast<u32x4*>(dst + (int64_t)(row0 + row) * rs + pc * 8) = v[c];
        }
    }
}

// accumulator tile initialised with the lane's 16 bias values (four runs of four floats from a padded fp32 row) over the score scale:
// the MFMAs then accumulate q k^T onto it and no extra registers live through the products
__device__ __forceinline__ f32x16 bias_tile(const float* brow, float inv_scale) {
    f32x16 t;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(brow + 8 * q4);
#pragma unroll
        for (int e = 0; e < 4; ++e) t[4 * q4 + e] = bv[e] * inv_scale;
    }
    return t;
}

#define AT_FW 2                        // forward: waves per workgroup (each wave owns whole (batch, head) pairs: no barrier)

struct AttnLds {
    // forward, per wave: V image | staging tile;  backward, per workgroup: Q, dO, K, V images | staging x AT_NW | per-row {lse2, delta, row key}
    __host__ __device__ static constexpr size_t fwd_wave_bytes(int Lkp) { return (size_t)Lkp * AT_ROW + (size_t)AT_STG + (size_t)Lkp * 4; }
    static size_t fwd_bytes(int Lkp) { return (size_t)AT_FW * fwd_wave_bytes(Lkp); }
    static size_t bwd_bytes(int Lqp, int Lkp) {
        return bwd_bytes_nw(Lqp, Lkp, AT_NW);
    }
    static size_t bwd_bytes_nw(int Lqp, int Lkp, int nw) {
        return (size_t)2 * (Lqp + Lkp) * AT_ROW + (size_t)nw * AT_STG + (size_t)Lqp * 12 + (size_t)Lkp * 4;
    }
};

// key-validity table of a pair: 0 / -inf per key (bounds and the boolean key mask), added to the scaled scores; the kernels hold no
// per-element test of the mask (as branches around a global load they cost more than the rest of the elementwise work)
__device__ __forceinline__ float key_bias(const AttnArgs& a, const uint8_t* km, int key) {
    return (key < a.Lk && (km == nullptr || km[key] != 0)) ? 0.f : -INFINITY;
}

// Forward: a WAVE owns a (batch, head) pair -- its V image and staging tile are private, so nothing is synchronised and a
// compute unit interleaves ~10 independent waves (the first version gave a pair to a 4-wave workgroup: half the waves had
// no query block at S = 56, 354 registers kept it at one workgroup per CU, and every workgroup paid load -> barrier -> load
// latencies in sequence: 208 us per call against 104 for the library kernel).
// BIAS: a.bias != nullptr -- scores = scale * q k^T + bias[h][i][j] (T5)
template <int T, bool BIAS>
__global__ __launch_bounds__(AT_FW * 64, (T <= 2 ? 3 : 2)) void attn_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.x * AT_FW + wave;
    if (bh >= a.B * a.H) return;
    const uint64_t seed = a.thr ? vlpet_eff_seed(a.seed, a.seed_ctr) : 0;
    const int b = bh / a.H, h = bh % a.H;
    const int m = lane & 31, hh = lane >> 5;
    constexpr int Lkp = 32 * T;
    const int64_t rs = (int64_t)a.H * 64, rq = a.ld_q, rk = a.ld_kv, rv = a.ld_v;
    const __bf16* qb_ = a.q + (int64_t)b * a.Lq * rq + h * 64;
    const __bf16* kb_ = a.k + (int64_t)b * a.Lk * rk + h * 64;
    const __bf16* vb_ = a.v + (int64_t)b * a.Lk * rv + h * 64;
    __bf16* ob_ = a.o + (int64_t)b * a.Lq * rs + h * 64;
    const uint8_t* km = a.key_mask ? a.key_mask + (int64_t)b * a.Lk : nullptr;
    uint8_t* Vs = smem + (size_t)wave * AttnLds::fwd_wave_bytes(Lkp);
    uint8_t* stg = Vs + (size_t)Lkp * AT_ROW;
    float* kval = reinterpret_cast<float*>(stg + AT_STG);            // per key: 0 / -inf
#pragma unroll
    for (int c = 0; c < (Lkp + 63) / 64; ++c)
        if (lane + 64 * c < Lkp) kval[lane + 64 * c] = key_bias(a, km, lane + 64 * c);
    const int coff = a.causal ? a.Lk - a.Lq : (1 << 20);           // key j is visible to query i iff j <= i + coff

    bf16x8 kf[T][4];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int j = 32 * t + m, jk = j < a.Lk ? j : a.Lk - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[t][ks] = *reinterpret_cast<const bf16x8*>(kb_ + (int64_t)jk * rk + 16 * ks + 8 * hh);
    }
    stage_image(Vs, vb_, rv, a.Lk, Lkp, lane, 64);

    const float sc2 = a.scale * LOG2E;
    const int NQB = (a.Lq + 31) >> 5;
    for (int qb = 0; qb < NQB; ++qb) {
        const int i = 32 * qb + m;
        const int iq = i < a.Lq ? i : a.Lq - 1;
        bf16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qb_ + (int64_t)iq * rq + 16 * ks + 8 * hh);
        f32x16 st[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if constexpr (BIAS) st[t] = bias_tile(a.bias + ((int64_t)h * (32 * NQB) + i) * Lkp + 32 * t + 4 * hh, 1.0f / a.scale);
            else st[t] = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) st[t] = mfma32(kf[t][ks], qf[ks], st[t]);
        }
        // ---- softmax over the keys of query i (this lane and lane ^ 32 hold them); masked keys: -inf from the table / the causal bound
        float mx = -INFINITY;
        const int ic = i + coff - 4 * hh;                          // key 32 t + ir + 4 hh is masked iff 32 t + ir > ic
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(kval + 32 * t + 8 * q4 + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q4 + e;
                    float s = fmaf(st[t][r], sc2, kv[e]);
                    s = 32 * t + e + 8 * q4 > ic ? -INFINITY : s;
                    st[t][r] = s;
                    mx = fmaxf(mx, s);
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (mx == -INFINITY) mx = 0.f;               // a row with no key to attend to: all-zero probabilities
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float p = fast_exp2(st[t][r] - mx); st[t][r] = p; sum += p; }
        }
        sum += __shfl_xor(sum, 32);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
        if (hh == 0 && i < a.Lq) a.lse[((int64_t)b * a.H + h) * a.Lq + i] = sum > 0.f ? mx + log2f(sum) : INFINITY;
        // ---- dropout (the hash always runs: threshold 0 keeps everything), probabilities -> B operands
        const uint32_t rk = row_key(seed, ((int64_t)b * a.H + h) * a.Lq + iq);
        const float inv_keep = a.thr ? a.inv_keep : 1.0f;
        f32x16 ot0 = zero16(), ot1 = zero16();
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (a.keep_out != nullptr && a.thr != 0) {             // (tests: export of the mask)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (i < a.Lq && key < a.Lk)
                        a.keep_out[(((int64_t)b * a.H + h) * a.Lq + i) * a.Lk + key] = keep_elem(rk, key, a.thr) ? 1 : 0;
                }
            }
            const uint32_t kg0 = rk + (uint32_t)(32 * t + 4 * hh) * 0x9E3779B9U;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ir = (r & 3) + 8 * (r >> 2);
                const bool kp = hash_elem(kg0 + (uint32_t)ir * 0x9E3779B9U) >= a.thr;
                st[t][r] = kp ? st[t][r] * inv * inv_keep : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8 pf = acc_frag(st[t], u);
                ot0 = mfma32(tr_acc_order(Vs, 32 * t + 16 * u, 0, lane), pf, ot0);
                ot1 = mfma32(tr_acc_order(Vs, 32 * t + 16 * u, 32, lane), pf, ot1);
            }
        }
        store_rows_T(stg, ot0, ot1, ob_, rs, 32 * qb, a.Lq, lane);
    }
}

// ------------------------------------------------------------------------------------------------ backward: the two work units
// Shared by both backward kernels (R = how a fragment is read from the kernel's LDS images).  The elementwise part is written
// without control flow: the first version tested key_ok() per element (bounds, causal, a global load of the key mask) and the
// dropout flag, which hipcc turned into ~100 exec-mask branches per unit with the mask load inside them -- every element a
// serialised round trip, 7 us of issue time per unit.  Here a masked key is an additive -inf from a per-pair LDS table (built once
// per pair from the key mask), the causal test a compare + select, a padded query row has lse = +inf, exp2(-inf) = 0 does the rest,
// and the dropout hash always runs (threshold 0 keeps everything).
enum { IMG_Q = 0, IMG_D = 1, IMG_K = 2, IMG_V = 3 };
struct BwdCtx {
    const float* rowt;      // [3][Lqp] per query row: lse2 (+inf for rows past Lq) | delta | dropout row key
    const float* kval;      // [Lkp] 0 for a key that may be attended to, -inf for a masked one and for keys past Lk
    int Lqp;
    float sc2, scale, inv_keep;
    uint32_t thr;
    int coff;               // key j is visible to query i iff j <= i + coff (Lk - Lq when causal, else out of reach)
};
__device__ __forceinline__ BwdCtx bwd_ctx(const AttnArgs& a, const float* rowt, const float* kval, int Lqp) {
    BwdCtx c;
    c.rowt = rowt; c.kval = kval; c.Lqp = Lqp;
    c.sc2 = a.scale * LOG2E; c.scale = a.scale; c.inv_keep = a.thr ? a.inv_keep : 1.0f; c.thr = a.thr;
    c.coff = a.causal ? a.Lk - a.Lq : (1 << 20);
    return c;
}

// phase K: key tile t (lane = key, registers = queries): dV^T, dK^T of the tile
template <class R, bool BIAS>
__device__ __forceinline__ void bwd_k_unit(const AttnArgs& a, const BwdCtx& c, const R& rd0, int t, int NQB, int Lqp, int Lkp, int h, int lane,
                                           f32x16& dv0, f32x16& dv1, f32x16& dk0, f32x16& dk1) {
    const int m = lane & 31, hh = lane >> 5;
    const int key = 32 * t + m;
    const float kb = c.kval[key];
    const uint32_t kg = (uint32_t)key * 0x9E3779B9U;
    const int kc = key - c.coff - 4 * hh;                  // masked iff key > i + coff with i = 32 qb + ir + 4 hh, i.e. kc - 32 qb > ir
    for (int qb = 0; qb < NQB; ++qb) {
        const R rd = rd0.fresh();                          // (register diet: nothing of the reads below is hoisted out of this loop)
        f32x16 s, dp = zero16();
        // bias[h][i][key] for the lane's key and its 16 queries: four runs of four along the transposed copy
        if constexpr (BIAS) s = bias_tile(a.bias_t + ((int64_t)h * Lkp + key) * Lqp + 32 * qb + 4 * hh, 1.0f / a.scale);
        else s = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = mfma32(rd.nat(IMG_Q, qb, ks), rd.nat(IMG_K, t, ks), s);          // D[query][key]
            dp = mfma32(rd.nat(IMG_D, qb, ks), rd.nat(IMG_V, t, ks), dp);
        }
        const int kcq = kc - 32 * qb;
        const float* rtp = c.rowt + 32 * qb + 4 * hh;      // the lane half's queries 32 qb + 8 q4 + 4 hh + e: four runs of four
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            __builtin_amdgcn_sched_barrier(0);             // (one run at a time: 12 table values live, not 48)
            const f32x4 ls = *reinterpret_cast<const f32x4*>(rtp + 8 * q4);
            const f32x4 dl = *reinterpret_cast<const f32x4*>(rtp + c.Lqp + 8 * q4);
            const f32x4 rk = *reinterpret_cast<const f32x4*>(rtp + 2 * c.Lqp + 8 * q4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * q4 + e, ir = e + 8 * q4;
                float x = fmaf(s[r], c.sc2, kb - ls[e]);
                x = kcq > ir ? -INFINITY : x;
                const float p = fast_exp2(x);
                const bool kp = hash_elem(__float_as_uint(rk[e]) + kg) >= c.thr;
                const float g = kp ? dp[r] * c.inv_keep : 0.f;
                s[r] = kp ? p * c.inv_keep : 0.f;          // P after dropout
                dp[r] = p * (g - dl[e]) * c.scale;         // dS
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16x8 pf = acc_frag(s, u), sf = acc_frag(dp, u);
            dv0 = mfma32(rd.tr(IMG_D, 2 * qb + u, 0), pf, dv0);                  // D[d][key] += dO^T P
            dv1 = mfma32(rd.tr(IMG_D, 2 * qb + u, 1), pf, dv1);
            dk0 = mfma32(rd.tr(IMG_Q, 2 * qb + u, 0), sf, dk0);                  // D[d][key] += Q^T dS
            dk1 = mfma32(rd.tr(IMG_Q, 2 * qb + u, 1), sf, dk1);
        }
    }
}
// phase Q: query block qb (lane = query, registers = keys): dQ^T of the block
template <class R, bool BIAS>
__device__ __forceinline__ void bwd_q_unit(const AttnArgs& a, const BwdCtx& c, const R& rd0, int qb, int T, int Lqp, int Lkp, int h, int lane,
                                           f32x16& dq0, f32x16& dq1) {
    const int m = lane & 31, hh = lane >> 5;
    const int i = 32 * qb + m;
    bf16x8 qf[4], df[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { qf[ks] = rd0.nat(IMG_Q, qb, ks); df[ks] = rd0.nat(IMG_D, qb, ks); }
    const float l2 = c.rowt[i], dl = c.rowt[c.Lqp + i];
    const uint32_t rkey = __float_as_uint(c.rowt[2 * c.Lqp + i]);
    const int ic = i + c.coff - 4 * hh;                    // masked iff key > i + coff with key = 32 t + ir + 4 hh, i.e. 32 t + ir > ic
    for (int t = 0; t < T; ++t) {
        const R rd = rd0.fresh();
        f32x16 s, dp = zero16();
        if constexpr (BIAS) s = bias_tile(a.bias + ((int64_t)h * Lqp + i) * Lkp + 32 * t + 4 * hh, 1.0f / a.scale);
        else s = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = mfma32(rd.nat(IMG_K, t, ks), qf[ks], s);                          // D[key][query]
            dp = mfma32(rd.nat(IMG_V, t, ks), df[ks], dp);
        }
        const float* kvp = c.kval + 32 * t + 4 * hh;
        const uint32_t kg0 = rkey + (uint32_t)(32 * t + 4 * hh) * 0x9E3779B9U;
        const int ict = ic - 32 * t;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 kv = *reinterpret_cast<const f32x4*>(kvp + 8 * q4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * q4 + e, ir = e + 8 * q4;
                float x = fmaf(s[r], c.sc2, kv[e] - l2);
                x = ir > ict ? -INFINITY : x;
                const float p = fast_exp2(x);
                const bool kp = hash_elem(kg0 + (uint32_t)ir * 0x9E3779B9U) >= c.thr;
                const float g = kp ? dp[r] * c.inv_keep : 0.f;
                dp[r] = p * (g - dl) * c.scale;            // dS in place of dP
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16x8 sf = acc_frag(dp, u);
            dq0 = mfma32(rd.tr(IMG_K, 2 * t + u, 0), sf, dq0);                    // D[d][query] += K^T dS
            dq1 = mfma32(rd.tr(IMG_K, 2 * t + u, 1), sf, dq1);
        }
    }
}
// fragment readers: padded 144-byte rows (attn_bwd_kernel) ...
struct PadRd {
    const uint8_t* img[4];
    int m, hh, lane;
    __device__ __forceinline__ PadRd fresh() const { PadRd r = *this; asm volatile("" : "+v"(r.m), "+v"(r.lane)); return r; }
    __device__ __forceinline__ bf16x8 nat(int im, int tile, int ks) const { return nat_frag(img[im], 32 * tile + m, ks, hh); }
    __device__ __forceinline__ bf16x8 tr(int im, int g16, int dt) const { return tr_acc_order(img[im], 16 * g16, 32 * dt, lane); }
};
// Backward: a workgroup owns a (batch, head) pair (the four images are shared by its waves).  Work units: one per key tile
// (phase K: dK, dV of the tile) and one per query block (phase Q: dQ of the block), handed out round-robin, so that at
// S = 56 (two tiles, two blocks) each of the four waves has exactly one and nothing is computed twice.
// NW = waves per workgroup: 4, or 6 for sequences of 65-96 tokens (three key tiles + three query blocks = six units: one round
// instead of a full one and a half-empty one; two 6-wave workgroups per CU = three waves per SIMD, the OCC = 3 register budget)
template <int OCC, int NW = AT_NW, bool BIAS = false>
__global__ __launch_bounds__(NW * 64, OCC) void attn_bwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t seed = a.thr ? vlpet_eff_seed(a.seed, a.seed_ctr) : 0;
    const int m = lane & 31, hh = lane >> 5;
    const int Lkp = (a.Lk + 31) & ~31, T = Lkp >> 5;
    const int Lqp = (a.Lq + 31) & ~31, NQB = Lqp >> 5;
    const int64_t rs = (int64_t)a.H * 64, rq = a.ld_q, rk = a.ld_kv, rv = a.ld_v;
    const int64_t ooff = (int64_t)b * a.Lq * rs + h * 64;                                            // o, dout
    const int64_t qoff = (int64_t)b * a.Lq * rq + h * 64, koff = (int64_t)b * a.Lk * rk + h * 64;    // q / dq, k / dk
    const int64_t voff = (int64_t)b * a.Lk * rv + h * 64;                                            // v / dv
    const uint8_t* km = a.key_mask ? a.key_mask + (int64_t)b * a.Lk : nullptr;
    uint8_t* Qs = smem;
    uint8_t* Ds = Qs + (size_t)Lqp * AT_ROW;                     // dO
    uint8_t* Ks = Ds + (size_t)Lqp * AT_ROW;
    uint8_t* Vs = Ks + (size_t)Lkp * AT_ROW;
    uint8_t* stg = Vs + (size_t)Lkp * AT_ROW + (size_t)wave * AT_STG;
    float* rowt = reinterpret_cast<float*>(Vs + (size_t)Lkp * AT_ROW + (size_t)NW * AT_STG);   // [3][Lqp] per query row: lse2 | delta | row key
    float* kval = rowt + 3 * Lqp;                                                                // per key: 0 / -inf

    {
        // delta[i] = sum_d dO[i][d] O[i][d]: four threads per row, 128 rows at most = two rows per thread quad; loads first
        bf16x8 xo[2][2], xd[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int idx = tid + c * NW * 64, row = idx >> 2, part = idx & 3;
            const int rr = row < a.Lq ? row : a.Lq - 1;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                xo[c][e] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(a.o + ooff + (int64_t)rr * rs + 16 * part + 8 * e));
                xd[c][e] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(a.dout + ooff + (int64_t)rr * rs + 16 * part + 8 * e));
            }
        }
        if (tid < Lkp) kval[tid] = key_bias(a, km, tid);
        uint8_t* const imgs[4] = {Qs, Ds, Ks, Vs};
        const __bf16* const srcs[4] = {a.q + qoff, a.dout + ooff, a.k + koff, a.v + voff};
        const int nr[4] = {a.Lq, a.Lq, a.Lk, a.Lk}, rp[4] = {Lqp, Lqp, Lkp, Lkp};
        const int64_t rss[4] = {rq, rs, rk, rv};
        stage_images<4, (NW == 6 ? 3 : 4)>(imgs, srcs, nr, rp, rss, tid, NW * 64);          // 128 rows x 8 pieces / 256 threads = 4 per image
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int idx = tid + c * NW * 64, row = idx >> 2, part = idx & 3;
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += (float)xo[c][e][j] * (float)xd[c][e][j];
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            if (part == 0 && row < Lqp) {
                const bool live = row < a.Lq;
                rowt[row] = live ? a.lse[((int64_t)b * a.H + h) * a.Lq + row] : INFINITY;
                rowt[Lqp + row] = live ? acc : 0.f;
                rowt[2 * Lqp + row] = __uint_as_float(row_key(seed, ((int64_t)b * a.H + h) * a.Lq + (live ? row : a.Lq - 1)));
            }
        }
    }
    __syncthreads();

    const BwdCtx c = bwd_ctx(a, rowt, kval, Lqp);
    PadRd rd;
    rd.img[IMG_Q] = Qs; rd.img[IMG_D] = Ds; rd.img[IMG_K] = Ks; rd.img[IMG_V] = Vs;
    rd.m = m; rd.hh = hh; rd.lane = lane;
    for (int un = (a.dbg & 1) ? T + NQB : wave; un < T + NQB; un += NW) {
        if (un < T) {
            f32x16 dv0 = zero16(), dv1 = zero16(), dk0 = zero16(), dk1 = zero16();
            if (!(a.dbg & 8)) bwd_k_unit<PadRd, BIAS>(a, c, rd, un, NQB, Lqp, Lkp, h, lane, dv0, dv1, dk0, dk1);
            if (a.dbg & 4) { if (dv0[0] + dv1[1] + dk0[2] + dk1[3] == 1.2345f) a.dv[0] = (__bf16)1.f; continue; }
            store_rows_T(stg, dv0, dv1, a.dv + voff, rv, 32 * un, a.Lk, lane);
            store_rows_T(stg, dk0, dk1, a.dk + koff, rk, 32 * un, a.Lk, lane);
        } else {
            f32x16 dq0 = zero16(), dq1 = zero16();
            if (!(a.dbg & 8)) bwd_q_unit<PadRd, BIAS>(a, c, rd, un - T, T, Lqp, Lkp, h, lane, dq0, dq1);
            if (a.dbg & 4) { if (dq0[0] + dq1[1] == 1.2345f) a.dq[0] = (__bf16)1.f; continue; }
            store_rows_T(stg, dq0, dq1, a.dq + qoff, rq, 32 * (un - T), a.Lq, lane);
        }
    }
}

size_t attn_lds_bytes(int Lq, int Lk, int bwd) {
    const int Lqp = (Lq + 31) & ~31, Lkp = (Lk + 31) & ~31;
    return bwd ? AttnLds::bwd_bytes(Lqp, Lkp) : AttnLds::fwd_bytes(Lkp);
}

template <int T, bool BIAS>
static hipError_t launch_attn_fwd_tb(const AttnArgs& a, hipStream_t stream) {
    const size_t lds = AttnLds::fwd_bytes(32 * T);
    auto kern = attn_fwd_kernel<T, BIAS>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const unsigned pairs = (unsigned)(a.B * a.H);
    hipLaunchKernelGGL(kern, dim3((pairs + AT_FW - 1) / AT_FW), dim3(AT_FW * 64), lds, stream, a);
    return hipGetLastError();
}

template <int T>
static hipError_t launch_attn_fwd_t(const AttnArgs& a, hipStream_t stream) {
    return a.bias != nullptr ? launch_attn_fwd_tb<T, true>(a, stream) : launch_attn_fwd_tb<T, false>(a, stream);
}

hipError_t launch_attn(const AttnArgs& a_in, bool bwd, hipStream_t stream) {
    AttnArgs a = a_in;
    a.dbg = VLPET_IS_DEBUG_BUILD ? vlpet_tuning().dbg : 0;
    if (!bwd) {
        switch ((a.Lk + 31) >> 5) {
            case 1: return launch_attn_fwd_t<1>(a, stream);
            case 2: return launch_attn_fwd_t<2>(a, stream);
            case 3: return launch_attn_fwd_t<3>(a, stream);
            case 4: return launch_attn_fwd_t<4>(a, stream);
            default: return hipErrorInvalidValue;
        }
    }
    const int Lqp = (a.Lq + 31) & ~31, Lkp = (a.Lk + 31) & ~31;
    const int units = (Lqp >> 5) + (Lkp >> 5);
    const int occ_env = vlpet_tuning().attn_occ;
    const int nw_env = vlpet_tuning().attn_nw;
    // Five or six units (sequences of 65-96 tokens) on six waves, one unit each, two such workgroups per CU (three waves per SIMD on
    // the 168-register build) instead of four waves taking a full round and a half-empty one: parity-green and SLOWER (238 vs 203 us
    // at B = 416, S = 76; 97 vs 92 at B = 166, S = 92, profiles/r02_attnbench2_s4_six_waves.txt) -- the workgroup still stages, waits,
    // computes and stores in sequence, and the longer compute phase was not what bounded it.  Kept behind VLPET_ATTN_NW=6 for A/B.
    const size_t lds6 = AttnLds::bwd_bytes_nw(Lqp, Lkp, 6);
    if (nw_env == 6 && (units == 5 || units == 6) && 2 * (lds6 + 512) <= (size_t)160 * 1024) {
        auto k6 = attn_bwd_kernel<3, 6>;
        hipError_t e6 = hipFuncSetAttribute(reinterpret_cast<const void*>(k6), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds6);
        if (e6 != hipSuccess) return e6;
        hipLaunchKernelGGL(k6, dim3((unsigned)(a.B * a.H)), dim3(6 * 64), lds6, stream, a);
        return hipGetLastError();
    }
    const size_t lds = attn_lds_bytes(a.Lq, a.Lk, 1);
    if (a.bias != nullptr) {       // T5's biased scores: same occupancy rule as below (VLPET_ATTN_OCC forces one in a diagnosis build)
        const bool o3 = occ_env == 3 || (occ_env != 2 && 3 * (lds + 512) <= (size_t)160 * 1024);
        const void* kb = o3 ? reinterpret_cast<const void*>(attn_bwd_kernel<3, AT_NW, true>) : reinterpret_cast<const void*>(attn_bwd_kernel<2, AT_NW, true>);
        hipError_t eb = hipFuncSetAttribute(kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (eb != hipSuccess) return eb;
        if (o3) hipLaunchKernelGGL((attn_bwd_kernel<3, AT_NW, true>), dim3((unsigned)(a.B * a.H)), dim3(AT_NW * 64), lds, stream, a);
        else hipLaunchKernelGGL((attn_bwd_kernel<2, AT_NW, true>), dim3((unsigned)(a.B * a.H)), dim3(AT_NW * 64), lds, stream, a);
        return hipGetLastError();
    }
    // Three waves per SIMD (168 registers, one spilled) whenever three workgroups fit the CU's LDS -- sequences of at most 64
    // tokens: the memory phase of a pair then overlaps the compute phase of two others (142 -> 114 us at B = 500, S = 56);
    // longer sequences (two workgroups per CU by LDS either way) keep the 171-register build.  VLPET_ATTN_OCC = 2 | 3 forces one.
    const bool occ3 = occ_env == 3 || (occ_env != 2 && 3 * (lds + 512) <= (size_t)160 * 1024);
    const void* kern = occ3 ? reinterpret_cast<const void*>(attn_bwd_kernel<3>) : reinterpret_cast<const void*>(attn_bwd_kernel<2>);
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (occ3) hipLaunchKernelGGL(attn_bwd_kernel<3>, dim3((unsigned)(a.B * a.H)), dim3(AT_NW * 64), lds, stream, a);
    else hipLaunchKernelGGL(attn_bwd_kernel<2>, dim3((unsigned)(a.B * a.H)), dim3(AT_NW * 64), lds, stream, a);
    return hipGetLastError();
}
