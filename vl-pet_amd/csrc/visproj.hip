// K4 forward: visual-feature projection with fused bias + LayerNorm (+ residual add)
//
//     out = LN( feats . W^T + b ) * gamma + beta  (+ R)          feats [M, F], W [d_out, F]
//
// Reference: the feat_embedding branch of VisualEmbedding.forward (src/modeling_bart.py:157, the
// nn.Sequential(Linear(feat_dim, d_model), LayerNorm) built at :91-110; T5: src/modeling_t5.py:56-66 with
// T5LayerNorm); R carries the position branch + order embeddings (:162-183).
//
// A real contraction (K = 2048, 560 flop/B): MFMA-bound.  Same swapped-operand structure and memory
// system as the adapter kernels (pet32.h): workgroup = 4 waves x 32 rows, every wave owns ALL d_out
// output features of its rows (d_out/32 accumulator tiles = 384 registers at d_out = 768), so the
// LayerNorm statistics need one cross-lane exchange (lane ^ 32) and no LDS.  W arrives as pre-packed A
// fragments (pack_down4 layout with r = d_out) in sub-stages of 16 input features through a 2-slot ring,
// the feature rows 128 bytes per row at a time through the 3-slot ring.  The epilogue walks the row in
// 128-byte chunks: R in by global_load_lds, out / xhat staged in LDS and stored as whole lines.
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "pet32.h"

template <typename IO, int NCT, int WAVES>
struct VisLds {
    static constexpr int NS = Geo4<IO>::NS;
    static constexpr int WSUB_B = NCT * NS * 1024;           // one 16-feature sub-stage of W
    static constexpr int TILE_B = WAVES * 32 * 128;
    // weight ring: as many sub-stage slots as fit next to the 3-slot row ring and the parameters (d_out = 768, bf16: 4 slots of
    // 24 KiB = three sub-stages, ~2.3k MFMA cycles, of prefetch distance; with 2 slots every sub-stage waited ~1 us for
    // its fragments: 241 us per launch at M = 18.7k, 10 % of the MFMA peak)
    static constexpr int NW_MAX = (160 * 1024 - 3 * TILE_B - 3 * 32 * NCT * 4 - 1024) / WSUB_B;
    static constexpr int NW = NW_MAX < 2 ? 2 : (NW_MAX > 6 ? 6 : NW_MAX);
    static constexpr int ROW_OFF = NW * WSUB_B;
    static constexpr int MAIN_B = ROW_OFF + 3 * TILE_B;
    static constexpr int EPI_B = 4 * TILE_B;                 // R0, R1, OUT, XHAT tiles (alias the rings)
    static constexpr int PARAM_OFF = MAIN_B > EPI_B ? MAIN_B : EPI_B;
    static size_t bytes(int d_out) { return (size_t)PARAM_OFF + (size_t)3 * d_out * 4; }
};

// 8 consecutive features of the lane's row inside a 128-byte chunk tile: index idx8 (bf16: piece idx8; fp32: two pieces)
template <typename IO>
__device__ __forceinline__ void put8(uint8_t* tile, int trow, int idx8, const float* v) {
    if constexpr (Geo4<IO>::NS == 1) {
        bf16x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = (__bf16)v[j];
        *reinterpret_cast<bf16x8*>(const_cast<uint8_t*>(tile_piece(tile, trow, idx8))) = t;
    } else {
        const f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4*>(const_cast<uint8_t*>(tile_piece(tile, trow, 2 * idx8))) = a;
        *reinterpret_cast<f32x4*>(const_cast<uint8_t*>(tile_piece(tile, trow, 2 * idx8 + 1))) = b;
    }
}
template <typename IO>
__device__ __forceinline__ void get8(const uint8_t* tile, int trow, int idx8, float* v) {
    if constexpr (Geo4<IO>::NS == 1) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(tile_piece(tile, trow, idx8));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
    } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(tile_piece(tile, trow, 2 * idx8));
        const f32x4 b = *reinterpret_cast<const f32x4*>(tile_piece(tile, trow, 2 * idx8 + 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
    }
}

template <typename IO, int NCT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void visproj_fwd_kernel(VisprojArgs a) {
    using G = Geo4<IO>;
    using L = VisLds<IO, NCT, WAVES>;
    constexpr int NS = G::NS;
    constexpr int GB = NCT < 12 ? NCT : 12;      // A fragments read per LDS burst
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar registers and scalar branches for everything derived from it
    const int m = lane & 31, h = lane >> 5;
    const int trow = 32 * wave + m;
    const int F = a.F, d_out = 32 * NCT;
    const int64_t row0_wave = (int64_t)blockIdx.x * (WAVES * 32) + wave * 32;
    const int S = F / G::FE;                     // row stages
    const int Q = S * G::KU;                     // weight sub-stages
    const uint8_t* feats = reinterpret_cast<const uint8_t*>(a.feats);
    const RowLanes rl = row_lanes<IO>(row0_wave, a.M, F, wave, lane);
    const RowLanes rlo = row_lanes<IO>(row0_wave, a.M, d_out, wave, lane);
    const int lane16 = lane * 16;

    auto slot_w = [&](int j) { return smem + (size_t)j * L::WSUB_B; };
    auto slot_t = [&](int j) { return smem + L::ROW_OFF + (size_t)j * L::TILE_B; };
    float* sp = reinterpret_cast<float*>(smem + L::PARAM_OFF);      // bias | gamma | beta
    auto issue_rows = [&](int s2) { if (s2 < S) glds_rows4(feats, rl, s2 * 128, slot_t(s2 % 3), wave); };
    auto issue_w = [&](int q1) {
        if (q1 >= Q) return;
        const uint8_t* src0 = a.pk + (int64_t)q1 * L::WSUB_B;
        uint8_t* dst = slot_w(q1 % L::NW);
        constexpr int KB = NCT * NS;
        for (int k = wave; k < KB; k += WAVES) glds16(src0 + (size_t)k * 1024 + lane16, dst + (size_t)k * 1024);
    };

    constexpr int KBW = NCT * NS;                                     // 1 KiB weight pieces per sub-stage
    // pieces this wave issues per sub-stage (k = wave, wave + WAVES, ...): the counted waits below need the exact number
    const int npw = (KBW - wave + WAVES - 1) / WAVES;
#pragma unroll
    for (int q0 = 0; q0 < L::NW - 1; ++q0) issue_w(q0);
    issue_rows(0);
    issue_rows(1);
    {
        const float* bias = reinterpret_cast<const float*>(a.pk + (int64_t)(F / 16) * NCT * NS * 1024);
        for (int i = tid; i < d_out; i += WAVES * 64) {
            sp[i] = bias[i];
            sp[d_out + i] = a.gamma[i];
            sp[2 * d_out + i] = a.beta ? a.beta[i] : 0.f;
        }
    }
    __syncthreads();

    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = zero16();
    for (int q = 0; q < Q; ++q) {
        const int s = q / G::KU, u = q % G::KU;
        issue_w(q + L::NW - 1);
        if (u == 0) issue_rows(s + 2);
        const uint8_t* w = slot_w(q % L::NW);
        const Frag<NS> b = tile_bfrag4<IO>(slot_t(s % 3), trow, h, u);
#pragma unroll
        for (int g0 = 0; g0 < NCT; g0 += GB) {
            Frag<NS> wa[GB];
#pragma unroll
            for (int i = 0; i < GB; ++i) wa[i] = wfrag<NS>(w, g0 + i, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < GB; ++i) acc[g0 + i] = mfma_ns<NS>(wa[i], b, acc[g0 + i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // In flight, oldest first: weights of q+1 .. q+NW-1 (npw pieces each) with the row pieces of stage s+2 (4, issued
        // right after the weights of q+NW-1 when u == 0) in between.  The next sub-stage needs the weights of q+1 and, when
        // it starts a new row stage, rows issued a whole stage (KU sub-stages) ago -- older than everything kept here.
        {
            int keep = 0;
            for (int j = 2; j <= L::NW - 1; ++j) if (q + j < Q) keep += npw;
            // row pieces of stage s+2 were issued at u == 0 of this stage, after the weights of (first sub-stage of s) + NW - 1:
            // they are younger than the weights of q+1 iff that sub-stage index is >= q+1, i.e. u <= NW - 2
            if (s + 2 < S && u <= L::NW - 2) keep += 4;
            wait_vm(keep);
        }
        __builtin_amdgcn_s_barrier();
    }

    // ---- bias, LayerNorm statistics (lane + partner lane^32 hold the whole row)
    const float* bias = sp + 8 * h;
    const float* gam = sp + d_out + 8 * h;
    const float* bet = sp + 2 * d_out + 8 * h;
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[ct][i] += bias[32 * ct + 16 * (i >> 3) + (i & 7)];
            sum += acc[ct][i];
        }
    }
    sum += __shfl_xor(sum, 32);
    const float mean = a.rms ? 0.f : sum / (float)d_out;
    float sq = 0.f;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float t = acc[ct][i] - mean; sq += t * t; }
    }
    sq += __shfl_xor(sq, 32);
    const float rstd = rsqrtf(sq / (float)d_out + a.eps);
    if (a.rstd != nullptr && h == 0 && row0_wave + m < a.M) a.rstd[row0_wave + m] = rstd;

    // ---- epilogue in 128-byte chunks of the output row
    constexpr int TPC = G::FE / 32;              // accumulator tiles per chunk
    constexpr int NCH = NCT / TPC;               // chunks
    uint8_t* tR[2] = {smem, smem + L::TILE_B};
    uint8_t* tO = smem + 2 * L::TILE_B;
    uint8_t* tX = smem + 3 * L::TILE_B;
    const uint8_t* Rg = reinterpret_cast<const uint8_t*>(a.R);
    uint8_t* Og = reinterpret_cast<uint8_t*>(a.out);
    uint8_t* Xg = reinterpret_cast<uint8_t*>(a.xhat);
    const int n_st = rlo.n_inst * (Xg ? 2 : 1);   // store instructions per chunk
    __syncthreads();                              // everyone is done with the rings before they are re-used
    if (Rg) glds_rows4(Rg, rlo, 0, tR[0], wave);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (Rg) {
            if (ch + 1 < NCH) glds_rows4(Rg, rlo, (ch + 1) * 128, tR[(ch + 1) & 1], wave);
            // R(ch) is older than the previous chunk's stores and the prefetch just issued
            wait_vm((ch + 1 < NCH ? 4 : 0) + (ch > 0 ? n_st : 0));
        }
#pragma unroll
        for (int tl = 0; tl < TPC; ++tl) {
            const int ct = ch * TPC + tl;
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                const int idx8 = (TPC == 2 ? 4 * tl : 0) + 2 * sh + h;
                float xv[8], ov[8], rv[8];
                if (Rg) get8<IO>(tR[ch & 1], trow, idx8, rv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = 32 * ct + 16 * sh + j;            // + 8h folded into gam / bet
                    xv[j] = (acc[ct][8 * sh + j] - mean) * rstd;
                    ov[j] = xv[j] * gam[c] + bet[c] + (Rg ? rv[j] : 0.f);
                }
                put8<IO>(tO, trow, idx8, ov);
                if (Xg) put8<IO>(tX, trow, idx8, xv);
            }
        }
        store_rows4(Og, rlo, ch * 128, tO, wave, lane);
        if (Xg) store_rows4(Xg, rlo, ch * 128, tX, wave, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // staging tiles are re-written next chunk
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Column-split form for d_out = 768 (the real shape): 8 waves = 4 row groups x 2 column halves.
//
// The 4-wave kernel above keeps all d_out / 32 = 24 accumulator tiles of its 32 rows in one wave: 384 of the 512
// registers a lane of a SIMD has, and hipcc spills an accumulator tile INSIDE the sub-stage loop (75-180 registers in
// all).  Scratch traffic retires in order with the LDS-DMA prefetches on vmcnt, so every reload waited for the weight
// stream just requested: 1.9 us per 16-feature sub-stage against 0.37 us of MFMA issue -- 241 us per launch at
// M = 18.7k, 10 % of the MFMA peak.  Here a wave owns 12 tiles (192 registers, two waves per SIMD, nothing spilled):
// same rows per workgroup, same weight bytes per MFMA (the fragment stream is shared by the eight waves, the row
// tile by the two halves of a row group), the LayerNorm statistics cross the two halves through LDS (two exchanges:
// mean, then centred variance -- the reference's two-pass order).
#ifndef VLPET_K4_GB
#define VLPET_K4_GB 3
#endif
#ifndef VLPET_K4_ABL
#define VLPET_K4_ABL 0
#endif
#ifndef VLPET_K4_SPLIT_ISSUE
#define VLPET_K4_SPLIT_ISSUE 0
#endif
template <typename IO, int NCT>
struct VisLds2 {
    static constexpr int NS = Geo4<IO>::NS;
    static constexpr int RGN = 4, CH = 2, WAVES = RGN * CH;
    static constexpr int WSUB_B = NCT * NS * 1024;           // one 16-feature sub-stage of W (all d_out columns)
    static constexpr int TILE_B = RGN * 32 * 128;
    static constexpr int STAT_B = CH * RGN * 32 * 4;          // one float per row and column half
    static constexpr int NW_MAX = (160 * 1024 - 3 * TILE_B - 3 * 32 * NCT * 4 - 2 * STAT_B - 1024) / WSUB_B;
    static constexpr int NW = NW_MAX < 2 ? 2 : (NW_MAX > 6 ? 6 : NW_MAX);
    static constexpr int ROW_OFF = NW * WSUB_B;
    static constexpr int MAIN_B = ROW_OFF + 3 * TILE_B;
    static constexpr int EPI_B = CH * 4 * TILE_B;             // per column half: R0, R1, OUT, XHAT tiles (alias the rings)
    static constexpr int STAT_OFF = MAIN_B > EPI_B ? MAIN_B : EPI_B;
    static constexpr int PARAM_OFF = STAT_OFF + 2 * STAT_B;
    static size_t bytes(int d_out) { return (size_t)PARAM_OFF + (size_t)3 * d_out * 4; }
};

template <typename IO, int NCT>
__global__ __launch_bounds__(512) void visproj_fwd2_kernel(VisprojArgs a) {
    using G = Geo4<IO>;
    using L = VisLds2<IO, NCT>;
    constexpr int NS = G::NS;
    constexpr int WAVES = L::WAVES, RGN = L::RGN;
    constexpr int NCW = NCT / L::CH;             // accumulator tiles per wave
    constexpr int GB = NCW % VLPET_K4_GB == 0 ? VLPET_K4_GB : 1;      // A fragments per LDS burst (double-buffered)
    static_assert(NCW % GB == 0, "burst must divide the tiles of a wave");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rgi = wave % RGN, ch = wave / RGN;             // waves rgi and rgi + 4 (the two column halves of a row group) share a SIMD
    const int m = lane & 31, h = lane >> 5;
    const int trow = 32 * rgi + m;
    const int F = a.F, d_out = 32 * NCT;
    const int64_t row0_wave = (int64_t)blockIdx.x * (RGN * 32) + rgi * 32;
    const int S = F / G::FE;
    const int Q = S * G::KU;
    const uint8_t* feats = reinterpret_cast<const uint8_t*>(a.feats);
    const RowLanes rl = row_lanes<IO>(row0_wave, a.M, F, rgi, lane);
    const RowLanes rlo = row_lanes<IO>(row0_wave, a.M, d_out, rgi, lane);
    const int lane16 = lane * 16;

    auto slot_w = [&](int j) { return smem + (size_t)j * L::WSUB_B; };
    auto slot_t = [&](int j) { return smem + L::ROW_OFF + (size_t)j * L::TILE_B; };
    float* sstat = reinterpret_cast<float*>(smem + L::STAT_OFF);    // [2 exchanges][CH][RGN * 32]
    float* sp = reinterpret_cast<float*>(smem + L::PARAM_OFF);      // bias | gamma | beta
    // the row tile of a row group is requested by its column-half-0 wave; the weight pieces by all eight waves
    auto issue_rows = [&](int s2) { if (ch == 0 && s2 < S) glds_rows4(feats, rl, s2 * 128, slot_t(s2 % 3), rgi); };
    constexpr int KBW = NCT * NS;
    auto issue_w = [&](int q1) {
        if (q1 >= Q) return;
        const uint8_t* src0 = a.pk + (int64_t)q1 * L::WSUB_B;
        uint8_t* dst = slot_w(q1 % L::NW);
        if constexpr (L::NW >= 4 && VLPET_K4_SPLIT_ISSUE) {
            // the weight pieces are requested by the column-half-1 waves only, the row pieces by the column-half-0 waves: the two
            // waves of a SIMD then stall in their requests at different times instead of together (the matrix pipe idled through both)
            if (ch == 1)
                for (int k = rgi; k < KBW; k += RGN) glds16(src0 + (size_t)k * 1024 + lane16, dst + (size_t)k * 1024);
        } else {
            for (int k = wave; k < KBW; k += WAVES) glds16(src0 + (size_t)k * 1024 + lane16, dst + (size_t)k * 1024);
        }
    };
    const int npw = wave < KBW ? (KBW - wave + WAVES - 1) / WAVES : 0;     // weight pieces this wave issues per sub-stage
    const int nrow = ch == 0 ? 4 : 0;                                        // row pieces this wave issues per stage

    constexpr bool PAIRS = L::NW >= 4;            // (fp32 IO: two 48-KiB weight slots, one sub-stage per barrier as before)
    if constexpr (PAIRS) { issue_w(0); issue_w(1); }
    else {
#pragma unroll
        for (int q0 = 0; q0 < L::NW - 1; ++q0) issue_w(q0);
    }
    issue_rows(0);
    issue_rows(1);
    {
        const float* bias = reinterpret_cast<const float*>(a.pk + (int64_t)(F / 16) * NCT * NS * 1024);
        for (int i = tid; i < d_out; i += WAVES * 64) {
            sp[i] = bias[i];
            sp[d_out + i] = a.gamma[i];
            sp[2 * d_out + i] = a.beta ? a.beta[i] : 0.f;
        }
    }
    __syncthreads();

    f32x16 acc[NCW];
#pragma unroll
    for (int ct = 0; ct < NCW; ++ct) acc[ct] = zero16();
    // Round 3: the loop walks PAIRS of 16-feature sub-stages -- one counted wait and one barrier per 32 features instead of per
    // 16, and the A fragments of the pair form one chain of bursts (the next burst is requested before the MFMAs of the current
    // one), so that only the first burst of a pair waits for its LDS round trip.  Before: 2 x [6 reads, wait, 6 MFMAs], wait,
    // barrier per sub-stage = 1.7 k cycles for 0.77 k cycles of MFMA issue per SIMD, at any M (a latency chain of 128 links).
    static_assert(G::KU % 2 == 0, "pairs of sub-stages inside a row stage");
    constexpr int NB = NCW / GB;                  // bursts per sub-stage
    if constexpr (PAIRS) {
    for (int q = 0; q < Q; q += 2) {
        const int s = q / G::KU, u = q % G::KU;
#if VLPET_K4_ABL != 2          // (ablation builds: 1 = the stream without the products, 2 = the products without the stream)
        issue_w(q + 2);
        issue_w(q + 3);
#endif
        const bool rows_now = u == 0 && s + 2 < S;
#if VLPET_K4_ABL != 2
        if (u == 0) issue_rows(s + 2);
#endif
        const uint8_t* w0 = slot_w(q % L::NW);
        const uint8_t* w1 = slot_w((q + 1) % L::NW);
        const Frag<NS> b0 = tile_bfrag4<IO>(slot_t(s % 3), trow, h, u);
        const Frag<NS> b1 = tile_bfrag4<IO>(slot_t(s % 3), trow, h, u + 1);
#if VLPET_K4_ABL != 1
        {
            Frag<NS> wa[2][GB];
#pragma unroll
            for (int i = 0; i < GB; ++i) wa[0][i] = wfrag<NS>(w0, ch * NCW + i, lane);
#pragma unroll
            for (int t = 0; t < 2 * NB; ++t) {
                if (t + 1 < 2 * NB) {
                    const uint8_t* wn = (t + 1 < NB) ? w0 : w1;
#pragma unroll
                    for (int i = 0; i < GB; ++i) wa[(t + 1) & 1][i] = wfrag<NS>(wn, ch * NCW + ((t + 1) % NB) * GB + i, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < GB; ++i) acc[(t % NB) * GB + i] = mfma_ns<NS>(wa[t & 1][i], t < NB ? b0 : b1, acc[(t % NB) * GB + i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#else
        acc[0][0] += (float)b0.p[0][0] + (float)b1.p[0][0] + (float)w0[lane] + (float)w1[lane];
#endif
        // this wave's requests in program order: W(q+2) W(q+3) [rows(s+2)]: the next pair's weights must have landed; the
        // rows of the stage after next may stay in flight
        wait_vm(rows_now ? nrow : 0);
        __builtin_amdgcn_s_barrier();
    }
    } else {
    constexpr int GB1 = NCW < 6 ? NCW : 6;
    for (int q = 0; q < Q; ++q) {
        const int s = q / G::KU, u = q % G::KU;
        issue_w(q + L::NW - 1);
        if (u == 0) issue_rows(s + 2);
        const uint8_t* w = slot_w(q % L::NW);
        const Frag<NS> b = tile_bfrag4<IO>(slot_t(s % 3), trow, h, u);
#pragma unroll
        for (int g0 = 0; g0 < NCW; g0 += GB1) {
            Frag<NS> wa[GB1];
#pragma unroll
            for (int i = 0; i < GB1; ++i) wa[i] = wfrag<NS>(w, ch * NCW + g0 + i, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < GB1; ++i) acc[g0 + i] = mfma_ns<NS>(wa[i], b, acc[g0 + i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // (see visproj_fwd_kernel: what may stay in flight while the next sub-stage's weights and rows must have landed)
        {
            int keep = 0;
            for (int j = 2; j <= L::NW - 1; ++j) if (q + j < Q) keep += npw;
            if (s + 2 < S && u <= L::NW - 2) keep += nrow;
            wait_vm(keep);
        }
        __builtin_amdgcn_s_barrier();
    }

    }

    // ---- bias, LayerNorm statistics: this wave holds half of the row (lane + partner lane^32), the partner wave the rest
    const float* bias = sp + 32 * NCW * ch + 8 * h;
    const float* gam = sp + d_out + 32 * NCW * ch + 8 * h;
    const float* bet = sp + 2 * d_out + 32 * NCW * ch + 8 * h;
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < NCW; ++ct) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[ct][i] += bias[32 * ct + 16 * (i >> 3) + (i & 7)];
            sum += acc[ct][i];
        }
    }
    sum += __shfl_xor(sum, 32);
    if (h == 0) sstat[ch * (RGN * 32) + trow] = sum;
    __syncthreads();                              // (also: everyone is done with the rings before they are re-used)
    sum += sstat[(1 - ch) * (RGN * 32) + trow];
    const float mean = a.rms ? 0.f : sum / (float)d_out;
    float sq = 0.f;
#pragma unroll
    for (int ct = 0; ct < NCW; ++ct) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float t = acc[ct][i] - mean; sq += t * t; }
    }
    sq += __shfl_xor(sq, 32);
    float* sstat2 = sstat + L::CH * RGN * 32;
    if (h == 0) sstat2[ch * (RGN * 32) + trow] = sq;
    __syncthreads();
    sq += sstat2[(1 - ch) * (RGN * 32) + trow];
    const float rstd = rsqrtf(sq / (float)d_out + a.eps);
    if (a.rstd != nullptr && ch == 0 && h == 0 && row0_wave + m < a.M) a.rstd[row0_wave + m] = rstd;

    // ---- epilogue in 128-byte chunks of this wave's half of the output row
    constexpr int TPC = G::FE / 32;              // accumulator tiles per chunk
    constexpr int NCHW = NCW / TPC;              // chunks per wave
    uint8_t* ebase = smem + (size_t)ch * 4 * L::TILE_B;
    uint8_t* tR[2] = {ebase, ebase + L::TILE_B};
    uint8_t* tO = ebase + 2 * L::TILE_B;
    uint8_t* tX = ebase + 3 * L::TILE_B;
    const uint8_t* Rg = reinterpret_cast<const uint8_t*>(a.R);
    uint8_t* Og = reinterpret_cast<uint8_t*>(a.out);
    uint8_t* Xg = reinterpret_cast<uint8_t*>(a.xhat);
    const int n_st = rlo.n_inst * (Xg ? 2 : 1);
    const int c0 = ch * NCHW;                    // first chunk of this half
    if (Rg) glds_rows4(Rg, rlo, c0 * 128, tR[0], rgi);
#pragma unroll
    for (int cl = 0; cl < NCHW; ++cl) {
        if (Rg) {
            if (cl + 1 < NCHW) glds_rows4(Rg, rlo, (c0 + cl + 1) * 128, tR[(cl + 1) & 1], rgi);
            wait_vm((cl + 1 < NCHW ? 4 : 0) + (cl > 0 ? n_st : 0));
        }
#pragma unroll
        for (int tl = 0; tl < TPC; ++tl) {
            const int ct = cl * TPC + tl;
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                const int idx8 = (TPC == 2 ? 4 * tl : 0) + 2 * sh + h;
                float xv[8], ov[8], rv[8];
                if (Rg) get8<IO>(tR[cl & 1], trow, idx8, rv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = 32 * ct + 16 * sh + j;            // (+ this half's offset and 8h folded into gam / bet)
                    xv[j] = (acc[ct][8 * sh + j] - mean) * rstd;
                    ov[j] = xv[j] * gam[c] + bet[c] + (Rg ? rv[j] : 0.f);
                }
                put8<IO>(tO, trow, idx8, ov);
                if (Xg) put8<IO>(tX, trow, idx8, xv);
            }
        }
        store_rows4(Og, rlo, (c0 + cl) * 128, tO, rgi, lane);
        if (Xg) store_rows4(Xg, rlo, (c0 + cl) * 128, tX, rgi, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // staging tiles are re-written next chunk
    }
}

template <typename IO, int NCT>
static hipError_t launch_nct2(const VisprojArgs& a, hipStream_t stream) {
    using L = VisLds2<IO, NCT>;
    const size_t lds = L::bytes(a.d_out);
    auto kern = visproj_fwd2_kernel<IO, NCT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int rows = L::RGN * 32;
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.M + rows - 1) / rows)), dim3(512), lds, stream, a);
    return hipGetLastError();
}

template <typename IO, int NCT>
static hipError_t launch_nct(const VisprojArgs& a, hipStream_t stream) {
    constexpr int WAVES = 4;
    using L = VisLds<IO, NCT, WAVES>;
    const size_t lds = L::bytes(a.d_out);
    auto kern = visproj_fwd_kernel<IO, NCT, WAVES>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int rows = WAVES * 32;
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.M + rows - 1) / rows)), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

template <typename IO>
static hipError_t launch_io(const VisprojArgs& a, hipStream_t stream) {
    switch (a.d_out) {
        case 64: return launch_nct<IO, 2>(a, stream);
        case 128: return launch_nct<IO, 4>(a, stream);
        case 768: {
            const bool old_form = vlpet_tuning().k4_waves4 != 0;
            return old_form ? launch_nct<IO, 24>(a, stream) : launch_nct2<IO, 24>(a, stream);      // (VLPET_K4_WAVES4=1: A/B)
        }
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_visproj_fwd(const VisprojArgs& a, int io_fp32, hipStream_t stream) {
    return io_fp32 ? launch_io<float>(a, stream) : launch_io<__bf16>(a, stream);
}
