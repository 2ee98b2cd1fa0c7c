// Pass 1 of the two-pass K1 backward (gated K1 with the forward's saved activations, bf16, r <= 96), second form.
// Autograd of my_transformers/modeling_bart.py:1147-1155,1195-1209 -- same contract as pet_gate_dz_kernel (pet_gate_bwd3.hip):
//     reads dy, x2 (+ the saved z, gelu'), recomputes both up projections, dh, dq tile by tile, contracts
//     dz_a = sd * Wu^T dh, dz_g = Wgu^T dq in registers, writes only dpre_a, dpre_g  [M, 32*RT].
//
// The first form gave a row group (32 rows) to two waves split by CHAIN (adapter / gate): each computed its chain's up
// projection, handed half of it to the partner through LDS in fp32, did half of the elementwise backward, staged dh / dq back
// into the row tiles and contracted its chain -- three barriers per 64-feature stage, every phase waiting for the slowest
// wave, two weight images (up, up_t) per chain in the ring; 43 us at M = 28,000 for an op whose traffic takes 14.
// Here the two waves of a row group split the stage's FEATURES (32 each) and never talk to each other inside the loop:
//   * a wave computes BOTH up projections of its 32 features (MFMA rows = features in natural order, so register rho of lane
//     (m, h) is feature (rho & 3) + 8 (rho >> 2) + 4 h of row m), the whole elementwise backward of those 16 values per lane, and
//     feeds the results -- packed to bf16 in that same register order -- straight back into the matrix cores as the B operand of
//     the contraction over features (a contraction does not care about the order of its k index as long as both operands agree);
//   * the A operand of that contraction comes out of the SAME LDS image of Wu that fed the up projection, by ds_read_b64_tr_b16
//     with the matching row order (rows 16 kappa + 4 h + 0..3 and + 8): one weight image per chain and stage instead of two;
//   * one barrier per stage (the hand-over of the two-slot ring); the two partial dz of a row group meet once, after the loop.
// LDS per stage: Wu, Wgu blocks [64 f x 64*RT B] (gathered from the "up" packs by the source addresses of the LDS-DMA, swizzled
// like pet_cols' bottleneck tiles) + the dy and x2 tiles [128 rows x 128 B] (pet16.h swz); three weight slots + two row slots + the biases = 142 KiB (see the loop for why three).
#include "cols_common.h"

// Cache policy of pass 1's row loads (dy, x2).  Pass 2 reads both tensors again right after this launch: with the non-temporal policy
// of the other row streams (aux 2) pass 1 discourages exactly the lines pass 2 is about to ask for; -DVLPET_P1_AUX=0 = default policy.
#ifndef VLPET_P1_AUX
#define VLPET_P1_AUX 2
#endif
__device__ __forceinline__ void glds16_p1(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gmem_cv*)gsrc, (lmem_v*)lds_wave_base, 16, 0, VLPET_P1_AUX);
}

#ifndef VLPET_DZ2_AW1
#define VLPET_DZ2_AW1 2     // stages the weight / row requests run ahead at RT = 1 (LDS allows it); RT = 3: 1 / 1
#define VLPET_DZ2_AX1 2
#endif
#ifndef VLPET_DZ2_SPREAD
#define VLPET_DZ2_SPREAD 1
#endif
template <int RT> struct Dz2Geo {
    static constexpr int AW = RT == 1 ? VLPET_DZ2_AW1 : 1, AX = RT == 1 ? VLPET_DZ2_AX1 : 1;
    static constexpr int NWS = AW + 2, NXS = AX + 1;   // slots (a stage's weight image is used for two steps)
    static constexpr int PB = 64 * RT;                 // bytes of a weight row
    static constexpr int NPS = PB / 16;                // its 16-byte slots
    static constexpr int WT_B = 64 * PB;               // one chain's block of a stage
    static constexpr int XT_B = 128 * 128;             // one row tensor's tile
    static constexpr int WS_B = 2 * WT_B;              // weight slot [Wu | Wgu]: three slots (a stage's image is used for two steps)
    static constexpr int XS_B = 2 * XT_B;              // row slot [dy | x2]: two slots
    static constexpr int X_OFF = NWS * WS_B;
    static constexpr int BIAS_OFF = X_OFF + NXS * XS_B;
    static size_t bytes(int d) { return (size_t)BIAS_OFF + (size_t)2 * d * 4; }
};

// YF (round 5, "from the output"): the caller still has the forward's result y = gs * h * g (PetBwdArgs::y).  Then
//     dq = dh * h * (1 - g) = gs * dy * g * h * (1 - g) = dy * y * (1 - g),       dh = gs * dy * g
// need the gate's up projection only: the adapter chain's (half of this pass's projection MFMAs and weight-fragment reads, its z
// fragments, its bias) drops out, and the row ring carries y in the place of x2 -- same traffic.  y is the forward's bf16 rounding of
// gs * h * g: a 2^-9 relative change of dq before dq itself is rounded to bf16 for the matrix cores.
template <int RT, bool ADD, bool YF>
__global__ __launch_bounds__(512, 2) void k1_dz2_kernel(PetBwdArgs a) {
    static_assert(!(ADD && YF), "the additive gate's backward needs neither h nor y");
    constexpr bool NEED_A = !ADD && !YF;               // the adapter chain's up projection feeds h only
    using GEO = Dz2Geo<RT>;
    constexpr int KT = 2 * RT;
    constexpr int PB = GEO::PB, NPS = GEO::NPS, WT_B = GEO::WT_B, XT_B = GEO::XT_B, WS_B = GEO::WS_B, XS_B = GEO::XS_B;
    constexpr int X_OFF = GEO::X_OFF, AW = GEO::AW, AX = GEO::AX, NWS = GEO::NWS, NXS = GEO::NXS;
    constexpr bool SPREAD = VLPET_DZ2_SPREAD != 0;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.red_ctrl != nullptr && blockIdx.x == 0 && blockIdx.y == 0)        // pass 2's reduce-scatter state (cols_reduce.h): zeroed HERE, so
        for (int i = tid; i < a.red_words; i += 512) a.red_ctrl[i] = 0u;     // that no memset node sits between the passes
    const int rg = wave & 3, fh = wave >> 2;            // waves w and w + 4 share a SIMD: the two feature halves of a row group
    const int m = lane & 31, h = lane >> 5;
    const int d = a.d;
    // feature split (small M: PetBwdArgs::fsplit > 1): workgroup (x, y) walks only the stages of feature block y and leaves fp32
    // partial dz; k1_dz_reduce_kernel sums the blocks.  S0 .. S = this workgroup's stages.
    const int NFB = a.fsplit > 1 ? a.fsplit : 1;
    const int S0 = (int)blockIdx.y * ((d >> 6) / NFB), S = S0 + (d >> 6) / NFB;
    const int64_t ld2 = (int64_t)d * 2;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    const int64_t grow_raw = row0 + 32 * rg + m;
    const bool row_ok = grow_raw < a.M;
    const int64_t grow = row_ok ? grow_raw : a.M - 1;
    const PackGeom pg = pack_geom(RT, d, 1);

    bf16x8 zA[KT], zG[KT];                                              // (loaded below, after the first stage requests)
    // ---- the stage pieces (1 KiB each) of this wave: two pieces (8 rows each) of the dy tile and of the x2 tile, RT pieces of the
    // weight blocks.  The weight blocks are gathered from the "up" packs (fragments (stage, v, ks): slot (i, hh, j) =
    // W[f_of4(stage, v, i)][16 ks + 8 hh + j], tests/packing_spec.py) into natural row-major [f][c] order.
    uint32_t xoff[2], xdst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = wave + 8 * j, row = 8 * p + (lane >> 3);
        int64_t gr = row0 + row;
        if (gr >= a.M) gr = a.M - 1;
        xoff[j] = (uint32_t)((gr - row0) * ld2) + (uint32_t)(((lane & 7) ^ swz(row)) * 16);
        xdst[j] = (uint32_t)(p * 1024);
    }
    const uint8_t* wbase[RT]; uint32_t woff[RT], wdst[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
        const int q = wave + 8 * j, t = q / (4 * RT), piece = q % (4 * RT);
        const int sig = piece * 64 + lane, f = sig / NPS, sl = (sig % NPS) ^ gsw(f);
        const int i = 8 * ((f >> 2) & 3) + 4 * (f >> 5) + (f & 3), v = (f >> 4) & 1;
        wbase[j] = (t == 0 ? a.pk_a : a.pk_g) + pg.pack_bytes;
        woff[j] = (uint32_t)((v * KT + (sl >> 1)) * 1024 + ((sl & 1) * 32 + i) * 16);
        wdst[j] = (uint32_t)(t * WT_B + piece * 1024);
    }
    auto sbase = [](const uint8_t* p) {     // a wave-uniform pointer as a fresh scalar (keeps the per-lane part a 32-bit loop invariant)
        const uint64_t u = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const uint8_t*>(((uint64_t)hi << 32) | lo);
    };
    const uint8_t* dyp = reinterpret_cast<const uint8_t*>(a.dy) + row0 * ld2;
    const uint8_t* x2p = reinterpret_cast<const uint8_t*>(YF ? a.y : a.res) + row0 * ld2;
    auto issue_w = [&](int s) {
        uint8_t* st = smem + (size_t)(s % NWS) * WS_B;
#pragma unroll
        for (int j = 0; j < RT; ++j) glds16(sbase(wbase[j] + (int64_t)s * (4 * RT * 1024)) + woff[j], st + wdst[j]);
    };
    auto issue_x1 = [&](int s, int j) {
        uint8_t* st = smem + X_OFF + (size_t)(s % NXS) * XS_B;
        glds16_p1(sbase(dyp + s * 128) + xoff[j], st + xdst[j]);
        glds16_p1(sbase(x2p + s * 128) + xoff[j], st + XT_B + xdst[j]);
    };
    auto issue_x = [&](int s) { issue_x1(s, 0); issue_x1(s, 1); };

    // ---- per-lane LDS byte addresses (relative to the stage base)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    uint32_t a_wup[2], a_wtr[2], a_row[4];
    {
        const int g = gsw(m);
#pragma unroll
        for (int k = 0; k < 2; ++k)                     // A fragment of the up projection: row 32 fh + m, k-step 2 j + k (+ 64 j)
            a_wup[k] = (uint32_t)((32 * fh + m) * PB + (((2 * k + h) ^ g) * 16));
        const int g4 = lane >> 4, sl = lane & 15, hp = g4 >> 1;
        const int tslot = 2 * (g4 & 1) + ((sl & 3) >> 1), thalf = 8 * (sl & 1);
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {                // transpose reads: rows 4 hp + 8 hi + (sl >> 2) (+ 16 kappa), columns 16 (g4 & 1) + 4 (sl & 3) .. (+ 32 ct)
            const int r = 32 * fh + 4 * hp + 8 * hi + (sl >> 2);
            a_wtr[hi] = (uint32_t)(r * PB + ((tslot ^ gsw(r)) * 16) + thalf);
        }
        const int row = 32 * rg + m;
#pragma unroll
        for (int q = 0; q < 4; ++q)                     // the lane's features 32 fh + 8 q + 4 h .. + 3 of its row
            a_row[q] = (uint32_t)(row * 128 + (((4 * fh + q) ^ swz(row)) * 16) + 8 * h);
    }
    const uint32_t a_bias = lds0 + (uint32_t)(GEO::BIAS_OFF + (32 * fh + 4 * h) * 4);

    f32x16 dzA[RT], dzG[RT];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) { dzA[ct] = zero16(); dzG[ct] = zero16(); }
    const float s2 = a.s2, sd = a.sd, gs = a.gs;

#ifdef VLPET_DZ2_STAMPS
    uint64_t tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
#define DZ2_STAMP(k) { const uint64_t tn = wall_clock64(); tacc[k] += tn - tlast; tlast = tn; }
#else
#define DZ2_STAMP(k)
#endif
    // ---- the pieces of a step
    // up projection of one chain over this wave's 32 features, starting at the bias
    auto project_up = [&](uint32_t sb, auto TC, const bf16x8* z, f32x16& acc, int bias_off) {
        constexpr int T = decltype(TC)::value;
        u32x4 bb[4], wf[KT];
        sfor<4>([&](auto Q) { lds_read16<32 * Q.value>(bb[Q.value], a_bias + (uint32_t)bias_off); });
        sfor<KT>([&](auto K) { lds_read16<T * WT_B + 64 * (K.value >> 1)>(wf[K.value], sb + a_wup[K.value & 1]); });
        lgkm_fence(bb[0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q) lgkm_tie(bb[q]);
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) acc[4 * q + w2] = __uint_as_float(bb[q][w2]);
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) { lgkm_tie(wf[k]); acc = mfma32(as_bf(wf[k]), z[k], acc); }
    };
    // both up projections of stage s and the elementwise backward of the 16 values per lane (4 features at a time); dh, dq come
    // out as bf16 pairs in register order: the B fragments kappa = 0 (words 0-3) and 1 (words 4-7) of the contraction
    auto up_ew = [&](int s, uint32_t* bh, uint32_t* bq) {
        const uint32_t sb = lds0 + (uint32_t)((s % NWS) * WS_B);
        const uint32_t xb = lds0 + (uint32_t)(X_OFF + (s % NXS) * XS_B);
        f32x16 aA, aG;
        if constexpr (NEED_A) project_up(sb, std::integral_constant<int, 0>{}, zA, aA, (s * 64) * 4);
        if (SPREAD && s + AX < S) issue_x1(s + AX, 0);     // the row requests of the stage ahead are spread over the step: up front,
        project_up(sb, std::integral_constant<int, 1>{}, zG, aG, (d + s * 64) * 4);
        if (SPREAD && s + AX < S) issue_x1(s + AX, 1);     // all 56 pieces of a workgroup queue at the texture path at once (~1 k cycles of blocked issue)
#ifdef VLPET_DZ2_STAMPS
        asm volatile("s_nop 0" : "+v"(aG[15]));
#endif
        DZ2_STAMP(2)
        u32x2 dyv[4], x2v[4];
        sfor<4>([&](auto Q) {
            lds_read8<0>(dyv[Q.value], xb + a_row[Q.value]);
            lds_read8<XT_B>(x2v[Q.value], xb + a_row[Q.value]);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dyv[0]), "+v"(x2v[0]), "+v"(dyv[1]), "+v"(x2v[1]), "+v"(dyv[2]), "+v"(x2v[2]), "+v"(dyv[3]), "+v"(x2v[3]) :: "memory");
        sfor<4>([&](auto Q) {
            constexpr int q = Q.value;
            asm volatile("" : "+v"(dyv[q]), "+v"(x2v[q]), "+v"(aG[4 * q]), "+v"(aG[4 * q + 1]), "+v"(aG[4 * q + 2]), "+v"(aG[4 * q + 3]) :: "memory");
            float dh[4], dq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * q + j;
                const float gt = sigm(aG[e]);
                const float dyr = (j & 1) ? bf_hi(dyv[q][j >> 1]) : bf_lo(dyv[q][j >> 1]);
                const float dy_ = gs * dyr;
                if constexpr (ADD) {
                    dh[j] = dy_;
                    dq[j] = dy_ * gt * (1.0f - gt);
                } else if constexpr (YF) {      // (the row ring's second tile is y)
                    const float yv = (j & 1) ? bf_hi(x2v[q][j >> 1]) : bf_lo(x2v[q][j >> 1]);
                    dh[j] = dy_ * gt;
                    dq[j] = dyr * yv * (1.0f - gt);
                } else {
                    const float hv = s2 * ((j & 1) ? bf_hi(x2v[q][j >> 1]) : bf_lo(x2v[q][j >> 1])) + sd * aA[e];
                    dh[j] = dy_ * gt;
                    dq[j] = dh[j] * hv * (1.0f - gt);
                }
            }
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            const bf16x4 th = {(__bf16)dh[0], (__bf16)dh[1], (__bf16)dh[2], (__bf16)dh[3]};
            const bf16x4 tq = {(__bf16)dq[0], (__bf16)dq[1], (__bf16)dq[2], (__bf16)dq[3]};
            const u32x2 uh = __builtin_bit_cast(u32x2, th), uq = __builtin_bit_cast(u32x2, tq);
            bh[2 * q] = uh[0]; bh[2 * q + 1] = uh[1];
            bq[2 * q] = uq[0]; bq[2 * q + 1] = uq[1];
        });
#ifdef VLPET_DZ2_STAMPS
        asm volatile("s_nop 0" : "+v"(bq[7]), "+v"(bh[7]));
#endif
        DZ2_STAMP(3)
    };
    // contraction over this wave's 32 features of stage s: dz[ct] += W^T (transpose reads of the same image) . dh / dq
    auto contract1 = [&](uint32_t sb, auto TC, const uint32_t* bw, f32x16* dz) {
        constexpr int T = decltype(TC)::value;
        TrOp ap[2][RT];
        sfor<2>([&](auto KP) {
            sfor<RT>([&](auto CT) {
                tr_read2<T * WT_B + KP.value * 16 * PB + 64 * CT.value>(ap[KP.value][CT.value], sb + a_wtr[0], sb + a_wtr[1]);
            });
        });
        tr_fence(ap[0][0]);
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            const u32x4 bv = {bw[4 * kp], bw[4 * kp + 1], bw[4 * kp + 2], bw[4 * kp + 3]};
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) {
                if (kp || ct) tr_tie(ap[kp][ct]);
                dz[ct] = mfma32(tr_val(ap[kp][ct]), as_bf(bv), dz[ct]);
            }
        }
    };
    auto contract = [&](int s, const uint32_t* bh, const uint32_t* bq) {
        const uint32_t sb = lds0 + (uint32_t)((s % NWS) * WS_B);
        contract1(sb, std::integral_constant<int, 0>{}, bh, dzA);
        contract1(sb, std::integral_constant<int, 1>{}, bq, dzG);
#ifdef VLPET_DZ2_STAMPS
        asm volatile("s_nop 0" : "+v"(dzG[RT - 1][15]), "+v"(dzA[RT - 1][15]));
#endif
        DZ2_STAMP(4)
    };
    // the frame of a step: everything this wave requested for stage s has landed (nothing younger is in flight), barrier (stage s is
    // complete for every wave; the weight slot of stage s - 2 and the row slot of stage s - 1 are free), request stage s + 1
    auto step_top = [&](int s) {
        // requests of this wave younger than the last one stage s needs (program order per step: W(t + AW), X(t + AX))
        constexpr int AM = AW < AX ? AW : AX;
        int n = (AW <= AX && AW < AX && s - AW + AX < S) ? 4 : 0;
#pragma unroll
        for (int t = 1; t < AM; ++t) n += (s - t + AW < S ? RT : 0) + (s - t + AX < S ? 4 : 0);
        DZ2_STAMP(5)
        vm_wait(n);
        DZ2_STAMP(0)
        __builtin_amdgcn_s_barrier();
        DZ2_STAMP(1)
        if (s + AW < S) issue_w(s + AW);
        if (!SPREAD && s + AX < S) issue_x(s + AX);
    };

    // The first stages' requests go out BEFORE this wave fetches its own z rows and the biases (round 5): the two used to be serial -- z and
    // biases, wait, then the stage requests and their full memory latency again before the first MFMA (~1.5 us of a launch that is a
    // 17-us chain at the per-rank sizes).  One wait covers both; the counted waits of the loop see nothing older than the stage requests.
#pragma unroll
    for (int t = 0; t < (AW > AX ? AW : AX); ++t) {        // (same order as in the loop: W(t' + AW), X(t' + AX) for t' = t - max .. )
        if (t < AW && S0 + t < S) issue_w(S0 + t);
        if (t < AX && S0 + t < S) issue_x(S0 + t);
    }
    // up-side biases -> LDS (fp32): [bu_a (d) | bu_g (d)]
    {
        float* sbias = reinterpret_cast<float*>(smem + GEO::BIAS_OFF);
        const float* ba = reinterpret_cast<const float*>(a.pk_a + pg.bias_off) + 32 * RT;
        const float* bg = reinterpret_cast<const float*>(a.pk_g + pg.bias_off) + 32 * RT;
        for (int i = tid; i < d; i += 512) { sbias[i] = ba[i]; sbias[d + i] = bg[i]; }
    }
    // the saved bottleneck activations of this lane's row: B fragments of the up projections (k-slot (h, j) of k-step ks = c 16 ks + 8 h + j)
    {
        const uint8_t* sv = reinterpret_cast<const uint8_t*>(a.saved);
        const __bf16* sa = reinterpret_cast<const __bf16*>(sv) + grow * (int64_t)(32 * RT) + 8 * h;
        const __bf16* sg = reinterpret_cast<const __bf16*>(sv + 2 * a.saved_stride) + grow * (int64_t)(32 * RT) + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            if constexpr (NEED_A) zA[ks] = *reinterpret_cast<const bf16x8*>(sa + 16 * ks);
            zG[ks] = *reinterpret_cast<const bf16x8*>(sg + 16 * ks);
        }
    }

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // z in registers, biases in LDS (and the first stages landed)
    // (hipcc does not read the wait above: without a use HERE it guards the first MFMA that touches z with its own vmcnt(0), after the
    //  first step has requested the stage ahead -- one full memory latency in front of the first product of every launch)
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) {
        if constexpr (NEED_A) asm volatile("" : "+v"(zA[ks]));
        asm volatile("" : "+v"(zG[ks]));
    }
    // The two waves of a SIMD (the feature halves of a row group) run the same three phases -- up projections (LDS + matrix
    // cores), elementwise (VALU, 1.1 k cycles), contraction (LDS + matrix cores) -- and in lockstep they would queue for the same
    // unit in every phase (measured: a step took the SUM of the three, 5.8 k cycles).  So the second one runs its contraction a
    // step late, at the START of the next step: while it is in the matrix-core phases the first one is in the elementwise phase
    // and vice versa.  The weight image of a stage therefore lives for two steps (three weight slots).
    if (fh == 0) {
#pragma unroll 1
        for (int s = S0; s < S; ++s) {
            step_top(s);
            uint32_t bh[8], bq[8];
            up_ew(s, bh, bq);
            contract(s, bh, bq);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (every LDS access of this step is complete at the next barrier)
        }
    } else {
        uint32_t bh[8], bq[8];
        step_top(S0);
        up_ew(S0, bh, bq);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int s = S0 + 1; s < S; ++s) {
            step_top(s);
            contract(s - 1, bh, bq);
            up_ew(s, bh, bq);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        contract(S - 1, bh, bq);
    }

#ifdef VLPET_DZ2_STAMPS
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && (wave == 0 || wave == 4))
        printf("dz2 stamps blk %d wave %d (10 ns units): vmwait %llu barrier %llu up %llu ew %llu contract %llu other %llu\n", (int)blockIdx.x, wave,
               (unsigned long long)tacc[0], (unsigned long long)tacc[1], (unsigned long long)tacc[2], (unsigned long long)tacc[3],
               (unsigned long long)tacc[4], (unsigned long long)tacc[5]);
#endif
    // ---- the two feature halves of a row group meet: fh = 0 finishes chain A, fh = 1 chain G
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 gpv[RT][4];                                                    // act'(pre) of the chain this wave finishes (requested now, used last)
    {
        const uint8_t* sv = reinterpret_cast<const uint8_t*>(a.saved) + (fh == 0 ? 1 : 3) * a.saved_stride;
        const __bf16* gp = reinterpret_cast<const __bf16*>(sv) + grow * (int64_t)(32 * RT) + 4 * h;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) gpv[ct][q] = *reinterpret_cast<const bf16x4*>(gp + 32 * ct + 8 * q);
    }
    __syncthreads();                                                      // the stages are free
    {
        float* xch = reinterpret_cast<float*>(smem) + (size_t)(4 * fh + rg) * (RT * 16 * 64);   // what this wave hands over
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 t;
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = fh == 0 ? dzG[ct][4 * q + j] : dzA[ct][4 * q + j];
                *reinterpret_cast<f32x4*>(xch + (size_t)((ct * 4 + q) * 64 + lane) * 4) = t;
            }
    }
    __syncthreads();
    if (NFB > 1) {      // feature split: this block's fp32 sums of the chain this wave finishes -> dz_part[fb][row][chain][32 RT]
        const float* got = reinterpret_cast<const float*>(smem) + (size_t)(4 * (1 - fh) + rg) * (RT * 16 * 64);
        float* out = a.dz_part + (((int64_t)blockIdx.y * a.M + grow) * 2 + fh) * (32 * RT) + 4 * h;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(got + (size_t)((ct * 4 + q) * 64 + lane) * 4);
                f32x4 r4;
#pragma unroll
                for (int j = 0; j < 4; ++j) r4[j] = (fh == 0 ? dzA[ct][4 * q + j] : dzG[ct][4 * q + j]) + o[j];
                if (row_ok) *reinterpret_cast<f32x4*>(out + 32 * ct + 8 * q) = r4;
            }
    } else {
        const float* got = reinterpret_cast<const float*>(smem) + (size_t)(4 * (1 - fh) + rg) * (RT * 16 * 64);
        __bf16* out = reinterpret_cast<__bf16*>(fh == 0 ? a.dp_a : a.dp_g) + grow * (int64_t)(32 * RT) + 4 * h;
        const float sc = fh == 0 ? sd : 1.0f;           // dz_a = sd * Wu^T dh: the delta scale once, here
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {               // columns 32 ct + 8 q + 4 h .. + 3
                const f32x4 o = *reinterpret_cast<const f32x4*>(got + (size_t)((ct * 4 + q) * 64 + lane) * 4);
                bf16x4 r4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float mine = fh == 0 ? dzA[ct][4 * q + j] : dzG[ct][4 * q + j];
                    r4[j] = (__bf16)(sc * (mine + o[j]) * (float)gpv[ct][q][j]);
                }
                if (row_ok) *reinterpret_cast<bf16x4*>(out + 32 * ct + 8 * q) = r4;
            }
    }
}

// sums the feature blocks of the split form: dpre[row][c] = (chain a: sd, chain g: 1) * act'(pre)[row][c] * sum_fb dz_part[fb][row][chain][c]
__global__ __launch_bounds__(256) void k1_dz_reduce_kernel(PetBwdArgs a, int PR) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    const int64_t n4 = a.M * 2 * (PR / 4);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % (PR / 4)), chain = (int)((i / (PR / 4)) & 1);
        const int64_t row = i / (2 * (PR / 4));
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int fb = 0; fb < a.fsplit; ++fb)
            s += *reinterpret_cast<const f32x4*>(a.dz_part + (((int64_t)fb * a.M + row) * 2 + chain) * PR + 4 * c4);
        const uint8_t* sv = reinterpret_cast<const uint8_t*>(a.saved) + (chain == 0 ? 1 : 3) * a.saved_stride;
        const bf16x4 gp = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(sv) + row * PR + 4 * c4);
        const float sc = chain == 0 ? a.sd : 1.0f;
        bf16x4 r4;
#pragma unroll
        for (int j = 0; j < 4; ++j) r4[j] = (__bf16)(sc * s[j] * (float)gp[j]);
        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(chain == 0 ? a.dp_a : a.dp_g) + row * PR + 4 * c4) = r4;
    }
}

hipError_t launch_k1_dz_reduce(const PetBwdArgs& a, int PR, hipStream_t stream) {
    const int64_t n4 = a.M * 2 * (PR / 4);
    const unsigned rb = (unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(k1_dz_reduce_kernel, dim3(rb), dim3(256), 0, stream, a, PR);
    return hipGetLastError();
}

// feature blocks of pass 1 by shape: the kernel is a chain of d / 64 stages whatever its rows, and below 8,192 rows fewer than 64 of
// the 256 CUs have a workgroup -- four feature blocks give each a quarter of the chain (profiles/r04_k1bench_small_m.txt)
int k1_dz2_feature_blocks(int64_t M, int d) {
    if (const int f = vlpet_tuning().dz2_fsplit; f >= 1) return ((d >> 6) % f == 0) ? f : 1;
    return (M <= 8192 && (d >> 6) % 4 == 0) ? 4 : 1;
}

bool k1_dz2_applies(const PetBwdArgs& a, int io_fp32) {
    if (io_fp32 || !(a.flags & PET_GATE) || a.saved == nullptr || drop_active(a.drop) || a.d % 64 != 0 || a.d < 64) return false;
    // (the up-side biases of every feature sit in LDS next to the rings: very wide models go to pet_gate_dz_kernel)
    if (a.RT == 1) return Dz2Geo<1>::bytes(a.d) <= (size_t)160 * 1024;
    if (a.RT == 3) return Dz2Geo<3>::bytes(a.d) <= (size_t)160 * 1024;
    return false;
}

template <int RT, bool ADD, bool YF>
static hipError_t launch_dz2_form(const PetBwdArgs& a, hipStream_t stream) {
    using GEO = Dz2Geo<RT>;
    const size_t lds = GEO::bytes(a.d);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k1_dz2_kernel<RT, ADD, YF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)((a.M + 127) / 128);
    const unsigned nfb = a.fsplit > 1 ? (unsigned)a.fsplit : 1u;
    hipLaunchKernelGGL((k1_dz2_kernel<RT, ADD, YF>), dim3(blocks, nfb), dim3(512), lds, stream, a);
    if (nfb > 1) return launch_k1_dz_reduce(a, 32 * RT, stream);
    return hipGetLastError();
}

template <int RT>
static hipError_t launch_dz2_rt(const PetBwdArgs& a, hipStream_t stream) {
    if (a.flags & PET_GATE_ADD) return launch_dz2_form<RT, true, false>(a, stream);
    if (a.y != nullptr) return launch_dz2_form<RT, false, true>(a, stream);      // from the forward's output (see the kernel)
    return launch_dz2_form<RT, false, false>(a, stream);
}

hipError_t launch_k1_dz2(const PetBwdArgs& a, hipStream_t stream) {
    return a.RT == 1 ? launch_dz2_rt<1>(a, stream) : launch_dz2_rt<3>(a, stream);
}
