// Pass 1 of the two-pass K1 backward at SIX tiles (r = 192, the T5 script; r = 128): the feature-split form of pet_dz2.hip with one
// wave per row group and the whole register file (round 4).
// Autograd of my_transformers/modeling_bart.py:1147-1155,1195-1209 (T5: my_transformers/modeling_t5.py:366-390, 782-806) -- same
// contract as k1_dz2_kernel / pet_gate_dz_kernel: reads dy, x2 (+ the saved z, gelu'), recomputes both up projections, dh, dq tile by
// tile, contracts dz_a = sd * Wu^T dh, dz_g = Wgu^T dq in registers, writes only dpre_a, dpre_g [M, 192].
//
// Why a third form.  k1_dz2_kernel gives a 32-row group to two waves (the feature halves of a stage) with 256 registers each; at six
// tiles a wave would hold z (96 registers) and dz (192) -- so r = 192 stayed on the chain-split pet_gate_dz_kernel: 64-row
// workgroups, three barriers per stage, two weight images per chain, 90 us at 18,250 rows (two rounds of workgroups).  Here a
// workgroup is FOUR waves, one per SIMD, each with the full 512-entry register file (accumulators in the AGPR half): a wave owns a
// 32-row group, keeps z_a, z_g (96) and dz_a, dz_g (192) for the whole launch and walks the 24 half-stages of 32 features alone --
// both up projections of the 32 features (24 MFMAs), the elementwise backward of its 16 values per lane, the contraction (24 MFMAs)
// with the A operand taken from the SAME LDS image of Wu / Wgu by transpose reads.  No exchange between waves at all; one barrier
// per half-stage (the ring hand-over).  128-row workgroups: one round up to 32,768 rows.
// LDS: weight ring 2 x [Wu | Wgu blocks of 32 features x 384 B] (48 KiB) + row ring 2 x [dy | x2 tiles of 128 rows x 128 B] (64 KiB,
// refilled every second half-stage) + the up biases (6 KiB) = 118 KiB.
#include "cols_common.h"

// Cache policy of pass 1's row loads (dy, x2).  Pass 2 reads both tensors again right after this launch: with the non-temporal policy
// of the other row streams (aux 2) pass 1 discourages exactly the lines pass 2 is about to ask for; -DVLPET_P1_AUX=0 = default policy.
#ifndef VLPET_P1_AUX
#define VLPET_P1_AUX 2
#endif
__device__ __forceinline__ void glds16_p1(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gmem_cv*)gsrc, (lmem_v*)lds_wave_base, 16, 0, VLPET_P1_AUX);
}

// A/B switches of the round-5 changes (tools/gpu/r5_q.sh builds the variants): first requests after the z / bias wait (rounds 4 order),
// no use of the z registers at the wait, the epilogue's act' loads one at a time
#ifndef VLPET_DZ6_LATE_ISSUE
#define VLPET_DZ6_LATE_ISSUE 0
#endif
#ifndef VLPET_DZ6_PIN_Z
#define VLPET_DZ6_PIN_Z 1
#endif
#ifndef VLPET_DZ6_EPI_SERIAL
#define VLPET_DZ6_EPI_SERIAL 3      // 3 = act' rows through LDS (below); 1 = the rounds 4 form, one load at a time; 0 = all 48 act' loads first (measured SLOWER: 69.9 vs 62.1 us at 18,250 rows -- 96 more live registers cost the loop more
#endif                              //     than the batched loads save; profiles/r05_k1bench_dz6_variants.txt); 2 = no act' loads at all (diagnosis: a lower bound)
template <int RT> struct Dz6Geo {
    static constexpr int PB = 64 * RT;                 // bytes of a weight row (one feature, all bottleneck columns)
    static constexpr int NPS = PB / 16;
    static constexpr int WT_B = 32 * PB;               // one chain's block of a half-stage (32 features)
    static constexpr int WS_B = 2 * WT_B;              // weight slot [Wu | Wgu]
    static constexpr int XT_B = 128 * 128;             // one row tensor's tile of a stage (64 features)
    static constexpr int XS_B = 2 * XT_B;              // row slot [dy | x2]
    static constexpr int X_OFF = 2 * WS_B;
    static constexpr int BIAS_OFF = X_OFF + 2 * XS_B;
    static constexpr int NWP = 2 * WT_B / 1024 / 4;    // weight pieces (1 KiB) per wave and half-stage
    static size_t bytes(int d) { return (size_t)BIAS_OFF + (size_t)2 * d * 4; }
};

// YF: from the forward's output y instead of x2 (k1_dz2_kernel's note): dq = dy * y * (1 - g) -- no adapter-chain projection (12 of the
// 48 MFMAs and a third of the LDS fragment reads of a half-stage), no z_a fragments (48 registers).
template <int RT, bool ADD, bool YF>
__global__ __launch_bounds__(256) void k1_dz6_kernel(PetBwdArgs a) {
    static_assert(!(ADD && YF), "the additive gate's backward needs neither h nor y");
    constexpr bool NEED_A = !ADD && !YF;
    using GEO = Dz6Geo<RT>;
    constexpr int KT = 2 * RT;
    constexpr int PB = GEO::PB, NPS = GEO::NPS, WT_B = GEO::WT_B, WS_B = GEO::WS_B, XT_B = GEO::XT_B, XS_B = GEO::XS_B;
    constexpr int X_OFF = GEO::X_OFF, NWP = GEO::NWP;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave;
    const int m = lane & 31, h = lane >> 5;
    const int d = a.d;
    // feature split (small M, PetBwdArgs::fsplit > 1, as in pet_dz2.hip): workgroup (x, y) walks only the stages [S0, S) of feature block
    // y and leaves fp32 partial sums in dz_part; k1_dz_reduce_kernel adds the blocks and applies act'(pre)
    const int NFB = a.fsplit > 1 ? a.fsplit : 1;
    const int S0 = (int)blockIdx.y * ((d >> 6) / NFB), S = S0 + (d >> 6) / NFB;
    const int64_t ld2 = (int64_t)d * 2;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    const int64_t grow_raw = row0 + 32 * rg + m;
    const bool row_ok = grow_raw < a.M;
    const int64_t grow = row_ok ? grow_raw : a.M - 1;
    const PackGeom pg = pack_geom(RT, d, 1);

    bf16x8 zA[KT], zG[KT];                                              // (loaded below, after the first stage requests)
    // ---- the pieces (1 KiB each) of this wave: four pieces (8 rows each) of the dy tile and of the x2 tile per STAGE, NWP pieces of
    // the weight blocks per HALF-stage.  The weight blocks are gathered from the "up" packs (fragments (stage, v, ks): slot (i, hh, j)
    // = W[f_of4(stage, v, i)][16 ks + 8 hh + j], tests/packing_spec.py) into natural row-major [f][c] order; feature f = 32 fh + fl of
    // the stage is pack lane i = 8 ((fl >> 2) & 3) + 4 fh + (fl & 3) of n-tile v = (fl >> 4) & 1.
    uint32_t xoff[4], xdst[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = wave + 4 * j, row = 8 * p + (lane >> 3);
        int64_t gr = row0 + row;
        if (gr >= a.M) gr = a.M - 1;
        xoff[j] = (uint32_t)((gr - row0) * ld2) + (uint32_t)(((lane & 7) ^ swz(row)) * 16);
        xdst[j] = (uint32_t)(p * 1024);
    }
    uint32_t woff[NWP], wdst[NWP]; int wten[NWP];
#pragma unroll
    for (int j = 0; j < NWP; ++j) {
        const int q = wave + 4 * j, t = q / (2 * RT), piece = q % (2 * RT);
        const int sig = piece * 64 + lane, fl = sig / NPS, sl = (sig % NPS) ^ gsw(fl);
        const int i = 8 * ((fl >> 2) & 3) + (fl & 3), v = (fl >> 4) & 1;
        wten[j] = t;
        woff[j] = (uint32_t)((v * KT + (sl >> 1)) * 1024 + ((sl & 1) * 32 + i) * 16);       // (+ 64 fh: the pack lane's 4 fh term)
        wdst[j] = (uint32_t)(t * WT_B + piece * 1024);
    }
    auto sbase = [](const uint8_t* p) {     // a wave-uniform pointer as a fresh scalar (keeps the per-lane part a 32-bit loop invariant)
        const uint64_t u = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const uint8_t*>(((uint64_t)hi << 32) | lo);
    };
    const uint8_t* dyp = reinterpret_cast<const uint8_t*>(a.dy) + row0 * ld2;
    const uint8_t* x2p = reinterpret_cast<const uint8_t*>(YF ? a.y : a.res) + row0 * ld2;
    const uint8_t* wpa = a.pk_a + pg.pack_bytes;
    const uint8_t* wpg = a.pk_g + pg.pack_bytes;
    auto issue_w = [&](int ss) {            // half-stage ss = 2 s + fh
        uint8_t* st = smem + (size_t)(ss & 1) * WS_B;
        const int64_t so = (int64_t)(ss >> 1) * (4 * RT * 1024) + (ss & 1) * 64;
#pragma unroll
        for (int j = 0; j < NWP; ++j) glds16(sbase((wten[j] ? wpg : wpa) + so) + woff[j], st + wdst[j]);
    };
    auto issue_x = [&](int s) {
        uint8_t* st = smem + X_OFF + (size_t)(s & 1) * XS_B;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16_p1(sbase(dyp + s * 128) + xoff[j], st + xdst[j]);
            glds16_p1(sbase(x2p + s * 128) + xoff[j], st + XT_B + xdst[j]);
        }
    };

    // ---- per-lane LDS byte addresses (relative to the slot bases)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    uint32_t a_wup[2], a_wtr[2], a_row[4];
    {
        const int g = gsw(m);
#pragma unroll
        for (int k = 0; k < 2; ++k)                     // A fragment of the up projection: row m of the block, k-step 2 j + k (+ 64 j)
            a_wup[k] = (uint32_t)(m * PB + (((2 * k + h) ^ g) * 16));
        const int g4 = lane >> 4, sl = lane & 15, hp = g4 >> 1;
        const int tslot = 2 * (g4 & 1) + ((sl & 3) >> 1), thalf = 8 * (sl & 1);
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {                // transpose reads: rows 4 hp + 8 hi + (sl >> 2) (+ 16 kappa), columns 16 (g4 & 1) + 4 (sl & 3) .. (+ 32 ct)
            const int r = 4 * hp + 8 * hi + (sl >> 2);
            a_wtr[hi] = (uint32_t)(r * PB + ((tslot ^ gsw(r)) * 16) + thalf);
        }
        const int row = 32 * rg + m;
#pragma unroll
        for (int q = 0; q < 4; ++q)                     // the lane's features 32 fh + 8 q + 4 h .. + 3 of its row: slot (4 fh + q) ^ swz(row)
            a_row[q] = (uint32_t)(row * 128 + ((q ^ swz(row)) * 16) + 8 * h);       //   = (q ^ swz(row)) ^ 4 fh: the half flips address bit 6
    }
    const uint32_t a_bias = lds0 + (uint32_t)(GEO::BIAS_OFF + (4 * h) * 4);

    f32x16 dzA[RT], dzG[RT];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) { dzA[ct] = zero16(); dzG[ct] = zero16(); }
    const float s2 = a.s2, sd = a.sd, gs = a.gs;

    // up projection of one chain over the 32 features of the half-stage, starting at the bias
    auto project_up = [&](uint32_t sb, auto TC, const bf16x8* z, f32x16& acc, int bias_off) {
        constexpr int T = decltype(TC)::value;
        constexpr int GRP = KT / 2;                       // A fragments per LDS batch (two batches: the VGPR half of the file is the tight one)
        u32x4 bb[4], wf[GRP];
        sfor<4>([&](auto Q) { lds_read16<32 * Q.value>(bb[Q.value], a_bias + (uint32_t)bias_off); });
        sfor<GRP>([&](auto K) { lds_read16<T * WT_B + 64 * (K.value >> 1)>(wf[K.value], sb + a_wup[K.value & 1]); });
        lgkm_fence(bb[0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q) lgkm_tie(bb[q]);
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) acc[4 * q + w2] = __uint_as_float(bb[q][w2]);
        }
#pragma unroll
        for (int k = 0; k < GRP; ++k) { lgkm_tie(wf[k]); acc = mfma32(as_bf(wf[k]), z[k], acc); }
        u32x4 wf2[GRP];
        sfor<GRP>([&](auto K) { constexpr int ks = GRP + K.value; lds_read16<T * WT_B + 64 * (ks >> 1)>(wf2[K.value], sb + a_wup[ks & 1]); });
        lgkm_fence(wf2[0]);
#pragma unroll
        for (int k = 0; k < GRP; ++k) { if (k) lgkm_tie(wf2[k]); acc = mfma32(as_bf(wf2[k]), z[GRP + k], acc); }
    };
    // contraction over the 32 features of the half-stage: dz[ct] += W^T (transpose reads of the same image) . dh / dq
    auto contract1 = [&](uint32_t sb, auto TC, const uint32_t* bw, f32x16* dz) {
        constexpr int T = decltype(TC)::value;
        sfor<2>([&](auto KP) {
            constexpr int kp = KP.value;
            TrOp ap[RT];
            sfor<RT>([&](auto CT) { tr_read2<T * WT_B + kp * 16 * PB + 64 * CT.value>(ap[CT.value], sb + a_wtr[0], sb + a_wtr[1]); });
            tr_fence(ap[0]);
            const u32x4 bv = {bw[4 * kp], bw[4 * kp + 1], bw[4 * kp + 2], bw[4 * kp + 3]};
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) {
                if (ct) tr_tie(ap[ct]);
                dz[ct] = mfma32(tr_val(ap[ct]), as_bf(bv), dz[ct]);
            }
        });
    };

    // the first half-stage's requests go out before this wave fetches its own z rows and the biases (see k1_dz2_kernel)
#if !VLPET_DZ6_LATE_ISSUE
    issue_w(2 * S0);
    issue_x(S0);
#endif
    // up-side biases -> LDS (fp32): [bu_a (d) | bu_g (d)]
    {
        float* sbias = reinterpret_cast<float*>(smem + GEO::BIAS_OFF);
        const float* ba = reinterpret_cast<const float*>(a.pk_a + pg.bias_off) + 32 * RT;
        const float* bg = reinterpret_cast<const float*>(a.pk_g + pg.bias_off) + 32 * RT;
        for (int i = tid; i < d; i += 256) { sbias[i] = ba[i]; sbias[d + i] = bg[i]; }
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

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // z in registers, biases in LDS (and the first half-stage landed)
#if VLPET_DZ6_PIN_Z
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) {                  // (a use here: hipcc's own guard of the z loads lands on the wait above, see k1_dz2_kernel)
        if constexpr (NEED_A) asm volatile("" : "+v"(zA[ks]));
        asm volatile("" : "+v"(zG[ks]));
    }
#endif
#if VLPET_DZ6_LATE_ISSUE
    issue_w(2 * S0);
    issue_x(S0);
#endif

    // request order per step (what the counted waits rely on): half-stage (s, 0): W(2s + 1), X(s + 1); half-stage (s, 1): W(2s + 2)
#pragma unroll 1
    for (int ss = 2 * S0; ss < 2 * S; ++ss) {
        const int s = ss >> 1, fh = ss & 1;
        // everything this wave requested for half-stage ss has landed: younger than W(ss) are only the row pieces of stage s + 1
        // (requested in half-stage (s, 0), after W(2s + 1))
        vm_wait(fh == 1 && s + 1 < S ? 8 : 0);
        __builtin_amdgcn_s_barrier();                                     // half-stage ss is complete for every wave; the slots of ss - 1 are free
        if (ss + 1 < 2 * S) issue_w(ss + 1);
        if (fh == 0 && s + 1 < S) issue_x(s + 1);
        const uint32_t sb = lds0 + (uint32_t)((ss & 1) * WS_B);
        const uint32_t xb = lds0 + (uint32_t)(X_OFF + (s & 1) * XS_B);
        f32x16 aA, aG;
        if constexpr (NEED_A) project_up(sb, std::integral_constant<int, 0>{}, zA, aA, (s * 64 + 32 * fh) * 4);
        project_up(sb, std::integral_constant<int, 1>{}, zG, aG, (d + s * 64 + 32 * fh) * 4);
        uint32_t bh[8], bq[8];
        {
            u32x2 dyv[4], x2v[4];
            const uint32_t fbit = (uint32_t)fh << 6;
            sfor<4>([&](auto Q) {
                lds_read8<0>(dyv[Q.value], xb + (a_row[Q.value] ^ fbit));
                lds_read8<XT_B>(x2v[Q.value], xb + (a_row[Q.value] ^ fbit));
            });
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dyv[0]), "+v"(x2v[0]), "+v"(dyv[1]), "+v"(x2v[1]), "+v"(dyv[2]), "+v"(x2v[2]), "+v"(dyv[3]), "+v"(x2v[3]) :: "memory");
            sfor<4>([&](auto Q) {
                constexpr int q = Q.value;
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
        }
        contract1(sb, std::integral_constant<int, 0>{}, bh, dzA);
        contract1(sb, std::integral_constant<int, 1>{}, bq, dzG);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (every LDS access of this step is complete at the next barrier)
    }

    if (NFB > 1) {      // feature split: this block's fp32 sums of both chains -> dz_part[fb][row][chain][32 RT]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float* out = a.dz_part + (((int64_t)blockIdx.y * a.M + grow) * 2 + t) * (32 * RT) + 4 * h;
#pragma unroll
            for (int ct = 0; ct < RT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 r4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) r4[j] = t == 0 ? dzA[ct][4 * q + j] : dzG[ct][4 * q + j];
                    if (row_ok) *reinterpret_cast<f32x4*>(out + 32 * ct + 8 * q) = r4;
                }
        }
        return;
    }
    // ---- dpre = dz * act'(pre) of both chains (dz_a carries the delta scale once, here).  All 48 act' pieces of the lane are requested
    // before the first product (the z fragments are dead: 96 free registers): written as load / multiply / store per piece, hipcc kept
    // every load behind the previous store (gp and out may alias for all it knows) -- 48 dependent memory round trips at the end of
    // every workgroup (round 5: found in the ISA, 48 x "global_load_dwordx2; s_waitcnt vmcnt(0)").
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#if VLPET_DZ6_EPI_SERIAL == 3
    // The act' rows of a wave's 32 rows are 2 x 12 KiB of CONTIGUOUS memory ([M, 192] bf16): they come in by LDS-DMA -- no registers, one
    // memory latency for all of them -- into the (now free) stage rings, 24 KiB per wave, and are read back per piece.  16-byte chunk c of
    // row r sits at slot (c + r) mod 24 of the row (the DMA's LDS side is fixed per lane, so the rotation is applied to the SOURCE address):
    // rows are 96 dwords apart, unrotated every second lane of a read would hit the same banks.  The serial form costs 6 us of a 62 us
    // launch (profiles/r05_k1bench_dz6_variants.txt: 62.3 vs 56.0 us without the loads).
    {
        __builtin_amdgcn_s_barrier();                                     // every wave is done with the rings
        int lane_e = lane;                                                // (an opaque copy: the address arithmetic below must not be hoisted above the
        asm volatile("" : "+v"(lane_e));                                  //  24-half-stage loop, where every register counts)
        const int m_e = lane_e & 31, h_e = lane_e >> 5;
        uint8_t* area = smem + (size_t)rg * (2 * 32 * PB);
        const int64_t rb = row0 + 32 * rg;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint8_t* sv = reinterpret_cast<const uint8_t*>(a.saved) + (t == 0 ? 1 : 3) * a.saved_stride;
#pragma unroll
            for (int pc = 0; pc < 32 * PB / 1024; ++pc) {
                const int slot = pc * 64 + lane_e, r = slot / (PB / 16), cs = slot % (PB / 16);
                int c = cs - (r % (PB / 16));
                if (c < 0) c += PB / 16;
                int64_t gr = rb + r;
                if (gr >= a.M) gr = a.M - 1;
                glds16(sv + gr * PB + c * 16, area + (size_t)t * (32 * PB) + (size_t)pc * 1024);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (a wave reads back only what it requested itself)
        const uint32_t abase = lds0 + (uint32_t)(rg * (2 * 32 * PB)) + (uint32_t)(m_e * PB) + 8 * h_e;
        const int mrot = m_e % (PB / 16);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            __bf16* out = reinterpret_cast<__bf16*>(t == 0 ? a.dp_a : a.dp_g) + grow * (int64_t)(32 * RT) + 4 * h;
            const float sc = t == 0 ? sd : 1.0f;
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) {
                u32x2 gv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int cs = 4 * ct + q + mrot;
                    if (cs >= PB / 16) cs -= PB / 16;
                    lds_read8<0>(gv[q], abase + (uint32_t)(t * 32 * PB) + (uint32_t)(cs * 16));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gv[0]), "+v"(gv[1]), "+v"(gv[2]), "+v"(gv[3]) :: "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) {           // columns 32 ct + 8 q + 4 h .. + 3
                    const bf16x4 g1 = __builtin_bit_cast(bf16x4, gv[q]);
                    bf16x4 r4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) r4[j] = (__bf16)(sc * (t == 0 ? dzA[ct][4 * q + j] : dzG[ct][4 * q + j]) * (float)g1[j]);
                    if (row_ok) *reinterpret_cast<bf16x4*>(out + 32 * ct + 8 * q) = r4;
                }
            }
        }
    }
#elif VLPET_DZ6_EPI_SERIAL
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint8_t* sv = reinterpret_cast<const uint8_t*>(a.saved) + (t == 0 ? 1 : 3) * a.saved_stride;
        const __bf16* gp = reinterpret_cast<const __bf16*>(sv) + grow * (int64_t)(32 * RT) + 4 * h;
        __bf16* out = reinterpret_cast<__bf16*>(t == 0 ? a.dp_a : a.dp_g) + grow * (int64_t)(32 * RT) + 4 * h;
        const float sc = t == 0 ? sd : 1.0f;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#if VLPET_DZ6_EPI_SERIAL == 2
                const bf16x4 g1 = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};
#else
                const bf16x4 g1 = *reinterpret_cast<const bf16x4*>(gp + 32 * ct + 8 * q);
#endif
                bf16x4 r4;
#pragma unroll
                for (int j = 0; j < 4; ++j) r4[j] = (__bf16)(sc * (t == 0 ? dzA[ct][4 * q + j] : dzG[ct][4 * q + j]) * (float)g1[j]);
                if (row_ok) *reinterpret_cast<bf16x4*>(out + 32 * ct + 8 * q) = r4;
            }
    }
#else
    bf16x4 gpv[2][RT][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint8_t* sv = reinterpret_cast<const uint8_t*>(a.saved) + (t == 0 ? 1 : 3) * a.saved_stride;
        const __bf16* gp = reinterpret_cast<const __bf16*>(sv) + grow * (int64_t)(32 * RT) + 4 * h;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) gpv[t][ct][q] = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(gp + 32 * ct + 8 * q));
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        __bf16* out = reinterpret_cast<__bf16*>(t == 0 ? a.dp_a : a.dp_g) + grow * (int64_t)(32 * RT) + 4 * h;
        const float sc = t == 0 ? sd : 1.0f;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {               // columns 32 ct + 8 q + 4 h .. + 3
                bf16x4 r4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float mine = t == 0 ? dzA[ct][4 * q + j] : dzG[ct][4 * q + j];
                    r4[j] = (__bf16)(sc * mine * (float)gpv[t][ct][q][j]);
                }
                if (row_ok) *reinterpret_cast<bf16x4*>(out + 32 * ct + 8 * q) = r4;
            }
    }
#endif
}

// ---- chain-split form (round 5, second session): two waves per SIMD.
// (A chain-split eight-wave form of this pass -- two waves per SIMD, one per chain -- was built in round 5, parity-green and 11 % slower:
//  65.2 vs 58.3 us at 18,250 rows, profiles/r05_k1bench_dz6c_ab.txt; removed in round 6.)

int k1_dz6_feature_blocks(int64_t M, int d) {
    if (const int f = vlpet_tuning().dz2_fsplit; f >= 1) return ((d >> 6) % f == 0) ? f : 1;
    // six blocks while the grid still fits one round of the chip (26.1 / 29.1 / 31.1 us at 1,100 / 2,128 / 3,500 rows against 27.5 / 30.8 / 32.9 with
    // four; 47 row blocks x 6 at 6,000 rows is two rounds: 52 us -- profiles/r05_k1bench_fsplit_small_m.txt), four up to 8,192 rows
    if (M <= 8192 && (d >> 6) % 6 == 0 && ((M + 127) / 128) * 6 <= 256) return 6;
    return (M <= 8192 && (d >> 6) % 4 == 0) ? 4 : 1;
}

bool k1_dz6_applies(const PetBwdArgs& a, int io_fp32) {
    if (io_fp32 || !(a.flags & PET_GATE) || a.saved == nullptr || drop_active(a.drop) || a.d % 64 != 0 || a.d < 64) return false;
    // by shape (profiles/r04_k1bench_r192_dz6_ab.txt): the chain-split pet_gate_dz_kernel runs 64-row workgroups, ONE round of them up to
    // 16,384 rows in 42-44 us; above that it needs two rounds (87 us at 16,800 / 18,250 rows, 90 at 28,000) and this kernel -- a fixed chain
    // of 24 half-stages, 66-74 us whatever the rows up to 32,768 -- is the faster one (71 us at 18,250 rows, 74 at 28,000)
    // Second session: below 8,192 rows this kernel runs in four feature blocks (k1_dz6_feature_blocks; the per-rank shapes of the 8-GPU T5
    // config): a quarter of the 24-half-stage chain per workgroup + the reduce launch, against the 42-us chain of pet_gate_dz_kernel.
    if (a.M > 8192 && a.M <= 16384 && vlpet_tuning().dz6 != 2) return false;
    if (a.M <= 8192 && vlpet_tuning().dz6 != 2 && k1_dz6_feature_blocks(a.M, a.d) <= 1) return false;
    return a.RT == 6 && Dz6Geo<6>::bytes(a.d) <= (size_t)160 * 1024;
}

template <bool ADD, bool YF>
static hipError_t launch_dz6_form(const PetBwdArgs& a, hipStream_t stream) {
    using GEO = Dz6Geo<6>;
    const size_t lds = GEO::bytes(a.d);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k1_dz6_kernel<6, ADD, YF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)((a.M + 127) / 128);
    const unsigned nfb = a.fsplit > 1 ? (unsigned)a.fsplit : 1u;
    hipLaunchKernelGGL((k1_dz6_kernel<6, ADD, YF>), dim3(blocks, nfb), dim3(256), lds, stream, a);
    if (nfb > 1) return launch_k1_dz_reduce(a, 32 * 6, stream);
    return hipGetLastError();
}

hipError_t launch_k1_dz6(const PetBwdArgs& a, hipStream_t stream) {
    if (a.flags & PET_GATE_ADD) return launch_dz6_form<true, false>(a, stream);
    if (a.y != nullptr) return launch_dz6_form<false, true>(a, stream);
    return launch_dz6_form<false, false>(a, stream);
}
