// K1 backward, row-parallel part, chain-split form (gated K1 with the forward's saved activations only;
// everything else stays with pet_bwd.hip).  Autograd of my_transformers/modeling_bart.py:1147-1155,1195-1209.
//
// Why a second kernel: pet_bwd_kernel gives each 32-row group ONE wave that holds both projection chains
// (256 VGPRs + 240 AGPRs, one wave per SIMD) and its time is the latency chain of that wave: 107 us with a
// single workgroup on an idle chip.  Here the adapter chain and the gate chain of a row
// group are two waves on the same SIMD, as in pet_gate_fwd.hip: half the MFMAs, half the memory pieces and half
// of the elementwise stage per wave, two instruction streams per SIMD.
//
//   wave rg      (chain A):  z_a, gelu'_a (saved) ... dz_a ... dx2
//   wave rg + RG (chain G):  z_g, gelu'_g (saved) ... dz_g ... dx1
//
// Middle phase, feature block su (64 bf16 / 32 fp32 features), two stages:
//   stage a  both chains recompute their up projection a = bu + Wu z (12 MFMAs each for r = 96) and hand the
//            half the OTHER wave will need to the exchange buffer: the lane's 2*LW/2 features split in two
//            halves, chain A takes the first half of the elementwise work, chain G the second, and each needs
//            both aA (for h) and aG (for the gate) of its half.
//   stage b  elementwise backward of the own half (res, dy tiles -> dh, dq tiles in place), barrier, then each
//            chain stores its side product (A: dh, G: dq), reads the B fragments of its contraction from its
//            tile and accumulates dz += Wu^T (dh | dq).
// Last phase (S stages): A: dx2 = s2*dh + Wd^T dpre_a (dh rows re-read through the row ring), G: dx1 = Wgd^T dpre_g.
//
// LDS (151 KiB at RG = 4, as pet_bwd.hip): weight ring 2 x [A segment | G segment]; three row slots of
// [t0 | t1] tiles -- the middle phase needs rows only every other stage, so two slots give a prefetch distance
// of three stages and the third slot is the fp32 exchange buffer; the last phase uses all three as its row ring.
// Every wave issues the global_load_lds pieces of its own chain: its weight segment share and one row tensor
// (A: res, G: dy; last phase A: dh), and stores its own outputs.
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "pet32.h"

template <typename IO, int RT, int RG>
struct Bwd2Lds {
    static constexpr int NS = Geo4<IO>::NS;
    static constexpr int SEG_KB = 4 * RT;
    static constexpr int SEG_FR = SEG_KB / NS;
    static constexpr int W_B = SEG_KB * 1024 * 2;
    static constexpr int TILE_B = RG * 32 * 128;
    static constexpr int ROW_B = 2 * TILE_B;
    static constexpr int ROW_OFF = 2 * W_B;
    static constexpr int BIAS_OFF = ROW_OFF + 3 * ROW_B;
    static size_t bytes(int d) { return (size_t)BIAS_OFF + (size_t)2 * (32 * RT + d) * 4; }
};

template <bool B> struct Bool2 { static constexpr bool value = B; };

#ifdef VLPET_STAMPS
// diagnosis build only: cycle stamps of one wave of workgroup 0 (VLPET_DBG & 128: the chain-G wave), printed after the launch
__device__ unsigned long long g_b2_ts[64];
#define B2STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == stamp_tid) g_b2_ts[k] = __builtin_readcyclecounter(); } while (0)
#else
#define B2STAMP(k) do { } while (0)
#endif

// LR = low-rank visual projector form (LowRankVisualEmbedding, autograd of src/modeling_bart.py:278-295): d is the OUTPUT
// width; no residual term (res is never read), gate value sigmoid(.) + go, and no input gradients -- the features are data --
// so the kernel ends after dpre (the weight-gradient kernel then contracts dpre with the [M, d_in] features itself).
template <typename IO, int RT, int RG, bool LR = false>
__global__ __launch_bounds__(RG * 128) void pet_gate_bwd2_kernel(PetBwdArgs a) {
    using G = Geo4<IO>;
    using L = Bwd2Lds<IO, RT, RG>;
    constexpr int NS = G::NS;
    constexpr int KT = 2 * RT;
    constexpr int PW = L::SEG_KB / RG;              // weight pieces of the own segment per wave and stage
    static_assert(L::SEG_KB % RG == 0, "weight segment must split evenly over the row groups");
    static_assert(G::NV == 2 || G::NV == 1, "two or one n-tiles per stage");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chain = wave / RG, rg = wave % RG;
    const bool isA = chain == 0;
    const int m = lane & 31, h = lane >> 5;
    const int trow = 32 * rg + m;
    const int d = a.d;
    const int64_t row0_wave = (int64_t)blockIdx.x * (RG * 32) + rg * 32;
    const int64_t grow_raw = row0_wave + m;
    const bool row_ok = grow_raw < a.M;
    const int64_t grow = row_ok ? grow_raw : a.M - 1;
    const int S = d / G::FE;
    const PackGeom pg = pack_geom(RT, d, NS);
    const uint8_t* res = reinterpret_cast<const uint8_t*>(a.res);
    const uint8_t* dy = reinterpret_cast<const uint8_t*>(a.dy);
    uint8_t* DH = reinterpret_cast<uint8_t*>(a.dh);
    uint8_t* DQ = reinterpret_cast<uint8_t*>(a.dq);
    const uint8_t* DXIN = reinterpret_cast<const uint8_t*>(a.dxg_in);     // optional: dx1 += the sublayer tail's residual gradient

    auto slot_w = [&](int j) { return smem + (size_t)j * L::W_B + (isA ? 0 : L::SEG_KB * 1024); };   // own segment
    auto slot_t0 = [&](int j) { return smem + L::ROW_OFF + (size_t)j * L::ROW_B; };
    auto slot_t1 = [&](int j) { return smem + L::ROW_OFF + (size_t)j * L::ROW_B + L::TILE_B; };
    // exchange buffer (row slot 2 during the middle phase): [written by A | written by G], per row group 16 fp32 per lane
    constexpr int XH = G::LW / 2;                   // features per lane and half
    uint8_t* xbuf = smem + L::ROW_OFF + (size_t)2 * L::ROW_B;
    auto xslot = [&](int writer_chain) { return xbuf + (size_t)writer_chain * L::TILE_B + (size_t)rg * (XH * 256); };
    float* sb = reinterpret_cast<float*>(smem + L::BIAS_OFF);
    const int nb = 32 * RT + d;
    const int NST = LR ? 2 * S : 3 * S;             // stages: 2S middle (a, b alternating) + S last (none in the LR form)

    const RowLanes rl = row_lanes<IO>(row0_wave, a.M, d, rg, lane);
    const int lane16 = lane * 16;
    const uint32_t wv_off = (uint32_t)(rg * 1024 + lane16);

    // weight pieces of stage st: middle a -> pack 1 (up), middle b -> pack 2 (up_t), last -> pack 3 (down_t).  `both` = the
    // segments of BOTH chains (the loading wave of the phase, see below); otherwise the own segment (prologue)
    auto issue_w = [&](int st, bool both) {
        if (st >= NST) return 0;
        int pack, ss;
        if (st < 2 * S) { ss = st >> 1; pack = (st & 1) ? 2 : 1; } else { ss = st - 2 * S; pack = 3; }
        const int64_t so = (int64_t)pack * pg.pack_bytes + (int64_t)ss * L::SEG_KB * 1024;
        uint8_t* slot = smem + (size_t)(st & 1) * L::W_B;
        int n = 0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (!both && c != chain) continue;
            const uint8_t* src = (c == 0 ? a.pk_a : a.pk_g) + so;
            uint8_t* dst = slot + (size_t)c * L::SEG_KB * 1024 + rg * 1024;
#pragma unroll
            for (int j = 0; j < PW; ++j) glds16(src + (wv_off + j * RG * 1024), dst + j * RG * 1024);
            n += PW;
        }
        return n;
    };
    // Who issues the LDS-DMA of a phase.  Cycle stamps (profiles/r02_stamps_k1_bwd_rows.txt): in the middle phase the gate-chain
    // wave is the pole of every feature block (9.2 k cycles, of which 1.26 k issuing its seven pieces behind the adapter-chain
    // waves' in the memory pipe) while the adapter-chain wave idles 2.2 k cycles at the barriers; in the last phase it is the
    // other way round.  So the wave with the slack loads for both chains of its row group: chain A in the middle phase,
    // chain G in the last phase; the pole wave issues no loads at all (its data is covered by the loader's vmcnt + the barrier).
    // middle-phase rows of block su: res -> tile t0, dy -> tile t1 of slot su & 1
    auto issue_mid_rows = [&](int su, bool both) {
        if (su >= S) return 0;
        int n = 0;
        if constexpr (!LR) { if (both || isA) { glds_rows4(res, rl, su * 128, slot_t0(su & 1), rg); n += 4; } }
        if (both || !isA) { glds_rows4(dy, rl, su * 128, slot_t1(su & 1), rg); n += 4; }
        return n;
    };
    // last-phase rows of block su: dh -> tile t0 of slot su % 3; the incoming dx1 (optional) -> tile t1
    auto issue_last_rows = [&](int su, bool both) {
        if (su >= S) return 0;
        int n = 0;
        if (both || isA) { glds_rows4(DH, rl, su * 128, slot_t0(su % 3), rg); n += 4; }
        if ((both || !isA) && DXIN != nullptr) { glds_rows4(DXIN, rl, su * 128, slot_t1(su % 3), rg); n += 4; }
        return n;
    };

#ifdef VLPET_STAMPS
    const int stamp_tid = (a.flags & (1 << 20)) ? RG * 64 : 0;
#endif
    B2STAMP(0);
    // ---- prologue: weights of stage 0, rows of block 0, biases, saved activations
    issue_w(0, false);
    issue_mid_rows(0, false);
    copy_bias<RG * 128>(sb, reinterpret_cast<const float*>(a.pk_a + pg.bias_off), nb, tid);
    copy_bias<RG * 128>(sb + nb, reinterpret_cast<const float*>(a.pk_g + pg.bias_off), nb, tid);
    Frag<NS> z[KT];
    f32x16 gp[RT];                                  // act'(pre)
    {
        const uint8_t* sv = reinterpret_cast<const uint8_t*>(a.saved) + (isA ? 0 : 2) * a.saved_stride;
        const int64_t ro = grow * (int64_t)(32 * RT) + 8 * h;
        const IO* sz = reinterpret_cast<const IO*>(sv) + ro;
        const IO* sg = reinterpret_cast<const IO*>(sv + a.saved_stride) + ro;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) {
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                float v[8];
                load8_f32(sz + 32 * ct + 16 * sh, v);
                z[2 * ct + sh] = frag_from_f32<NS>(v);
                load8_f32(sg + 32 * ct + 16 * sh, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) gp[ct][8 * sh + j] = v[j];
            }
        }
    }
    __syncthreads();

    B2STAMP(1);
    // ---- middle phase
    const float* bu = sb + (isA ? 0 : nb) + 32 * RT + G::LW * h;
    const float s2 = a.s2, sd_ = a.sd, gs = a.gs;
    float gm = 1.f, go = 0.f;
    if constexpr (LR) { gm = a.gm; go = a.go; }
    const bool gate_add = (a.flags & PET_GATE_ADD) != 0;
    f32x16 dz[RT];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) dz[ct] = zero16();
    constexpr int EH = G::E4 / 2;                   // 8-feature groups per half (bf16: 2, fp32: 1)
    const int e0 = isA ? 0 : EH;                    // this chain's half of the elementwise work
    int st = 0;
    for (int su = 0; su < S; ++su) {
        // ================= stage a: recompute this chain's up projection, exchange the other half
        f32x16 au[G::NV];
        {
            if (su == 5) B2STAMP(8);
            int nrows = rl.n_inst;                  // chain G: only its dq stores of the previous block may still be in flight
            if (isA) { issue_w(st + 1, true); nrows = issue_mid_rows(su + 1, true); }
            if (su == 5) B2STAMP(9);
            const uint8_t* w = slot_w(st & 1);
#pragma unroll
            for (int v = 0; v < G::NV; ++v) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 tb = *reinterpret_cast<const f32x4*>(bu + su * G::FE + 16 * v + 4 * q);
                    au[v][4 * q] = tb[0]; au[v][4 * q + 1] = tb[1]; au[v][4 * q + 2] = tb[2]; au[v][4 * q + 3] = tb[3];
                }
            }
            Frag<NS> wf[G::NV * KT];
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
#pragma unroll
                for (int v = 0; v < G::NV; ++v) wf[v * KT + ks] = wfrag<NS>(w, v * KT + ks, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
#pragma unroll
                for (int v = 0; v < G::NV; ++v) au[v] = mfma_ns<NS>(wf[v * KT + ks], z[ks], au[v]);
            }
            if (su == 5) B2STAMP(10);
            // the half the partner works on: lane feature i <-> au[i >> 4][i & 15]; A keeps i < LW/2, G keeps i >= LW/2
            {
                uint8_t* xb = xslot(chain);
                const int ib = isA ? XH : 0;
#pragma unroll
                for (int q = 0; q < XH / 4; ++q) {
                    f32x4 t;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int i = ib + 4 * q + j; t[j] = au[(i >> 4) % G::NV][i & 15]; }
                    *reinterpret_cast<f32x4*>(xb + (size_t)q * 1024 + lane16) = t;
                }
            }
            if (su == 5) B2STAMP(11);
            wait_vm(nrows);
            if (su == 5) B2STAMP(12);
            __builtin_amdgcn_s_barrier();
            if (su == 5) B2STAMP(13);
            ++st;
        }
        // ================= stage b: elementwise backward of the own half, then the contraction over features
        {
            if (isA) issue_w(st + 1, true);
            const uint8_t* w = slot_w(st & 1);
            uint8_t* t0 = slot_t0(su & 1);
            uint8_t* t1 = slot_t1(su & 1);
            const uint8_t* xb = xslot(1 - chain);
            auto elementwise = [&](auto add_c) {
                constexpr bool ADD = decltype(add_c)::value;
#pragma unroll
                for (int ee = 0; ee < EH; ++ee) {
                    const int e = e0 + ee;
                    float r8[8], dy8[8], dh8[8], dq8[8], ox[8];
                    if constexpr (LR) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) r8[j] = 0.f;      // (t0 is only the staging tile of dh here)
                    } else {
                        tile_lane_vals8<IO>(t0, trow, h, e, r8);
                    }
                    tile_lane_vals8<IO>(t1, trow, h, e, dy8);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(xb + (size_t)(2 * ee + q) * 1024 + lane16);
                        ox[4 * q] = t[0]; ox[4 * q + 1] = t[1]; ox[4 * q + 2] = t[2]; ox[4 * q + 3] = t[3];
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = 8 * e + j;
                        const float own = au[(i >> 4) % G::NV][i & 15];
                        const float aAv = isA ? own : ox[j], aGv = isA ? ox[j] : own;
                        const float gt = sigmoid_f(aGv);
                        const float dyp = gs * dy8[j];
                        if constexpr (ADD) {
                            dh8[j] = dyp;
                            dq8[j] = dyp * gt * (1.0f - gt);
                        } else {
                            if constexpr (LR) {         // out = sd*aA * (gm*gt + go)
                                dh8[j] = dyp * (gm * gt + go);
                                dq8[j] = dyp * (sd_ * aAv) * gm * gt * (1.0f - gt);
                            } else {
                                const float hv = s2 * r8[j] + sd_ * aAv;
                                dh8[j] = dyp * gt;
                                dq8[j] = dh8[j] * hv * (1.0f - gt);
                            }
                        }
                    }
                    stage_lane_vals8<IO>(t0, trow, h, e, dh8);
                    stage_lane_vals8<IO>(t1, trow, h, e, dq8);
                }
            };
            if (su == 5) B2STAMP(14);
            if (gate_add) elementwise(Bool2<true>{}); else elementwise(Bool2<false>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (su == 5) B2STAMP(15);
            __builtin_amdgcn_s_barrier();           // both halves of the dh / dq tiles of this row group are complete
            if (su == 5) B2STAMP(16);
            uint8_t* mine = isA ? t0 : t1;
            store_rows4(isA ? DH : DQ, rl, su * 128, mine, rg, lane);
            if (su == 5) B2STAMP(17);
            Frag<NS> df[G::E4], wf[G::E4 * RT];
#pragma unroll
            for (int e = 0; e < G::E4; ++e) {
                float v8[8];
                tile_lane_vals8<IO>(mine, trow, h, e, v8);
                df[e] = frag_from_f32<NS>(v8);
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) wf[e * RT + ct] = wfrag<NS>(w, e * RT + ct, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < G::E4; ++e) {
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) dz[ct] = mfma_ns<NS>(wf[e * RT + ct], df[e], dz[ct]);
            }
            if (su == 5) B2STAMP(18);
            wait_vm(rl.n_inst);
            if (su == 5) B2STAMP(19);
            __builtin_amdgcn_s_barrier();
            if (su == 5) B2STAMP(20);
            ++st;
        }
    }
    B2STAMP(2);

    // ---- the dh rows of the last phase: this wave's own stores must have completed; start their stream, then dpre
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!LR) {
        issue_last_rows(0, false);
        issue_last_rows(1, false);
    }
    const int ldz = 32 * RT;
    Frag<NS> dp[KT];
    {
        IO* dps = reinterpret_cast<IO*>(isA ? a.dp_a : a.dp_g);
        const float sc = isA ? sd_ : 1.0f;          // dz_a = sd * Wu^T dh: the delta scale once, here
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) {
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = sc * dz[ct][8 * sh + j] * gp[ct][8 * sh + j];
                dp[2 * ct + sh] = frag_from_f32<NS>(v);
                if (row_ok) store8_f32(dps + grow * ldz + 32 * ct + 16 * sh + 8 * h, v);
            }
        }
    }
    if constexpr (LR) return;                       // no input gradients in the low-rank projector form
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // (slot 2 was the exchange buffer: every wave is past its last read)

    B2STAMP(3);
    // ---- last phase: input gradients
    uint8_t* dxo = reinterpret_cast<uint8_t*>(isA ? a.dxa : a.dxg);
    for (int su = 0; su < S; ++su, ++st) {
        if (su == 5) B2STAMP(24);
        int nrows = 0;
        if (!isA) { issue_w(st + 1, true); nrows = issue_last_rows(su + 2, true); }
        if (su == 5) B2STAMP(25);
        const uint8_t* w = slot_w(st & 1);
        uint8_t* tile = isA ? slot_t0(su % 3) : slot_t1(su % 3);
        f32x16 ax[G::NV];
#pragma unroll
        for (int v = 0; v < G::NV; ++v) ax[v] = zero16();
        {
            Frag<NS> wf[G::NV * KT];
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
#pragma unroll
                for (int v = 0; v < G::NV; ++v) wf[v * KT + ks] = wfrag<NS>(w, v * KT + ks, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
#pragma unroll
                for (int v = 0; v < G::NV; ++v) ax[v] = mfma_ns<NS>(wf[v * KT + ks], dp[ks], ax[v]);
            }
        }
        if (su == 5) B2STAMP(26);
#pragma unroll
        for (int e = 0; e < G::E4; ++e) {
            float o8[8], dh8[8];
            const bool add_in = !isA && DXIN != nullptr;
            if (isA || add_in) tile_lane_vals8<IO>(tile, trow, h, e, dh8);        // A: dh;  G: the incoming dx1
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = 8 * e + j;
                float v = ax[(i >> 4) % G::NV][i & 15];
                if (isA) v += s2 * dh8[j];
                else if (add_in) v += dh8[j];
                o8[j] = v;
            }
            stage_lane_vals8<IO>(tile, trow, h, e, o8);
        }
        if (su == 5) B2STAMP(27);
        store_rows4(dxo, rl, su * 128, tile, rg, lane);
        if (su == 5) B2STAMP(28);
        wait_vm(nrows + rl.n_inst);
        if (su == 5) B2STAMP(29);
        __builtin_amdgcn_s_barrier();
        if (su == 5) B2STAMP(30);
    }
    B2STAMP(4);
}

template <typename IO, int RT, int RG, bool LR = false>
static hipError_t launch_one(const PetBwdArgs& a, hipStream_t stream) {
    using L = Bwd2Lds<IO, RT, RG>;
    const size_t lds = L::bytes(a.d);
    auto kern = pet_gate_bwd2_kernel<IO, RT, RG, LR>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int rows = RG * 32;
    const int blocks = (int)((a.M + rows - 1) / rows);
#ifdef VLPET_STAMPS
    PetBwdArgs b = a;
    if (vlpet_tuning().dbg & 128) b.flags |= 1 << 20;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(RG * 128), lds, stream, b);
    if (vlpet_tuning().dbg & 16) {
        (void)hipDeviceSynchronize();
        unsigned long long t[64];
        (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_b2_ts), sizeof(t));
        auto dd = [&](int i, int j) { return (long long)(t[j] - t[i]); };
        fprintf(stderr, "[vlpet bwd2 ts] prologue=%lld middle=%lld dpre=%lld last=%lld | stage a: issue=%lld reads+mfma=%lld xwrite=%lld wait=%lld barrier=%lld | "
                        "stage b: issue=%lld elementwise=%lld barrier1=%lld store=%lld reads+mfma=%lld wait=%lld barrier2=%lld | last: issue=%lld mfma=%lld epilogue=%lld store=%lld wait=%lld barrier=%lld\n",
                dd(0, 1), dd(1, 2), dd(2, 3), dd(3, 4), dd(8, 9), dd(9, 10), dd(10, 11), dd(11, 12), dd(12, 13),
                dd(13, 14), dd(14, 15), dd(15, 16), dd(16, 17), dd(17, 18), dd(18, 19), dd(19, 20),
                dd(24, 25), dd(25, 26), dd(26, 27), dd(27, 28), dd(28, 29), dd(29, 30));
    }
#else
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(RG * 128), lds, stream, a);
#endif
    return hipGetLastError();
}

template <typename IO, int RT>
static hipError_t launch_rt(const PetBwdArgs& a, hipStream_t stream) {
    switch (pick_row_groups(a.M, 4, 2)) {
        case 4: return launch_one<IO, RT, 4>(a, stream);
        case 3: if constexpr ((4 * RT) % 3 == 0) return launch_one<IO, RT, 3>(a, stream);   // (else: falls through)
        default: return launch_one<IO, RT, 2>(a, stream);
    }
}

// true when this form applies: gated K1, saved activations, r <= 96, every dimension it was written for
bool pet_gate_bwd2_applies(const PetBwdArgs& a) {
    if (vlpet_tuning().bwd2 == 0) return false;
    return (a.flags & PET_GATE) && a.saved != nullptr && !drop_active(a.drop) && (a.RT == 1 || a.RT == 3);
}

hipError_t launch_pet_gate_bwd2(const PetBwdArgs& a, int io_fp32, hipStream_t stream) {
    if (a.RT == 1) return io_fp32 ? launch_rt<float, 1>(a, stream) : launch_rt<__bf16, 1>(a, stream);
    if (a.RT == 3) return io_fp32 ? launch_rt<float, 3>(a, stream) : launch_rt<__bf16, 3>(a, stream);
    return hipErrorInvalidValue;
}

// low-rank visual projector form: saved activations, multiplicative gate, r, r_g <= 96
template <typename IO, int RT>
static hipError_t launch_lr(const PetBwdArgs& a, hipStream_t stream) {
    switch (pick_row_groups(a.M, 4, 2)) {
        case 4: return launch_one<IO, RT, 4, true>(a, stream);
        case 3: if constexpr ((4 * RT) % 3 == 0) return launch_one<IO, RT, 3, true>(a, stream);   // (else: falls through)
        default: return launch_one<IO, RT, 2, true>(a, stream);
    }
}
hipError_t launch_pet_lowrank_bwd(const PetBwdArgs& a, int io_fp32, hipStream_t stream) {
    if (!a.saved || !(a.flags & PET_GATE) || (a.flags & PET_GATE_ADD) || drop_active(a.drop)) return hipErrorInvalidValue;
    if (a.RT == 1) return io_fp32 ? launch_lr<float, 1>(a, stream) : launch_lr<__bf16, 1>(a, stream);
    if (a.RT == 3) return io_fp32 ? launch_lr<float, 3>(a, stream) : launch_lr<__bf16, 3>(a, stream);
    return hipErrorInvalidValue;
}
