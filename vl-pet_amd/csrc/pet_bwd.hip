// Fused PET backward, row-parallel part (same structure as pet_fwd.hip, see pet32.h):
//   recompute z / zg / h / g from (xa, xg, res) -- no [M,r] or [M,d] intermediate was saved by the
//   forward -- then
//     dh = gs*dy (*g)           dq = gs*dy*h*g*(1-g)          (gate)
//     dz  = (sd*dh) . Wu        dpre  = dz  * gelu'(pre)      dxa = s2*dh + dpre  . Wd
//     dzg =  dq     . Wgu       dpreg = dzg * gelu'(preg)     dxg =         dpreg . Wgd
//   and leave the row-major side products (z, dpre, zg, dpreg [M,32RT]; dh, dq [M,d]) for the
//   column-parallel weight-gradient kernel (wgrad.hip), because a row-parallel kernel cannot hold
//   the [d x r] weight-gradient accumulators of four matrices (4 x 295 KB fp32 per workgroup).
// Autograd of: my_transformers/modeling_bart.py:1147-1155,1195-1209 (K1);
// adapters/adapter_modeling.py:55-61 (K2); lora/controller.py:56-70 (K3).
//
// Stage stream (a stage = <= 8RT KiB of weight fragments through the 2-slot weight ring + <= 2 row
// tensors through the 3-slot row ring, rows prefetched two stages ahead):
//   gate:     [down A|G + xa,xg] x S   [ (up A|G + res,dy), (up_t A|G) ] x S   [down_t A|G + dh] x S
//   no gate:  [down A + xa] x S        [ up_t A + dy ] x S                      [down_t A] x S
// The dh rows of the last phase are re-read from the dh side product this workgroup stored itself;
// their stream starts once those stores have completed, before the dpre block.
// With the forward's saved activations (PetBwdArgs::saved: z and act'(pre) of each chain) the first phase does
// not exist: the kernel starts at the middle phase, reads x1 not at all and x2 once, and leaves the z side
// products to the forward's copy.  Rows per workgroup (WAVES = 2 / 3 / 4 row groups) are chosen per launch
// (kernels.h pick_row_groups): a workgroup's time hardly depends on its rows, the number of 256-workgroup
// rounds does.
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "pet32.h"

template <bool B> struct BoolK { static constexpr bool value = B; };

template <typename IO, int RT, bool GATE, int WAVES>
struct BwdLds {
    static constexpr int NS = Geo4<IO>::NS;
    static constexpr int SEG_KB = 4 * RT;
    static constexpr int SEG_FR = SEG_KB / NS;
    static constexpr int W_B = SEG_KB * 1024 * (GATE ? 2 : 1);
    static constexpr int TILE_B = WAVES * 32 * 128;
    static constexpr int ROW_B = TILE_B * (GATE ? 2 : 1);
    static constexpr int NR = 3;
    static constexpr int ROW_OFF = 2 * W_B;
    static constexpr int BIAS_OFF = ROW_OFF + NR * ROW_B;
    static size_t bytes(int d) { return (size_t)BIAS_OFF + (size_t)2 * (32 * RT + d) * 4; }
};

#ifdef VLPET_STAMPS
// diagnosis build only (-DVLPET_STAMPS): cycle stamps of wave 0 of workgroup 0, printed after the launch when VLPET_DBG & 16
__device__ unsigned long long g_bwd_ts[64];
#define BSTAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_bwd_ts[k] = __builtin_readcyclecounter(); } while (0)
#else
#define BSTAMP(k) do { } while (0)
#endif

template <typename IO, int RT, bool GATE, bool ACT_ID, bool DROP, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void pet_bwd_kernel(PetBwdArgs a) {
    using G = Geo4<IO>;
    using L = BwdLds<IO, RT, GATE, WAVES>;
    constexpr int NS = G::NS;
    constexpr int KT = 2 * RT;
    constexpr bool PIPE = NS == 1 && RT <= 3;      // LDS fragment reads one k-step ahead (bf16 IO; fp32 IO has no registers left)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar registers and scalar branches for everything derived from it
    const int m = lane & 31, h = lane >> 5;
    const int trow = 32 * wave + m;
    const int d = a.d;
    const int64_t row0_wave = (int64_t)blockIdx.x * (WAVES * 32) + wave * 32;
    const int64_t grow_raw = row0_wave + m;
    const bool row_ok = grow_raw < a.M;
    const int64_t grow = row_ok ? grow_raw : a.M - 1;
    const int S = d / G::FE;
    const PackGeom pg = pack_geom(RT, d, NS);
    const uint8_t* pkA = a.pk_a;
    const uint8_t* pkG = GATE ? a.pk_g : a.pk_a;
    const uint8_t* xa = reinterpret_cast<const uint8_t*>(a.xa);
    const uint8_t* xg = reinterpret_cast<const uint8_t*>(a.xg);
    const uint8_t* res = reinterpret_cast<const uint8_t*>(a.res);
    const uint8_t* dy = reinterpret_cast<const uint8_t*>(a.dy);
    uint8_t* DH = reinterpret_cast<uint8_t*>(a.dh);
    uint8_t* DQ = reinterpret_cast<uint8_t*>(a.dq);

    auto slot_w = [&](int j) { return smem + (size_t)j * L::W_B; };
    auto slot_t0 = [&](int j) { return smem + L::ROW_OFF + (size_t)j * L::ROW_B; };
    auto slot_t1 = [&](int j) { return smem + L::ROW_OFF + (size_t)j * L::ROW_B + L::TILE_B; };
    float* sb = reinterpret_cast<float*>(smem + L::BIAS_OFF);
    const int nb = 32 * RT + d;
    const int NU = GATE ? 2 * S : S;            // stages of the middle phase
    const int P3 = S + NU;                      // first stage of the input-gradient phase
    const int total = P3 + S;

    const RowLanes rl = row_lanes<IO>(row0_wave, a.M, d, wave, lane);
    const DropSpec drop_r = DROP ? drop_resolved(a.drop) : a.drop;      // (the step counter folded into the seed: rng.h)
    const int lane16 = lane * 16;

    // stage decoding: weight pack / pack stage, row tensors (t0, t1) and their stage offset
    struct StageInfo { int pack, ss; const uint8_t* t0; const uint8_t* t1; };
    auto info = [&](int s) {
        StageInfo I{0, 0, nullptr, nullptr};
        if (s < S) { I.pack = 0; I.ss = s; I.t0 = xa; if constexpr (GATE) I.t1 = xg; }
        else if (s < P3) {
            const int u = s - S;
            if constexpr (GATE) {
                I.ss = u >> 1;
                if (u & 1) I.pack = 2; else { I.pack = 1; I.t0 = res; I.t1 = dy; }
            } else { I.ss = u; I.pack = 2; I.t0 = dy; }
        } else { I.pack = 3; I.ss = s - P3; if constexpr (GATE) I.t0 = DH; }
        return I;
    };
    // rows of stage s2 are issued two stages ahead, except the dh rows of phase 3 (see header)
    auto rows_deferred = [&](int s2, int from) { return GATE && s2 >= P3 && from < P3; };
    auto rows_count = [&](int s2, int from) {
        if (s2 >= total || rows_deferred(s2, from)) return 0;
        const StageInfo I = info(s2);
        return (I.t0 ? 4 : 0) + (I.t1 ? 4 : 0);
    };
    auto issue_rows = [&](int s2, int from) {
        if (s2 >= total || rows_deferred(s2, from)) return;
        const StageInfo I = info(s2);
        const int j = s2 % L::NR;
        if (I.t0) glds_rows4(I.t0, rl, I.ss * 128, slot_t0(j), wave);
        if constexpr (GATE) {
            if (I.t1) glds_rows4(I.t1, rl, I.ss * 128, slot_t1(j), wave);
        }
    };
    // weight pieces of stage s1: this wave's share is pieces wave, wave+WAVES, ... of each segment.  The per-piece
    // address work is what a memory instruction costs here (stamps: 90-150 cycles per global_load_lds with the piece
    // index decoded inside the loop, mostly scalar address arithmetic), so: one scalar base per segment and stage, one
    // per-lane offset computed once, constant strides.
    const uint32_t wv_off = (uint32_t)(wave * 1024 + lane16);
    auto issue_w = [&](int s1) {
        if (s1 >= total) return;
        const StageInfo I = info(s1);
        const int64_t woff = (int64_t)I.pack * pg.pack_bytes + (int64_t)I.ss * L::SEG_KB * 1024;
        uint8_t* dst = slot_w(s1 & 1) + wave * 1024;
        if constexpr (L::SEG_KB % WAVES == 0) {
            const uint8_t* bA = pkA + woff;
            const uint8_t* bG = pkG + woff;
#pragma unroll
            for (int j = 0; j < L::SEG_KB / WAVES; ++j) glds16(bA + (wv_off + j * WAVES * 1024), dst + j * WAVES * 1024);
            if constexpr (GATE) {
#pragma unroll
                for (int j = 0; j < L::SEG_KB / WAVES; ++j)
                    glds16(bG + (wv_off + j * WAVES * 1024), dst + L::SEG_KB * 1024 + j * WAVES * 1024);
            }
        } else {
            constexpr int KB = L::SEG_KB * (GATE ? 2 : 1);
            for (int k = wave; k < KB; k += WAVES) {
                const uint8_t* src = (k < L::SEG_KB ? pkA + woff + (size_t)k * 1024
                                                    : pkG + woff + (size_t)(k - L::SEG_KB) * 1024) + lane16;
                glds16(src, dst - wave * 1024 + (size_t)k * 1024);
            }
        }
    };

    // ---- the memory instructions of a stage as one flat list (weights of stage s+1, then rows of stage s+2: the order
    // the counted vmcnt waits rely on) with every address precomputed per stage.  The texture path takes one 1 KiB piece
    // per ~16 cycles and four waves feed it, so a wave's 14-20 pieces cost it >= 0.9-1.3 k cycles per stage however they
    // are placed: handing them out between the MFMA groups was measured 3 % SLOWER than issuing them up front (in-order
    // issue: the MFMAs behind a blocked global_load_lds wait with it).
    constexpr int PW = (L::SEG_KB % WAVES == 0) ? L::SEG_KB / WAVES : 0;      // weight pieces per segment and wave
    constexpr int NWP = PW * (GATE ? 2 : 1);
    constexpr int NPIECE = NWP + (GATE ? 8 : 4);
    struct Plan { const uint8_t* wA; const uint8_t* wG; uint8_t* wd; const uint8_t* r0; const uint8_t* r1; uint8_t* d0; uint8_t* d1; };
    auto plan = [&](int sc) {          // what stage sc issues
        Plan P{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        if constexpr (PW == 0) { issue_w(sc + 1); issue_rows(sc + 2, sc); return P; }
        if (sc + 1 < total) {
            const StageInfo I = info(sc + 1);
            const int64_t woff = (int64_t)I.pack * pg.pack_bytes + (int64_t)I.ss * L::SEG_KB * 1024;
            P.wA = pkA + woff; P.wG = pkG + woff;
            P.wd = slot_w((sc + 1) & 1) + wave * 1024;
        }
        if (sc + 2 < total && !rows_deferred(sc + 2, sc)) {
            const StageInfo I = info(sc + 2);
            const int j = (sc + 2) % L::NR;
            if (I.t0) { P.r0 = I.t0 + I.ss * 128; P.d0 = slot_t0(j) + (size_t)(32 * wave) * 128; }
            if constexpr (GATE) { if (I.t1) { P.r1 = I.t1 + I.ss * 128; P.d1 = slot_t1(j) + (size_t)(32 * wave) * 128; } }
        }
        return P;
    };
    auto emit = [&](const Plan& P, int lo, int hi) {       // pieces lo .. hi-1 of the stage's list
        if constexpr (PW == 0) return;
#pragma unroll
        for (int i = lo; i < hi && i < NPIECE; ++i) {
            if (i < NWP) {
                const int seg = i / (PW > 0 ? PW : 1), j = i % (PW > 0 ? PW : 1);
                if (P.wd) glds16((seg ? P.wG : P.wA) + (wv_off + j * WAVES * 1024), P.wd + seg * L::SEG_KB * 1024 + j * WAVES * 1024);
            } else {
                const int k = (i - NWP) / 4, q = (i - NWP) % 4;
                const uint8_t* base = k ? P.r1 : P.r0;
                uint8_t* dst = k ? P.d1 : P.d0;
                if (base) glds16_row(base + rl.off[q], dst + q * 1024);
            }
        }
    };
    // with the forward's saved activations phase 1 (stages 0 .. S-1) does not exist
    const bool use_saved = a.saved != nullptr;      // (K3: z only -- the identity activation has no act'(pre) to save)
    const int S0 = use_saved ? S : 0;
    BSTAMP(0);
    issue_w(S0);
    issue_rows(S0, S0);
    issue_rows(S0 + 1, S0);
    {
        copy_bias<WAVES * 64>(sb, reinterpret_cast<const float*>(a.pk_a + pg.bias_off), nb, tid);
        if constexpr (GATE) copy_bias<WAVES * 64>(sb + nb, reinterpret_cast<const float*>(a.pk_g + pg.bias_off), nb, tid);
    }
    f32x16 accA[RT];
    f32x16 accG[GATE ? RT : 1];
    Frag<NS> zA[KT];
    Frag<NS> zG[GATE ? KT : 1];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) accA[ct] = zero16();
    if constexpr (GATE) {
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) accG[ct] = zero16();
    }
    // the forward's saved activations are fetched while the prologue's weight / row pieces are still landing
    if (use_saved) {
        const uint8_t* sv = reinterpret_cast<const uint8_t*>(a.saved);
        const int64_t ro = grow * (int64_t)(32 * RT) + 8 * h;
        const IO* sza = reinterpret_cast<const IO*>(sv) + ro;
        const IO* sga = reinterpret_cast<const IO*>(sv + a.saved_stride) + ro;
        const IO* szg = reinterpret_cast<const IO*>(sv + 2 * a.saved_stride) + ro;
        const IO* sgg = reinterpret_cast<const IO*>(sv + 3 * a.saved_stride) + ro;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) {
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                float v[8];
                load8_f32(sza + 32 * ct + 16 * sh, v);
                zA[2 * ct + sh] = frag_from_f32<NS>(v);
                if constexpr (ACT_ID) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) accA[ct][8 * sh + j] = 1.0f;
                } else {
                    load8_f32(sga + 32 * ct + 16 * sh, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) accA[ct][8 * sh + j] = v[j];
                }
                if constexpr (GATE) {
                    load8_f32(szg + 32 * ct + 16 * sh, v);
                    zG[2 * ct + sh] = frag_from_f32<NS>(v);
                    load8_f32(sgg + 32 * ct + 16 * sh, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) accG[ct][8 * sh + j] = v[j];
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 1: recompute the bottleneck pre-activations of both chains
    int s = S0;
    BSTAMP(1);
    for (; s < S; ++s) {
        if (s == 5) BSTAMP(8);
        const Plan pl = plan(s);
        emit(pl, 0, NPIECE);
        if (s == 5) BSTAMP(9);
        const uint8_t* w = slot_w(s & 1);
        const uint8_t* ta = slot_t0(s % L::NR);
        const uint8_t* tg = slot_t1(s % L::NR);
        // k-steps are software-pipelined: the LDS fragments of k-step u+1 are requested before the MFMAs of k-step u
        // (a lone wave per SIMD has nobody else to hide the LDS latency behind; left alone, hipcc issues each read
        // right before its use and waits for it).  One k-step ahead, not a whole stage: the kernel is at 512 registers.
        struct KFr { Frag<NS> bA, bG, wA[RT], wG[GATE ? RT : 1]; };
        auto load_k = [&](int u) {
            KFr f;
            f.bA = tile_bfrag4<IO>(ta, trow, h, u);
            if constexpr (GATE) f.bG = tile_bfrag4<IO>(tg, trow, h, u);
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) {
                f.wA[ct] = wfrag<NS>(w, u * RT + ct, lane);
                if constexpr (GATE) f.wG[ct] = wfrag<NS>(w, L::SEG_FR + u * RT + ct, lane);
            }
            return f;
        };
        KFr cur;
                if constexpr (PIPE) cur = load_k(0);
#pragma unroll
        for (int u = 0; u < G::KU; ++u) {
            KFr nxt;
            if constexpr (PIPE) {
                if (u + 1 < G::KU) nxt = load_k(u + 1);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                cur = load_k(u);
            }
            Frag<NS> bA = cur.bA;
            if constexpr (DROP) {
                const uint32_t kb = drop_bits8(drop_r, grow, s * G::FE + 16 * u + 8 * h, d);
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = (float)bA.p[0][j];
                    if constexpr (NS == 2) v[j] += (float)bA.p[1][j];
                    v[j] = ((kb >> j) & 1u) ? v[j] * a.drop.keep_scale : 0.f;
                }
                bA = frag_from_f32<NS>(v);
            }
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) {
                accA[ct] = mfma_ns<NS>(cur.wA[ct], bA, accA[ct]);
                if constexpr (GATE) accG[ct] = mfma_ns<NS>(cur.wG[ct], cur.bG, accG[ct]);
            }
            if constexpr (PIPE) { if (u + 1 < G::KU) cur = nxt; }
        }
        if (s == 5) BSTAMP(10);
        wait_vm(rows_count(s + 2, s));
        if (s == 5) BSTAMP(11);
        __builtin_amdgcn_s_barrier();
        if (s == 5) BSTAMP(12);
    }
    BSTAMP(2);

    // z = act(pre) as B fragments; accA / accG are overwritten with act'(pre)
    if (!use_saved) {
        {
            const float* bdA = sb + 8 * h;
            const float* bdG = sb + nb + 8 * h;
    #pragma unroll
            for (int ct = 0; ct < RT; ++ct) {
    #pragma unroll
                for (int sh = 0; sh < 2; ++sh) {
                    float v[8];
    #pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float pre = accA[ct][8 * sh + j] + bdA[32 * ct + 16 * sh + j];
                        v[j] = ACT_ID ? pre : gelu_new_f(pre);
                        accA[ct][8 * sh + j] = ACT_ID ? 1.0f : gelu_new_grad_f(pre);
                    }
                    zA[2 * ct + sh] = frag_from_f32<NS>(v);
                    if constexpr (GATE) {
    #pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float pre = accG[ct][8 * sh + j] + bdG[32 * ct + 16 * sh + j];
                            v[j] = gelu_new_f(pre);
                            accG[ct][8 * sh + j] = gelu_new_grad_f(pre);
                        }
                        zG[2 * ct + sh] = frag_from_f32<NS>(v);
                    }
                }
            }
        }
    }

    // ---- phase 2: elementwise backward per feature group, then contraction over features
    const float* buA = sb + 32 * RT + G::LW * h;
    const float* buG = sb + nb + 32 * RT + G::LW * h;
    const float s2 = a.s2, sd_ = a.sd, gs = a.gs;
    const bool gate_add = (a.flags & PET_GATE_ADD) != 0;
    f32x16 dzA[RT];
    f32x16 dzG[GATE ? RT : 1];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) dzA[ct] = zero16();
    if constexpr (GATE) {
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) dzG[ct] = zero16();
    }
    BSTAMP(3);
    for (int su = 0; su < S; ++su) {
        Frag<NS> dfA[G::E4], dfG[G::E4];
        if constexpr (GATE) {
            if (su == 5) BSTAMP(16);
            const Plan pl = plan(s);
        emit(pl, 0, NPIECE);
            if (su == 5) BSTAMP(17);
            const uint8_t* w = slot_w(s & 1);
            uint8_t* t0 = slot_t0(s % L::NR);
            uint8_t* t1 = slot_t1(s % L::NR);
            f32x16 aA[G::NV], aG[G::NV];
#pragma unroll
            for (int v = 0; v < G::NV; ++v) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 ta = *reinterpret_cast<const f32x4*>(buA + su * G::FE + 16 * v + 4 * q);
                    const f32x4 tg = *reinterpret_cast<const f32x4*>(buG + su * G::FE + 16 * v + 4 * q);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { aA[v][4 * q + c] = ta[c]; aG[v][4 * q + c] = tg[c]; }
                }
            }
            {   // weight fragments one k-step ahead of their MFMAs (see phase 1)
                struct UFr { Frag<NS> wA[G::NV], wG[G::NV]; };
                auto load_u = [&](int ks) {
                    UFr f;
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) {
                        f.wG[v] = wfrag<NS>(w, L::SEG_FR + v * KT + ks, lane);
                        f.wA[v] = wfrag<NS>(w, v * KT + ks, lane);
                    }
                    return f;
                };
                UFr cur;
                if constexpr (PIPE) cur = load_u(0);
#pragma unroll
                for (int ks = 0; ks < KT; ++ks) {
                    UFr nxt;
                    if constexpr (PIPE) {
                        if (ks + 1 < KT) nxt = load_u(ks + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        cur = load_u(ks);
                    }
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) {
                        aG[v] = mfma_ns<NS>(cur.wG[v], zG[ks], aG[v]);
                        aA[v] = mfma_ns<NS>(cur.wA[v], zA[ks], aA[v]);
                    }
                    if constexpr (PIPE) { if (ks + 1 < KT) cur = nxt; }
                }
            }
            if (su == 5) BSTAMP(18);
            // elementwise backward in fragment-sized steps (8 features): dh / dq are staged in place of the res / dy
            // rows of this wave and become the B fragments of the feature contraction (dh itself: the delta scale of
            // dz = sd * Wu^T dh is applied once to dz after the phase).  The multiplicative / additive forms are two
            // copies of the loop (a per-element select on a wave-uniform flag otherwise).
            auto elementwise = [&](auto add_c) {
                constexpr bool ADD = decltype(add_c)::value;
#pragma unroll
                for (int e = 0; e < G::E4; ++e) {
                    float r8[8], dy8[8], dh8[8], dq8[8];
                    tile_lane_vals8<IO>(t0, trow, h, e, r8);
                    tile_lane_vals8<IO>(t1, trow, h, e, dy8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = 8 * e + j;
                        const float gt = sigmoid_f(aG[i >> 4][i & 15]);
                        const float dyp = gs * dy8[j];
                        if constexpr (ADD) {
                            dh8[j] = dyp;
                            dq8[j] = dyp * gt * (1.0f - gt);
                        } else {
                            const float hv = s2 * r8[j] + sd_ * aA[i >> 4][i & 15];
                            dh8[j] = dyp * gt;
                            dq8[j] = dh8[j] * hv * (1.0f - gt);
                        }
                    }
                    stage_lane_vals8<IO>(t0, trow, h, e, dh8);
                    stage_lane_vals8<IO>(t1, trow, h, e, dq8);
                    dfA[e] = frag_from_f32<NS>(dh8);
                    dfG[e] = frag_from_f32<NS>(dq8);
                }
            };
            if (gate_add) elementwise(BoolK<true>{}); else elementwise(BoolK<false>{});
            if (su == 5) BSTAMP(19);
            store_rows4(DH, rl, su * 128, t0, wave, lane);
            store_rows4(DQ, rl, su * 128, t1, wave, lane);
            if (su == 5) BSTAMP(20);
            wait_vm(rows_count(s + 2, s) + 2 * rl.n_inst);
            if (su == 5) BSTAMP(21);
            __builtin_amdgcn_s_barrier();
            if (su == 5) BSTAMP(22);
            ++s;
        }
        {
            const Plan pl = plan(s);
        emit(pl, 0, NPIECE);
            const uint8_t* w = slot_w(s & 1);
            if constexpr (!GATE) {
                float dyv[G::LW];
                tile_lane_vals4<IO>(slot_t0(s % L::NR), trow, h, dyv);
#pragma unroll
                for (int e = 0; e < G::E4; ++e) dfA[e] = frag_from_f32<NS>(dyv + 8 * e);
            }
            {
                struct CFr { Frag<NS> wA[RT], wG[GATE ? RT : 1]; };
                auto load_c = [&](int e) {
                    CFr f;
#pragma unroll
                    for (int ct = 0; ct < RT; ++ct) {
                        f.wA[ct] = wfrag<NS>(w, e * RT + ct, lane);
                        if constexpr (GATE) f.wG[ct] = wfrag<NS>(w, L::SEG_FR + e * RT + ct, lane);
                    }
                    return f;
                };
                CFr cur;
                if constexpr (PIPE) cur = load_c(0);
#pragma unroll
                for (int e = 0; e < G::E4; ++e) {
                    CFr nxt;
                    if constexpr (PIPE) {
                        if (e + 1 < G::E4) nxt = load_c(e + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        cur = load_c(e);
                    }
#pragma unroll
                    for (int ct = 0; ct < RT; ++ct) {
                        dzA[ct] = mfma_ns<NS>(cur.wA[ct], dfA[e], dzA[ct]);
                        if constexpr (GATE) dzG[ct] = mfma_ns<NS>(cur.wG[ct], dfG[e], dzG[ct]);
                    }
                    if constexpr (PIPE) { if (e + 1 < G::E4) cur = nxt; }
                }
            }
            if (su == 5) BSTAMP(23);
            wait_vm(rows_count(s + 2, s));
            __builtin_amdgcn_s_barrier();
            if (su == 5) BSTAMP(24);
            ++s;
        }
    }
    BSTAMP(4);

    if constexpr (GATE) {
        // The dh rows of phase 3 are read back from the dh side product this wave stored itself: once those stores
        // have completed, start the row stream -- it lands while the dpre block below computes and stores.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        issue_rows(P3, P3);
        issue_rows(P3 + 1, P3);
    }
    // ---- dpre = dz * act'(pre); row-major side products for the weight-gradient kernel
    const int ldz = 32 * RT;
    Frag<NS> dpA[KT];
    Frag<NS> dpG[GATE ? KT : 1];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) {
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
            float v[8], zv[8];
            const int col = 32 * ct + 16 * sh + 8 * h;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = sd_ * dzA[ct][8 * sh + j] * accA[ct][8 * sh + j];
            dpA[2 * ct + sh] = frag_from_f32<NS>(v);
            if (row_ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    zv[j] = (float)zA[2 * ct + sh].p[0][j];
                    if constexpr (NS == 2) zv[j] += (float)zA[2 * ct + sh].p[1][j];
                }
                if (!use_saved) store8_f32(reinterpret_cast<IO*>(a.z_a) + grow * ldz + col, zv);
                store8_f32(reinterpret_cast<IO*>(a.dp_a) + grow * ldz + col, v);
            }
            if constexpr (GATE) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = dzG[ct][8 * sh + j] * accG[ct][8 * sh + j];
                dpG[2 * ct + sh] = frag_from_f32<NS>(v);
                if (row_ok) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        zv[j] = (float)zG[2 * ct + sh].p[0][j];
                        if constexpr (NS == 2) zv[j] += (float)zG[2 * ct + sh].p[1][j];
                    }
                    if (!use_saved) store8_f32(reinterpret_cast<IO*>(a.z_g) + grow * ldz + col, zv);
                    store8_f32(reinterpret_cast<IO*>(a.dp_g) + grow * ldz + col, v);
                }
            }
        }
    }

    // ---- phase 3: input gradients
    uint8_t* dxa = reinterpret_cast<uint8_t*>(a.dxa);
    uint8_t* dxg = reinterpret_cast<uint8_t*>(a.dxg);
    if constexpr (GATE) {
        // rows of the first two phase-3 stages and the side-product stores issued after them (a counted wait would have
        // to know how many of those predicated stores this wave really issued)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    BSTAMP(5);
    for (int su = 0; su < S; ++su, ++s) {
        if (su == 5) BSTAMP(32);
        const Plan pl = plan(s);
        emit(pl, 0, NPIECE);
        const uint8_t* w = slot_w(s & 1);
        uint8_t* t0 = slot_t0(s % L::NR);
        uint8_t* t1 = slot_t1(s % L::NR);
        f32x16 aA[G::NV], aG[G::NV];
#pragma unroll
        for (int v = 0; v < G::NV; ++v) { aA[v] = zero16(); aG[v] = zero16(); }
        {
            struct DFr { Frag<NS> wA[G::NV], wG[GATE ? G::NV : 1]; };
            auto load_d = [&](int ks) {
                DFr f;
#pragma unroll
                for (int v = 0; v < G::NV; ++v) {
                    f.wA[v] = wfrag<NS>(w, v * KT + ks, lane);
                    if constexpr (GATE) f.wG[v] = wfrag<NS>(w, L::SEG_FR + v * KT + ks, lane);
                }
                return f;
            };
            DFr cur;
                if constexpr (PIPE) cur = load_d(0);
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
                DFr nxt;
                if constexpr (PIPE) {
                    if (ks + 1 < KT) nxt = load_d(ks + 1);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    cur = load_d(ks);
                }
#pragma unroll
                for (int v = 0; v < G::NV; ++v) {
                    aA[v] = mfma_ns<NS>(cur.wA[v], dpA[ks], aA[v]);
                    if constexpr (GATE) aG[v] = mfma_ns<NS>(cur.wG[v], dpG[ks], aG[v]);
                }
                if constexpr (PIPE) { if (ks + 1 < KT) cur = nxt; }
            }
        }
        if (su == 5) BSTAMP(33);
        uint32_t kp[4] = {0, 0, 0, 0};
        if constexpr (DROP) {
#pragma unroll
            for (int c = 0; c < G::LW / 8; ++c) kp[c] = drop_bits8(drop_r, grow, su * G::FE + G::LW * h + 8 * c, d);
        }
#pragma unroll
        for (int e = 0; e < G::E4; ++e) {
            float oa8[8], og8[8], dh8[8];
            if constexpr (GATE) tile_lane_vals8<IO>(t0, trow, h, e, dh8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = 8 * e + j;
                float v = aA[i >> 4][i & 15];
                if constexpr (GATE) v += s2 * dh8[j];
                if constexpr (DROP) v = ((kp[i >> 3] >> (i & 7)) & 1u) ? v * a.drop.keep_scale : 0.f;
                oa8[j] = v;
                if constexpr (GATE) og8[j] = aG[i >> 4][i & 15];
            }
            stage_lane_vals8<IO>(t0, trow, h, e, oa8);
            if constexpr (GATE) stage_lane_vals8<IO>(t1, trow, h, e, og8);
        }
        store_rows4(dxa, rl, su * 128, t0, wave, lane);
        if constexpr (GATE) store_rows4(dxg, rl, su * 128, t1, wave, lane);
        if (su == 5) BSTAMP(34);
        wait_vm(rows_count(s + 2, s) + (GATE ? 2 : 1) * rl.n_inst);
        __builtin_amdgcn_s_barrier();
        if (su == 5) BSTAMP(35);
    }
    BSTAMP(6);
}

template <typename IO, int RT, bool GATE, bool ACT_ID, bool DROP, int WAVES>
static hipError_t launch_one(const PetBwdArgs& a, hipStream_t stream) {
    using L = BwdLds<IO, RT, GATE, WAVES>;
    const size_t lds = L::bytes(a.d);
    auto kern = pet_bwd_kernel<IO, RT, GATE, ACT_ID, DROP, WAVES>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int rows = WAVES * 32;
    const int blocks = (int)((a.M + rows - 1) / rows);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), lds, stream, a);
#ifdef VLPET_STAMPS
    if ((vlpet_tuning().dbg & 16) && GATE) {
        (void)hipDeviceSynchronize();
        unsigned long long t[64];
        (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_bwd_ts), sizeof(t));
        auto d = [&](int i, int j) { return (long long)(t[j] - t[i]); };
        fprintf(stderr, "[vlpet bwd ts] prologue=%lld phase1=%lld gelu=%lld phase2=%lld dpre=%lld phase3=%lld | p1 stage: issue=%lld mfma=%lld wait=%lld barrier=%lld | "
                        "gate stage: issue=%lld mfma=%lld elementwise=%lld stores=%lld wait=%lld barrier=%lld contraction stage=%lld | p3 stage: issue=%lld mfma=%lld rest=%lld\n",
                d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(5, 6), d(8, 9), d(9, 10), d(10, 11), d(11, 12),
                d(16, 17), d(17, 18), d(18, 19), d(19, 20), d(20, 21), d(21, 22), d(22, 24), d(32, 33), d(33, 34), d(34, 35));
    }
#endif
    return hipGetLastError();
}

template <typename IO, int RT, bool GATE, bool ACT_ID, bool DROP>
static hipError_t launch_waves(const PetBwdArgs& a, hipStream_t stream) {
    if constexpr (BwdLds<IO, RT, GATE, 4>::BIAS_OFF + 8 * 1024 <= 160 * 1024) {
        switch (pick_row_groups(a.M, 4, 2)) {       // rows per workgroup by rounds of 256 workgroups (kernels.h)
            case 4: return launch_one<IO, RT, GATE, ACT_ID, DROP, 4>(a, stream);
            case 3: return launch_one<IO, RT, GATE, ACT_ID, DROP, 3>(a, stream);
            default: return launch_one<IO, RT, GATE, ACT_ID, DROP, 2>(a, stream);
        }
    } else {
        return launch_one<IO, RT, GATE, ACT_ID, DROP, 2>(a, stream);
    }
}

template <typename IO, int RT>
static hipError_t launch_rt(const PetBwdArgs& a, hipStream_t stream) {
    const bool gate = a.flags & PET_GATE, act_id = a.flags & PET_ACT_IDENTITY, drop = drop_active(a.drop);
    if (gate) return launch_waves<IO, RT, true, false, false>(a, stream);
    if (act_id) return drop ? launch_waves<IO, RT, false, true, true>(a, stream)
                            : launch_waves<IO, RT, false, true, false>(a, stream);
    return launch_waves<IO, RT, false, false, false>(a, stream);
}

template <typename IO>
static hipError_t launch_io(const PetBwdArgs& a, hipStream_t stream) {
    switch (a.RT) {
        case 1: return launch_rt<IO, 1>(a, stream);
        case 3: return launch_rt<IO, 3>(a, stream);
        case 6: return launch_rt<IO, 6>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_pet_bwd(const PetBwdArgs& a, int io_fp32, hipStream_t stream) {
    return io_fp32 ? launch_io<float>(a, stream) : launch_io<__bf16>(a, stream);
}
