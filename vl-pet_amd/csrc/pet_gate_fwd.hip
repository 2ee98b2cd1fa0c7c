// K1 forward, chain-split form:  out = ( s2*x2 + sd*up_A(gelu_new(down_A(x2))) ) (*|+) sigmoid( up_G(gelu_new(down_G(x1))) ) * gs
// (my_transformers/modeling_bart.py:1147-1155, 1195-1209, 1256-1257).
//
// Why a second kernel: the one-wave-per-32-rows form (pet_fwd.hip) leaves one wave per SIMD (M/32 waves
// for 1024 SIMDs at the benchmark's M ~ 28k), and a lone wave serialises its own VALU epilogue, MFMAs, LDS
// reads and waits (rocprofv3: MFMA busy 17 %, VALU 32 %, waits 58 % of wave cycles).  Here the two
// projection chains of a 32-row group run on two different waves placed on the same SIMD:
//
//     wave rg      (chain A): x2 -> down_A -> gelu -> up_A, residual/scale, product with the gate, stores
//     wave rg + RG (chain G): x1 -> down_G -> gelu -> up_G, sigmoid  -> gate values to LDS
//
// so every SIMD holds two independent instruction streams with half the registers each; the chain-G wave
// runs the up phase one stage ahead and hands the gate tile over through a double-buffered LDS exchange
// (one s_barrier per stage, which the weight ring needs anyway).  Memory system as in pet_fwd.hip:
// everything arrives by global_load_lds as whole 128-byte lines, counted vmcnt waits, outputs leave as
// whole lines.  Stage t of the 2S+1 stages (S = d / features per stage):
//     t <  S : both chains, down projection of feature block t
//     t >= S : chain G, up projection + sigmoid of block t-S (t < 2S); chain A, up projection + epilogue of block t-S-1 (t > S)
// LDS: weight ring 2 x [A segment | G segment]; 96 KiB row area = 3 down-phase slots [x2 tile | x1 tile],
// re-used in the up phase as 3 residual tiles + 2 gate-exchange buffers (IO precision).
//
// Loader waves (template LD, one per row group, i.e. one per SIMD): with them the compute waves issue no
// global_load_lds at all -- a wave that waits for the memory pipe to take its next piece costs no issue slots --
// and the down phase runs at the HBM rate.  Rows per workgroup (RG = 2 / 4 row groups) are
// chosen per launch.  Training form (PetFwdArgs::save): after the activation each chain also stores z and
// gelu_new'(pre) for the backward, which then skips its recompute phase.
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "pet32.h"

template <typename IO, int RT, int RG>
struct GateLds {
    static constexpr int NS = Geo4<IO>::NS;
    static constexpr int SEG_KB = 4 * RT;
    static constexpr int SEG_FR = SEG_KB / NS;
    static constexpr int W_B = SEG_KB * 1024 * 2;
    static constexpr int TILE_B = RG * 32 * 128;
    static constexpr int ROW_OFF = 2 * W_B;
    static constexpr int BIAS_OFF = ROW_OFF + 6 * TILE_B;      // row area: 3 down-phase slots of [x2 tile | x1 tile]
    static size_t bytes(int d) { return (size_t)BIAS_OFF + (size_t)2 * (32 * RT + d) * 4; }
};

// LR = low-rank visual projector form (LowRankVisualEmbedding, src/modeling_bart.py:278-295): both chains read the same
// [M, d_in] feature rows (one LDS tile per stage instead of two), the down phase runs d_in / FE stages on the separate
// down-only packs, the up phase d / FE stages; no residual term; the gate value is sigmoid(.) + go.  Loader-wave form only.
template <typename IO, int RT, bool GATE_ADD, int RG, bool LD, bool LR = false>
__global__ __launch_bounds__((LD ? 3 : 2) * RG * 64) void pet_gate_fwd_kernel(PetFwdArgs a) {
    static_assert(!LR || LD, "the low-rank projector form exists with loader waves only");
    using G = Geo4<IO>;
    using L = GateLds<IO, RT, RG>;
    constexpr int NS = G::NS;
    constexpr int KT = 2 * RT;
    constexpr int NW = 2 * RG;                   // waves
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar registers and scalar branches for everything derived from it
    const int chain = wave / RG, rg = wave % RG;             // waves rg and rg + RG share a SIMD (RG = 4)
    const bool isA = chain == 0;
    const int m = lane & 31, h = lane >> 5;
    const int trow = 32 * rg + m;
    const int d = a.d;
    const int64_t row0_wave = (int64_t)blockIdx.x * (RG * 32) + rg * 32;
    const int SU = d / G::FE;                                // up-phase stages (output feature blocks)
    const int S = LR ? a.d_in / G::FE : SU;                  // down-phase stages (input feature blocks)
    const PackGeom pg = pack_geom(RT, d, NS);
    const uint8_t* pkA = a.pk_a;
    const uint8_t* pkG = a.pk_g;
    const uint8_t* pkAd = LR ? a.pk_a_dn : a.pk_a;           // down-phase weights (LR: down-only packs of width d_in)
    const uint8_t* pkGd = LR ? a.pk_g_dn : a.pk_g;
    const uint8_t* xin = reinterpret_cast<const uint8_t*>(isA ? a.xa : a.xg);
    const uint8_t* res = reinterpret_cast<const uint8_t*>(a.res);
    uint8_t* out = reinterpret_cast<uint8_t*>(a.out);

    auto slot_w = [&](int j) { return smem + (size_t)j * L::W_B; };
    auto slot_d = [&](int j) { return smem + L::ROW_OFF + (size_t)j * 2 * L::TILE_B + ((isA || LR) ? 0 : L::TILE_B); };
    // up-phase use of the row area, placed by when each down-phase slot was last read (slot (S-1)%3 at stage S-1,
    // (S-2)%3 at S-2, S%3 at S-3): three residual tiles (block b in tile b%3) in slots S%3, S%3, (S+1)%3 and the two
    // gate-exchange buffers (block b in buffer b&1, 16 bytes per lane and piece: 8 bf16 or 4 fp32 values) in the second
    // half of slot (S+1)%3 and the first half of slot (S+2)%3
    auto region = [&](int r) { return smem + L::ROW_OFF + (size_t)(r % 3) * 2 * L::TILE_B; };
    auto slot_res = [&](int j) { return j == 0 ? region(S) : j == 1 ? region(S) + L::TILE_B : region(S + 1); };
    auto slot_x = [&](int j) { return (j == 0 ? region(S + 1) + L::TILE_B : region(S + 2)) + (size_t)rg * 4096; };
    float* sb = reinterpret_cast<float*>(smem + L::BIAS_OFF);
    const int nb = 32 * RT + d;

    const RowLanes rl = row_lanes<IO>(row0_wave, a.M, d, rg, lane);
    const int lane16 = lane * 16;
    constexpr bool no_res = LR;          // compile-time: a runtime test here changes the register allocation of the K1 builds

    // ---- weight pieces of stage t (1 KiB each, piece k of [A segment | G segment]); returns how many this wave issued
    auto issue_w = [&](int t) -> int {
        if (LD || t > S + SU) return 0;
        uint8_t* dst = slot_w(t & 1);
        int n = 0;
        for (int k = wave; k < 2 * L::SEG_KB; k += NW) {
            const bool segA = k < L::SEG_KB;
            const int kk = segA ? k : k - L::SEG_KB;
            int64_t woff;
            if (t < S) woff = (int64_t)t * L::SEG_KB * 1024;
            else {
                const int su = segA ? t - S - 1 : t - S;
                if (su < 0 || su >= SU) continue;
                woff = pg.pack_bytes + (int64_t)su * L::SEG_KB * 1024;
            }
            glds16((segA ? pkA : pkG) + woff + (size_t)kk * 1024 + lane16, dst + (size_t)k * 1024);
            ++n;
        }
        return n;
    };
    // ---- row pieces issued during stage t: down rows two stages ahead; chain A's residual rows one stage ahead
    auto issue_rows = [&](int t) -> int {
        if (LD) return 0;
        if (t + 2 < S) {
            glds_rows4(xin, rl, (t + 2) * 128, slot_d((t + 2) % 3), rg);
            return 4;
        }
        if (isA && !no_res && t >= S && t < S + SU) {        // residual block su = t - S, consumed at stage t + 1
            const int su = t - S;
            glds_rows4(res, rl, su * 128, slot_res(su % 3), rg);
            return 4;
        }
        return 0;
    };

    // debug timestamps (build with -DVLPET_STAMPS, run with VLPET_DBG & 16; lane 0 of the first chain-A wave -- or, with
    // VLPET_DBG & 128, the first chain-G wave -- of every block).  Compiled out by default: an s_memtime anywhere in a
    // loop makes hipcc fall back to s_waitcnt lgkmcnt(0) for every LDS read in it (scalar-memory returns are unordered).
    auto stamp = [&](int k) {
#ifdef VLPET_STAMPS
        if ((a.dbg & 16) && tid == ((a.dbg & 128) ? RG * 64 : 0) && blockIdx.x < 4096) a.dbg_ts[blockIdx.x * 8 + k] = __builtin_readcyclecounter();
#else
        (void)k;
#endif
    };
    if constexpr (LD) {
        // ---- loader waves (one per row group, so one per SIMD): every global_load_lds of the workgroup.  A wave that is
        // waiting for the memory pipe to accept its next piece costs no issue slots, so the compute waves never stall on
        // memory issue; what must have landed at which barrier is the same protocol as without loaders.
        if (wave >= NW) {
            const uint8_t* x2p = reinterpret_cast<const uint8_t*>(a.xa);
            const uint8_t* x1p = reinterpret_cast<const uint8_t*>(a.xg);
            // LR: the input rows are d_in wide (rl addresses the d-wide output / residual rows)
            RowLanes rli_lr;
            if constexpr (LR) rli_lr = row_lanes<IO>(row0_wave, a.M, a.d_in, rg, lane);
            const RowLanes& rli = LR ? rli_lr : rl;
            auto ld_w = [&](int t) {
                if (t > S + SU) return;
                uint8_t* dst = slot_w(t & 1);
                for (int k = rg; k < 2 * L::SEG_KB; k += RG) {
                    const bool segA = k < L::SEG_KB;
                    const int kk = segA ? k : k - L::SEG_KB;
                    const uint8_t* src;
                    if (t < S) src = (segA ? pkAd : pkGd) + (int64_t)t * L::SEG_KB * 1024;
                    else {
                        const int su = segA ? t - S - 1 : t - S;
                        if (su < 0 || su >= SU) continue;
                        src = (segA ? pkA : pkG) + pg.pack_bytes + (int64_t)su * L::SEG_KB * 1024;
                    }
                    glds16(src + (size_t)kk * 1024 + lane16, dst + (size_t)k * 1024);
                }
            };
            auto ld_rows = [&](int t2) {        // down-phase rows of stage t2, both chains of this row group (LR: one shared tile)
                uint8_t* base = smem + L::ROW_OFF + (size_t)(t2 % 3) * 2 * L::TILE_B;
                glds_rows4(x2p, rli, t2 * 128, base, rg);
                if constexpr (!LR) glds_rows4(x1p, rli, t2 * 128, base + L::TILE_B, rg);
            };
            ld_w(0);
            ld_rows(0);
            if (S > 1) ld_rows(1);
            __syncthreads();
            // residual block b goes out at stage S+b-1 and is consumed at S+b+1: like the down-phase rows, the newest
            // row pieces stay in flight across the barrier (only the weights have a prefetch distance of one stage)
            for (int t = 0; t <= S + SU; ++t) {
                ld_w(t + 1);
                int nrows = 0;
                const int b = t - S + 1;
                if (t + 2 < S) { ld_rows(t + 2); nrows = LR ? 4 : 8; }
                else if (!no_res && b >= 0 && b < SU) { glds_rows4(res, rl, b * 128, slot_res(b % 3), rg); nrows = 4; }
                wait_vm(nrows);
                __builtin_amdgcn_s_barrier();
            }
            return;
        }
    }
    stamp(0);
    if constexpr (!LD) {
        issue_w(0);
        glds_rows4(xin, rl, 0, slot_d(0), rg);
        if (S > 1) glds_rows4(xin, rl, 128, slot_d(1), rg);
    }
    {   // biases -> LDS: [bdA(32RT) | buA(d) | bdG(32RT) | buG(d)]
        copy_bias<NW * 64>(sb, reinterpret_cast<const float*>(a.pk_a + pg.bias_off), nb, tid);
        copy_bias<NW * 64>(sb + nb, reinterpret_cast<const float*>(a.pk_g + pg.bias_off), nb, tid);
    }
    __syncthreads();
    stamp(1);

    // ---- down projection of this wave's chain: register 8*sh + j of c-tile ct <-> c = 32ct + 16sh + 8h + j
    f32x16 acc[RT];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) acc[ct] = zero16();
    int t = 0;
    for (; t < S; ++t) {
        if (t == 5 && !(a.dbg & 32)) stamp(5);
        issue_w(t + 1);
        const int nrows = issue_rows(t);
        const uint8_t* w = slot_w(t & 1) + (isA ? 0 : L::SEG_KB * 1024);
        const uint8_t* tile = slot_d(t % 3);
        // all LDS fragment reads of the stage first, then the MFMAs behind counted lgkmcnt waits: with two waves per
        // SIMD a read -> wait -> MFMA chain per fragment leaves the matrix pipe idle for the LDS latency every time
        if constexpr (G::KU * RT <= 12) {
            Frag<NS> bf[G::KU], wf[G::KU * RT];
#pragma unroll
            for (int u = 0; u < G::KU; ++u) {
                bf[u] = tile_bfrag4<IO>(tile, trow, h, u);
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) wf[u * RT + ct] = wfrag<NS>(w, u * RT + ct, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < G::KU; ++u) {
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) acc[ct] = mfma_ns<NS>(wf[u * RT + ct], bf[u], acc[ct]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < G::KU; ++u) {
                const Frag<NS> b = tile_bfrag4<IO>(tile, trow, h, u);
                Frag<NS> wf[RT];
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) wf[ct] = wfrag<NS>(w, u * RT + ct, lane);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) acc[ct] = mfma_ns<NS>(wf[ct], b, acc[ct]);
            }
        }
        if (t == 5 && !(a.dbg & 32)) stamp(6);
        wait_vm(nrows);
        __builtin_amdgcn_s_barrier();
        if (t == 5 && !(a.dbg & 32)) stamp(7);
    }
    stamp(2);

    // ---- bias + gelu_new -> B fragments of the up projection (k-step 2ct+sh holds c = 32ct+16sh+8h+j)
    Frag<NS> z[KT];
    {
        const float* bd = sb + (isA ? 0 : nb) + 8 * h;
        // training: leave z and gelu'(pre) of this chain for the backward ([M, 32RT] each, IO dtype), which then
        // neither recomputes the down projections nor re-reads x1 / x2 for them
        const bool save = a.save != nullptr && row0_wave + m < a.M;
        IO* sv_z = reinterpret_cast<IO*>(reinterpret_cast<uint8_t*>(a.save) + (isA ? 0 : 2) * a.save_stride) +
                   (row0_wave + m) * (int64_t)(32 * RT) + 8 * h;
        IO* sv_g = reinterpret_cast<IO*>(reinterpret_cast<uint8_t*>(a.save) + (isA ? 1 : 3) * a.save_stride) +
                   (row0_wave + m) * (int64_t)(32 * RT) + 8 * h;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) {
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = acc[ct][8 * sh + j] + bd[32 * ct + 16 * sh + j];
                if (save) {
                    float g[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] = gelu_new_grad_f(v[j]);
                    store8_f32(sv_g + 32 * ct + 16 * sh, g);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = gelu_new_f(v[j]);
                if (save) store8_f32(sv_z + 32 * ct + 16 * sh, v);
                z[2 * ct + sh] = frag_from_f32<NS>(v);
            }
        }
    }

    stamp(3);
    // ---- up phase
    const float* bu = sb + (isA ? 0 : nb) + 32 * RT + G::LW * h;
    const float gs = a.gs;
    const float s2g = a.s2 * gs, sdg = a.sd * gs;      // gate scale folded into the linear part
    float gm = 1.f, go = 0.f;
    if constexpr (LR) { gm = a.gm; go = a.go; }
    for (; t <= S + SU; ++t) {
        if (t == S + 5 && (a.dbg & 32)) stamp(5);      // VLPET_DBG & 32: stage stamps from the up phase instead
        issue_w(t + 1);
        const int nrows = issue_rows(t);
        (void)nrows;
        const int su = isA ? t - S - 1 : t - S;
        int n_after = 0;                                // vector-memory operations allowed to stay in flight
        if (su >= 0 && su < SU) {
            const uint8_t* w = slot_w(t & 1) + (isA ? 0 : L::SEG_KB * 1024);
            f32x16 au[G::NV];
#pragma unroll
            for (int v = 0; v < G::NV; ++v) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 tb = *reinterpret_cast<const f32x4*>(bu + su * G::FE + 16 * v + 4 * q);
                    au[v][4 * q] = tb[0]; au[v][4 * q + 1] = tb[1]; au[v][4 * q + 2] = tb[2]; au[v][4 * q + 3] = tb[3];
                }
            }
            if constexpr (G::NV * KT <= 12) {
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
            } else {
#pragma unroll
                for (int ks = 0; ks < KT; ++ks) {
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) au[v] = mfma_ns<NS>(wfrag<NS>(w, v * KT + ks, lane), z[ks], au[v]);
                }
            }
            constexpr int XE = 16 / (int)sizeof(IO);        // gate values per 16-byte exchange piece (IO precision)
            if (!isA) {
                // gate values -> exchange buffer of this block; piece p of the lane at p*1 KiB + lane*16
                uint8_t* xb = slot_x(su & 1);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float g[XE];
#pragma unroll
                    for (int e = 0; e < XE; ++e) { const int i = XE * p + e; g[e] = sigmoid_f(au[i >> 4][i & 15]); }
                    if constexpr (NS == 1) {
                        bf16x8 pk;
#pragma unroll
                        for (int e = 0; e < 8; ++e) pk[e] = (__bf16)g[e];
                        *reinterpret_cast<bf16x8*>(xb + (size_t)p * 1024 + lane16) = pk;
                    } else {
                        const f32x4 pk = {g[0], g[1], g[2], g[3]};
                        *reinterpret_cast<f32x4*>(xb + (size_t)p * 1024 + lane16) = pk;
                    }
                }
            } else {
                uint8_t* tr = slot_res(su % 3);
                const uint8_t* xb = slot_x(su & 1);
                float r[G::LW], o[G::LW], gv[G::LW];
                if constexpr (no_res) {
#pragma unroll
                    for (int i = 0; i < G::LW; ++i) r[i] = 0.f;       // the tile is only the staging area of the output here
                } else {
                    tile_lane_vals4<IO>(tr, trow, h, r);
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if constexpr (NS == 1) {
                        const bf16x8 pk = *reinterpret_cast<const bf16x8*>(xb + (size_t)p * 1024 + lane16);
#pragma unroll
                        for (int e = 0; e < 8; ++e) gv[8 * p + e] = (float)pk[e];
                    } else {
                        const f32x4 pk = *reinterpret_cast<const f32x4*>(xb + (size_t)p * 1024 + lane16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) gv[4 * p + e] = pk[e];
                    }
                }
#pragma unroll
                for (int i = 0; i < G::LW; ++i) {
                    const float lin = s2g * r[i] + sdg * au[i >> 4][i & 15];
                    if constexpr (LR) o[i] = lin * (gm * gv[i] + go);  // scale / offset applied in fp32, after the IO-precision exchange
                    else o[i] = GATE_ADD ? lin + gs * gv[i] : lin * gv[i];
                }
                stage_lane_vals4<IO>(tr, trow, h, o);
                store_rows4(out, rl, su * 128, tr, rg, lane);
                n_after = rl.n_inst;
            }
        }
        if (t == S + 5 && (a.dbg & 32)) stamp(6);
        wait_vm(n_after);
        __builtin_amdgcn_s_barrier();
        if (t == S + 5 && (a.dbg & 32)) stamp(7);
    }
    stamp(4);
}

template <typename IO, int RT, bool GATE_ADD, int RG, bool LD, bool LR = false>
static hipError_t launch_one(const PetFwdArgs& a, hipStream_t stream) {
    using L = GateLds<IO, RT, RG>;
    const size_t lds = L::bytes(a.d);
    auto kern = pet_gate_fwd_kernel<IO, RT, GATE_ADD, RG, LD, LR>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int rows = RG * 32;
    const int blocks = (int)((a.M + rows - 1) / rows);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3((LD ? 3 : 2) * RG * 64), lds, stream, a);
    return hipGetLastError();
}

template <typename IO, int RT>
static hipError_t launch_rt(const PetFwdArgs& a, hipStream_t stream) {
    const bool add = a.flags & PET_GATE_ADD;
    // up to 4 row groups (128 rows) with loader waves unless the rings would not fit the 160 KiB LDS
    if constexpr (GateLds<IO, RT, 4>::BIAS_OFF + 8 * 1024 <= 160 * 1024) {
        // forward: 128-row workgroups unless 64-row ones still fit one round of 256 (measured at M = 15 k / 32 k / 47 k:
        // 96-row workgroups -- 9 waves, uneven over the 4 SIMDs -- are the slowest form at every size, and two rounds of
        // 128 rows beat three of 64); the backward rows kernel follows pick_row_groups' cost model
        int rg = a.M <= 256 * 64 ? 2 : 4;
        const int rg_env = vlpet_tuning().rg;     // (A/B override: diagnosis builds only)
        if (rg_env >= 2 && rg_env <= 4) rg = rg_env;
        switch (rg) {
            case 4: return add ? launch_one<IO, RT, true, 4, true>(a, stream) : launch_one<IO, RT, false, 4, true>(a, stream);
            case 3: return add ? launch_one<IO, RT, true, 3, true>(a, stream) : launch_one<IO, RT, false, 3, true>(a, stream);
            default: return add ? launch_one<IO, RT, true, 2, true>(a, stream) : launch_one<IO, RT, false, 2, true>(a, stream);
        }
    } else {
        return add ? launch_one<IO, RT, true, 2, false>(a, stream) : launch_one<IO, RT, false, 2, false>(a, stream);
    }
}

template <typename IO>
static hipError_t launch_io(const PetFwdArgs& a, hipStream_t stream) {
    switch (a.RT) {
        case 1: return launch_rt<IO, 1>(a, stream);
        case 3: return launch_rt<IO, 3>(a, stream);
        case 6: return launch_rt<IO, 6>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_pet_gate_fwd(const PetFwdArgs& a, int io_fp32, hipStream_t stream) {
    return io_fp32 ? launch_io<float>(a, stream) : launch_io<__bf16>(a, stream);
}

// low-rank visual projector form (PetFwdArgs::d_in, pk_a_dn, pk_g_dn, go): r, r_g <= 96, multiplicative gate
template <typename IO, int RT>
static hipError_t launch_lr(const PetFwdArgs& a, hipStream_t stream) {
    static_assert(GateLds<IO, RT, 4>::BIAS_OFF + 8 * 1024 <= 160 * 1024, "loader-wave form must fit");
    return a.M <= 256 * 64 ? launch_one<IO, RT, false, 2, true, true>(a, stream)
                           : launch_one<IO, RT, false, 4, true, true>(a, stream);
}
hipError_t launch_pet_lowrank_fwd(const PetFwdArgs& a, int io_fp32, hipStream_t stream) {
    if (a.d_in <= 0 || !a.pk_a_dn || !a.pk_g_dn || (a.flags & PET_GATE_ADD)) return hipErrorInvalidValue;
    if (a.RT == 1) return io_fp32 ? launch_lr<float, 1>(a, stream) : launch_lr<__bf16, 1>(a, stream);
    if (a.RT == 3) return io_fp32 ? launch_lr<float, 3>(a, stream) : launch_lr<__bf16, 3>(a, stream);
    return hipErrorInvalidValue;
}
