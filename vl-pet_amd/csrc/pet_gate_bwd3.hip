// K1 backward in two passes that move each [M, d] tensor as few times as the op allows (gated K1 with the forward's
// saved activations, r <= 96).  Autograd of my_transformers/modeling_bart.py:1147-1155,1195-1209.
//
// The previous form (pet_gate_bwd2.hip + wgrad.hip) moved ~11 units of d*M*b for 5 algorithmic: the row kernel wrote the
// side products dh, dq, re-read dh for its last phase, and the column-parallel weight-gradient kernel then read x1, x2, dh
// and dq again.  The structure of the op forces two passes (dz = Wu^T dh needs EVERY feature of a row before any input
// gradient of that row exists, and the four [r x d] weight gradients are sums over EVERY row), so here the passes are
//
//   pass 1  pet_gate_dz_kernel (row-parallel; the middle phase of pet_gate_bwd2_kernel and nothing else):
//           reads dy, x2 (+ the saved z, gelu');  recomputes both up projections, dh, dq tile by tile, contracts
//           dz_a += Wu^T dh, dz_g += Wgu^T dq in registers;  writes only dpre_a, dpre_g  [M, 32*RT].
//   pass 2  pet_gate_cols_kernel (column-parallel: a workgroup owns one feature block of 128 bytes per row = one stage
//           of the packs, and a chunk of rows): reads dy, x1, x2 (+ z, dpre of its rows, L2-resident across the
//           feature blocks), recomputes ITS block of dh, dq, and from them and dpre produces everything that is left:
//               dx2 = s2*dh + Wd^T dpre_a      dx1 = Wgd^T dpre_g                     (whole 128-byte lines per row)
//               dWd += dpre_a^T x2   dWu += z_a^T dh   dWgd += dpre_g^T x1   dWgu += z_g^T dq   (+ the four bias sums)
//           Row-chunk partials go to the workspace in wgrad.hip's layout and wgrad_finalize_kernel sums them
//           (deterministic, no atomics).
//
// HBM traffic: pass 1 reads 2 units, pass 2 reads 3 and writes 2  ->  7 units + the [M, 32*RT] tensors + the partials.
//
// Pass 2, one workgroup = 8 waves = 2 row groups (32 rows each) x 4 roles, every role owning one weight gradient
// (96 accumulator registers at r = 96) and one quarter of the rest:
//     A1: loads x2 tile, dpre_a;  dWd (transposes + products);  dx2 = s2*dh + Wd^T dpre_a, stores
//     A2: loads z_a;  a_A = bu + Wu z_a, h = s2*x2 + sd*a_A -> exchange;  dWu from the dh tile
//     G1: loads x1 tile, dpre_g;  dWgd;  dx1 = Wgd^T dpre_g, stores
//     G2: loads dy tile, z_g;  g = sigmoid(bgu + Wgu z_g);  dh, dq from h, dy, g -> tiles;  dWgu from the dq tile
// (waves w and w + 4 share a SIMD: A1/G1 and A2/G2 of a row group).  The m-contractions use wgrad.hip's operand
// construction: natural fragments (lane = row) times an identity fragment, the C/D layout being the transpose.
// Three barriers per 64-row iteration; row tiles arrive by global_load_lds one iteration ahead.
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "pet32.h"

// ================================================================================================== pass 1
template <typename IO, int RT, int RG>
struct DzLds {
    static constexpr int NS = Geo4<IO>::NS;
    static constexpr int SEG_KB = 4 * RT;
    static constexpr int W_B = SEG_KB * 1024 * 2;
    static constexpr int TILE_B = RG * 32 * 128;
    static constexpr int ROW_B = 2 * TILE_B;
    static constexpr int ROW_OFF = 2 * W_B;
    static constexpr int BIAS_OFF = ROW_OFF + 3 * ROW_B;          // two row slots + the fp32 exchange buffer
    static size_t bytes(int d) { return (size_t)BIAS_OFF + (size_t)2 * (32 * RT + d) * 4; }
};

template <bool B> struct Bool3 { static constexpr bool value = B; };

template <typename IO, int RT, int RG>
__global__ __launch_bounds__(RG * 128) void pet_gate_dz_kernel(PetBwdArgs a) {
    using G = Geo4<IO>;
    using L = DzLds<IO, RT, RG>;
    constexpr int NS = G::NS;
    constexpr int KT = 2 * RT;
    constexpr int PW = L::SEG_KB / RG;
    static_assert(L::SEG_KB % RG == 0, "weight segment must split evenly over the row groups");
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
    const uint8_t* pk = isA ? a.pk_a : a.pk_g;
    const uint8_t* res = reinterpret_cast<const uint8_t*>(a.res);
    const uint8_t* dy = reinterpret_cast<const uint8_t*>(a.dy);

    auto slot_w = [&](int j) { return smem + (size_t)j * L::W_B + (isA ? 0 : L::SEG_KB * 1024); };
    auto slot_t0 = [&](int j) { return smem + L::ROW_OFF + (size_t)j * L::ROW_B; };
    auto slot_t1 = [&](int j) { return smem + L::ROW_OFF + (size_t)j * L::ROW_B + L::TILE_B; };
    constexpr int XH = G::LW / 2;
    uint8_t* xbuf = smem + L::ROW_OFF + (size_t)2 * L::ROW_B;
    auto xslot = [&](int writer_chain) { return xbuf + (size_t)writer_chain * L::TILE_B + (size_t)rg * (XH * 256); };
    float* sb = reinterpret_cast<float*>(smem + L::BIAS_OFF);
    const int nb = 32 * RT + d;
    const int NST = 2 * S;                          // stages: a, b alternating

    const RowLanes rl = row_lanes<IO>(row0_wave, a.M, d, rg, lane);
    const int lane16 = lane * 16;
    const uint32_t wv_off = (uint32_t)(rg * 1024 + lane16);

    // weight pieces of stage st (own segment): a -> pack 1 (up), b -> pack 2 (up_t)
    auto issue_w = [&](int st) {
        if (st >= NST) return;
        const int ss = st >> 1, pack = (st & 1) ? 2 : 1;
        const uint8_t* src = pk + (int64_t)pack * pg.pack_bytes + (int64_t)ss * L::SEG_KB * 1024;
        uint8_t* dst = slot_w(st & 1) + rg * 1024;
#pragma unroll
        for (int j = 0; j < PW; ++j) glds16(src + (wv_off + j * RG * 1024), dst + j * RG * 1024);
    };
    auto issue_rows = [&](int su) {
        if (su >= S) return 0;
        if (isA) glds_rows4(res, rl, su * 128, slot_t0(su & 1), rg);
        else glds_rows4(dy, rl, su * 128, slot_t1(su & 1), rg);
        return 4;
    };

    issue_w(0);
    issue_rows(0);
    copy_bias<RG * 128>(sb, reinterpret_cast<const float*>(a.pk_a + pg.bias_off), nb, tid);
    copy_bias<RG * 128>(sb + nb, reinterpret_cast<const float*>(a.pk_g + pg.bias_off), nb, tid);
    Frag<NS> z[KT];
    const uint8_t* sv = reinterpret_cast<const uint8_t*>(a.saved) + (isA ? 0 : 2) * a.saved_stride;
    const int64_t ro = grow * (int64_t)(32 * RT) + 8 * h;
    {
        const IO* sz = reinterpret_cast<const IO*>(sv) + ro;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            float v[8];
            load8_f32(sz + 16 * ks, v);
            z[ks] = frag_from_f32<NS>(v);
        }
    }
    __syncthreads();
    constexpr bool PIPE = RT > 3;                   // r = 192: fragment reads one k-step ahead instead of a stage's worth up front

    const float* bu = sb + (isA ? 0 : nb) + 32 * RT + G::LW * h;
    const float s2 = a.s2, sd_ = a.sd, gs = a.gs;
    const bool gate_add = (a.flags & PET_GATE_ADD) != 0;
    f32x16 dz[RT];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) dz[ct] = zero16();
    constexpr int EH = G::E4 / 2;
    const int e0 = isA ? 0 : EH;
    int st = 0;
    for (int su = 0; su < S; ++su) {
        // ---- stage a: this chain's up projection, hand the partner its half
        f32x16 au[G::NV];
        {
            issue_w(st + 1);
            const int nrows = issue_rows(su + 1);
            const uint8_t* w = slot_w(st & 1);
#pragma unroll
            for (int v = 0; v < G::NV; ++v) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 tb = *reinterpret_cast<const f32x4*>(bu + su * G::FE + 16 * v + 4 * q);
                    au[v][4 * q] = tb[0]; au[v][4 * q + 1] = tb[1]; au[v][4 * q + 2] = tb[2]; au[v][4 * q + 3] = tb[3];
                }
            }
            if constexpr (!PIPE) {
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
                Frag<NS> cur[G::NV], nxt[G::NV];
#pragma unroll
                for (int v = 0; v < G::NV; ++v) cur[v] = wfrag<NS>(w, v * KT, lane);
#pragma unroll
                for (int ks = 0; ks < KT; ++ks) {
                    if (ks + 1 < KT) {
#pragma unroll
                        for (int v = 0; v < G::NV; ++v) nxt[v] = wfrag<NS>(w, v * KT + ks + 1, lane);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) au[v] = mfma_ns<NS>(cur[v], z[ks], au[v]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + 1 < KT) {
#pragma unroll
                        for (int v = 0; v < G::NV; ++v) cur[v] = nxt[v];
                    }
                }
            }
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
            wait_vm(nrows);
            __builtin_amdgcn_s_barrier();
            ++st;
        }
        // ---- stage b: elementwise backward of the own half, then the contraction over this block's features
        {
            issue_w(st + 1);
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
                    tile_lane_vals8<IO>(t0, trow, h, e, r8);
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
                            const float hv = s2 * r8[j] + sd_ * aAv;
                            dh8[j] = dyp * gt;
                            dq8[j] = dh8[j] * hv * (1.0f - gt);
                        }
                    }
                    stage_lane_vals8<IO>(t0, trow, h, e, dh8);
                    stage_lane_vals8<IO>(t1, trow, h, e, dq8);
                }
            };
            if (gate_add) elementwise(Bool3<true>{}); else elementwise(Bool3<false>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // both halves of the dh / dq tiles of this row group are complete
            const uint8_t* mine = isA ? t0 : t1;
            if constexpr (!PIPE) {
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
            } else {
#pragma unroll
                for (int e = 0; e < G::E4; ++e) {
                    float v8[8];
                    tile_lane_vals8<IO>(mine, trow, h, e, v8);
                    const Frag<NS> dfe = frag_from_f32<NS>(v8);
                    Frag<NS> wf[RT];
#pragma unroll
                    for (int ct = 0; ct < RT; ++ct) wf[ct] = wfrag<NS>(w, e * RT + ct, lane);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ct = 0; ct < RT; ++ct) dz[ct] = mfma_ns<NS>(wf[ct], dfe, dz[ct]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            wait_vm(0);                             // the next stage's weights (and the rows issued a stage ago) have landed
            __builtin_amdgcn_s_barrier();
            ++st;
        }
    }
    // ---- dpre = dz * act'(pre)  ->  [M, 32*RT] (IO dtype), the only output of this pass
    {
        const int ldz = 32 * RT;
        IO* dps = reinterpret_cast<IO*>(isA ? a.dp_a : a.dp_g);
        const IO* sg = reinterpret_cast<const IO*>(sv + a.saved_stride) + ro;      // act'(pre), saved by the forward
        const float sc = isA ? sd_ : 1.0f;          // dz_a = sd * Wu^T dh: the delta scale once, here
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) {
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                float v[8], gp[8];
                load8_f32(sg + 32 * ct + 16 * sh, gp);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = sc * dz[ct][8 * sh + j] * gp[j];
                if (row_ok) store8_f32(dps + grow * ldz + 32 * ct + 16 * sh + 8 * h, v);
            }
        }
    }
}

// ================================================================================================== pass 2
struct ColsArgs {
    const void* dy; const void* x1; const void* x2;
    const void* z_a; const void* z_g;       // the forward's saved z  [M, 32*RT]
    const void* dp_a; const void* dp_g;     // pass 1's dpre         [M, 32*RT]
    void* dx1; void* dx2;
    const uint8_t* pk_a; const uint8_t* pk_g;
    int64_t M;
    int d, RT;
    float s2, sd, gs;
    int flags;
    int GS, NG;                             // feature blocks per XCD group, groups per row chunk (S = GS * NG)
    WgradArgs wg;                           // partial layout (job 0 dWd, 1 dWu, 2 dWgd, 3 dWgu), row_chunks, rows_per_chunk
};

template <typename IO, int NRG, int NBUF_>
struct ColsLds {
    using G = Geo4<IO>;
    static constexpr int NBUF = NBUF_;                              // input tile buffers (NBUF - 1 iterations ahead)
    static constexpr int TILE_B = 32 * 128;
    static constexpr int XH_B = 32 * G::FE * 4;                     // fp32 exchange of h: LW values per lane
    static constexpr int RG_B = (3 * NBUF + 2) * TILE_B + XH_B;     // x2, dy, x1 buffers; dh, dq tiles; exchange
    static int w_bytes(int RT) { return 4 * 4 * RT * 1024; }        // [up A | up G | down_t A | down_t G] of one stage
    static size_t bytes(int RT) {
        const size_t main = (size_t)w_bytes(RT) + NRG * RG_B + 2 * G::FE * 4;
        const size_t red = NRG > 1 ? (size_t)4 * (RT * G::NV * 16 + RT + G::NV) * 64 * 4 : 0;     // end-of-kernel reduction over the row groups
        return main > red ? main : red;
    }
};

template <int NS>
__device__ __forceinline__ f32x16 transpose32(const Frag<NS>& lo16, const Frag<NS>& hi16, bf16x8 I0, bf16x8 I1) {
    f32x16 t = zero16();
#pragma unroll
    for (int p = 0; p < NS; ++p) {
        t = mfma32(lo16.p[p], I0, t);
        t = mfma32(hi16.p[p], I1, t);
    }
    return t;
}

template <int NS>
__device__ __forceinline__ Frag<NS> zero_frag() {
    Frag<NS> f;
#pragma unroll
    for (int p = 0; p < NS; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) f.p[p][j] = (__bf16)0.0f;
    return f;
}

// NRG = row groups (32 rows) per workgroup: 2 for r <= 96 (8 waves, two per SIMD, 256 registers each); 1 for r = 192
// (4 waves, one per SIMD: the 192 accumulator registers of a role need the whole register file of a SIMD lane)
// NBUF = row-tile buffers (tiles are requested NBUF - 1 iterations ahead); PPF = the P rows of the next iteration are
// requested at the top of the current one into a second register set (needs the registers: NRG = 1)
template <typename IO, int RT, int NRG, int NBUF, bool PPF>
__global__ __launch_bounds__(NRG * 256) void pet_gate_cols_kernel(ColsArgs a) {
    using G = Geo4<IO>;
    using L = ColsLds<IO, NRG, NBUF>;
    constexpr int IR = 32 * NRG;                                     // rows per iteration
    constexpr int NS = G::NS, KT = 2 * RT, NV = G::NV, FE = G::FE, LW = G::LW;
    constexpr int SEG_B = 4 * RT * 1024;
    constexpr int PR = 32 * RT;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    // ---- which (feature block, row chunk): XCD-aware numbering (block b runs on XCD b % 8; the GS feature blocks of a
    // group read the same z / dpre rows and should share an L2 -- speed only, any placement is correct)
    const int b = blockIdx.x, xcd = b & 7, kq = b >> 3;
    const int sg = kq % a.GS, unit = (kq / a.GS) * 8 + xcd;
    const int rc = unit / a.NG, su = (unit % a.NG) * a.GS + sg;
    if (rc >= a.wg.row_chunks) return;                               // uniform per block, before any barrier
    const int64_t r_begin = (int64_t)rc * a.wg.rows_per_chunk;
    if (r_begin >= a.M) return;
    int64_t r_end = r_begin + a.wg.rows_per_chunk;
    if (r_end > a.M) r_end = a.M;
    const int n_it = (int)((r_end - r_begin + IR - 1) / IR);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave % NRG, role = wave / NRG;                    // role 0 A1, 1 A2, 2 G1, 3 G2
    const bool chainA = role < 2;
    const int m = lane & 31, h = lane >> 5;
    const int lane16 = lane * 16;
    const int d = a.d;
    const PackGeom pg = pack_geom(RT, d, NS);

    // ---- LDS map
    uint8_t* Wup = smem + (chainA ? 0 : SEG_B);                     // this chain's up fragments (v, ks) of stage su
    uint8_t* Wdt = smem + 2 * SEG_B + (chainA ? 0 : SEG_B);         // this chain's down_t fragments (v, ks)
    uint8_t* rgb = smem + 4 * SEG_B + (size_t)rg * L::RG_B;
    auto Tx2 = [&](int bf) { return rgb + (size_t)(0 * L::NBUF + bf) * L::TILE_B; };
    auto Tdy = [&](int bf) { return rgb + (size_t)(1 * L::NBUF + bf) * L::TILE_B; };
    auto Tx1 = [&](int bf) { return rgb + (size_t)(2 * L::NBUF + bf) * L::TILE_B; };
    uint8_t* Tdh = rgb + (size_t)(3 * L::NBUF) * L::TILE_B;
    uint8_t* Tdq = Tdh + L::TILE_B;
    uint8_t* XHb = Tdq + L::TILE_B;
    float* sbias = reinterpret_cast<float*>(smem + 4 * SEG_B + NRG * L::RG_B);   // [bu block (FE) | bgu block (FE)]

    // ---- prologue: this stage's weight fragments and bias blocks
    for (int k = wave; k < 16 * RT; k += 4 * NRG) {
        const int blk = k / (4 * RT), piece = k % (4 * RT);          // 0 up A, 1 up G, 2 down_t A, 3 down_t G
        const uint8_t* pkx = (blk & 1) ? a.pk_g : a.pk_a;
        const uint8_t* src = pkx + (int64_t)(blk < 2 ? 1 : 3) * pg.pack_bytes + (int64_t)su * SEG_B + (size_t)piece * 1024;
        glds16(src + lane16, smem + (size_t)blk * SEG_B + (size_t)piece * 1024);
    }
    if (tid < 2 * FE) {
        const uint8_t* pkx = tid < FE ? a.pk_a : a.pk_g;
        sbias[tid] = reinterpret_cast<const float*>(pkx + pg.bias_off)[PR + su * FE + (tid % FE)];
    }

    bf16x8 I0, I1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        I0[j] = (m == 8 * h + j) ? (__bf16)1.0f : (__bf16)0.0f;
        I1[j] = (m == 16 + 8 * h + j) ? (__bf16)1.0f : (__bf16)0.0f;
    }
    const uint8_t* Xsrc = reinterpret_cast<const uint8_t*>(role == 0 ? a.x2 : (role == 2 ? a.x1 : a.dy));   // role 1 loads no tile
    const IO* Psrc = reinterpret_cast<const IO*>(role == 0 ? a.dp_a : (role == 1 ? a.z_a : (role == 2 ? a.dp_g : a.z_g)));
    uint8_t* dxo = reinterpret_cast<uint8_t*>(role == 0 ? a.dx2 : a.dx1);
    const float s2 = a.s2, sd_ = a.sd, gs = a.gs;
    const bool gate_add = (a.flags & PET_GATE_ADD) != 0;

    f32x16 acc[RT][NV];
    float csp[RT], csx[NV];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) {
        csp[ct] = 0.f;
#pragma unroll
        for (int nt = 0; nt < NV; ++nt) acc[ct][nt] = zero16();
    }
#pragma unroll
    for (int nt = 0; nt < NV; ++nt) csx[nt] = 0.f;

    auto tile_of = [&](int bf) { return role == 0 ? Tx2(bf) : (role == 2 ? Tx1(bf) : Tdy(bf)); };
    // row-tile addressing: instruction i of a wave moves rows 8i + (lane >> 3), 16-byte piece (lane & 7) ^ swz(row) -- a
    // wave-uniform 64-bit base per block plus a 32-bit per-lane offset (full blocks; the ragged last block clamps per lane)
    uint32_t voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tr = 8 * i + (lane >> 3);
        voff[i] = (uint32_t)tr * (uint32_t)(d * (int)sizeof(IO)) + (uint32_t)(((lane & 7) ^ swz(tr)) * 16);
    }
    auto issue_tile = [&](int it) {                                  // this wave's input tile of iteration it
        if (role == 1 || it >= n_it) return 0;
        const int64_t row0t = r_begin + (int64_t)it * IR + 32 * rg;
        uint8_t* dst = tile_of(it % NBUF);
        if (row0t + 32 <= a.M) {
            const uint8_t* base = Xsrc + row0t * d * (int64_t)sizeof(IO) + su * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16_row(base + voff[i], dst + i * 1024);
        } else {
            const RowLanes rl = row_lanes<IO>(row0t, a.M, d, 0, lane);
            glds_rows4(Xsrc, rl, su * 128, dst, 0);
        }
        return 4;
    };
    // whole-line stores of a staged tile (the wave's 32 rows of feature block su)
    auto store_tile = [&](uint8_t* base_out, int64_t row0t, const uint8_t* tile) {
        if (row0t + 32 <= a.M) {
            uint8_t* base = base_out + row0t * d * (int64_t)sizeof(IO) + su * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(tile + ((size_t)(8 * i + (lane >> 3)) * 8 + (lane & 7)) * 16);
                *reinterpret_cast<u32x4*>(base + voff[i]) = v;
            }
            return 4;
        }
        const RowLanes rl = row_lanes<IO>(row0t, a.M, d, 0, lane);
        store_rows4(base_out, rl, su * 128, tile, 0, lane);
        return rl.n_inst;
    };
    Frag<NS> pn[KT];                                                 // this role's P rows, natural fragments
    Frag<NS> pnn[PPF ? KT : 1];                                      // ... of the next iteration (PPF)
    constexpr int NPL = KT * (int)(sizeof(IO) / 2);                  // global loads of one load_p
    auto load_p = [&](int it, Frag<NS>* dst) {                       // (rows past the end: clamped address, zeroed at use)
        int64_t row = r_begin + (int64_t)it * IR + 32 * rg + m;
        if (row >= a.M) row = a.M - 1;
        const IO* pr = Psrc + row * (int64_t)PR + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) dst[ks] = load_frag8(pr + 16 * ks);
    };
    // accumulate  Out[c, n] += sum_m P[m, c] X[m, n]  for this wave's 32 rows (P = pn, X = the NV natural-fragment pairs)
    // (scheduling fences between the sub-steps: without them hipcc interleaves all transposes and keeps every intermediate
    // alive at once -- the wave holds 96 accumulator registers and has 256 in all)
    auto accumulate = [&](const Frag<NS>* xn) {
        Frag<NS> xt[NV][2];
#pragma unroll
        for (int nt = 0; nt < NV; ++nt) {
            const f32x16 t = transpose32<NS>(xn[2 * nt], xn[2 * nt + 1], I0, I1);
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i] = t[i]; csx[nt] += t[i]; }
            xt[nt][0] = frag_from_f32<NS>(v);
            xt[nt][1] = frag_from_f32<NS>(v + 8);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) {
            const f32x16 t = transpose32<NS>(pn[2 * ct], pn[2 * ct + 1], I0, I1);
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i] = t[i]; csp[ct] += t[i]; }
            const Frag<NS> pt0 = frag_from_f32<NS>(v), pt1 = frag_from_f32<NS>(v + 8);
#pragma unroll
            for (int nt = 0; nt < NV; ++nt) {
                acc[ct][nt] = mfma_ns<NS>(pt0, xt[nt][0], acc[ct][nt]);
                acc[ct][nt] = mfma_ns<NS>(pt1, xt[nt][1], acc[ct][nt]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // X natural fragments of a row tile (rows past the chunk contribute nothing)
    auto tile_frags = [&](const uint8_t* tile, bool valid, Frag<NS>* xn) {
#pragma unroll
        for (int u = 0; u < G::KU; ++u) {
            const Frag<NS> f = tile_bfrag4<IO>(tile, m, h, u);
            xn[u] = valid ? f : zero_frag<NS>();
        }
    };
    // projection of this stage: out[v] = init + W(v, ks) . pn[ks]
    // (fragment reads one k-step ahead of the MFMAs: 2 * NV fragments live instead of NV * KT -- this kernel holds 96
    // accumulator registers per wave and two waves share a SIMD's register file)
    auto project = [&](const uint8_t* w, f32x16* out) {
        Frag<NS> cur[NV], nxt[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) cur[v] = wfrag<NS>(w, v * KT, lane);
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            if (ks + 1 < KT) {
#pragma unroll
                for (int v = 0; v < NV; ++v) nxt[v] = wfrag<NS>(w, v * KT + ks + 1, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < NV; ++v) out[v] = mfma_ns<NS>(cur[v], pn[ks], out[v]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < KT) {
#pragma unroll
                for (int v = 0; v < NV; ++v) cur[v] = nxt[v];
            }
        }
    };

#pragma unroll
    for (int k = 0; k < NBUF - 1; ++k) issue_tile(k);
    load_p(0, pn);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const float* bb = sbias + (chainA ? 0 : FE) + LW * h;
    for (int it = 0; it < n_it; ++it) {
        const int bf = it % NBUF;
        const int64_t row0 = r_begin + (int64_t)it * IR + 32 * rg;
        const bool valid = row0 + m < r_end;
        if constexpr (PPF) load_p(it + 1, pnn);                      // requested first: the end-of-iteration wait covers it
        const int n_tile = issue_tile(it + NBUF - 1);                // in flight during this (and the next NBUF - 2) iterations
        if (row0 + 32 > r_end) {                                     // ragged end of the last chunk (wave-uniform branch)
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) pn[ks] = valid ? pn[ks] : zero_frag<NS>();
        }
        int n_store = 0;
        // G2's gate values of this lane's features, phase 1 -> phase 2: IO precision (bf16 IO: 16 registers instead of 32;
        // dh is rounded to the IO dtype when it is staged anyway)
        Frag<1> gqp[NS == 1 ? LW / 8 : 1];                          // bf16 IO: packed, 8 values per 4 registers
        float gqf[NS == 1 ? 1 : LW];                                  // fp32 IO                                                // G2: the gate values of this lane's features

        // ================= phase 1
        if (role == 1 || role == 3) {
            f32x16 au[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 tb = *reinterpret_cast<const f32x4*>(bb + 16 * v + 4 * q);
                    au[v][4 * q] = tb[0]; au[v][4 * q + 1] = tb[1]; au[v][4 * q + 2] = tb[2]; au[v][4 * q + 3] = tb[3];
                }
            }
            project(Wup, au);
            if (role == 1) {                                         // A2: h = s2*x2 + sd*a_A  -> exchange (fp32)
                const uint8_t* tx = Tx2(bf);
#pragma unroll
                for (int e = 0; e < G::E4; ++e) {
                    float r8[8];
                    tile_lane_vals8<IO>(tx, m, h, e, r8);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x4 t;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int i = 8 * e + 4 * q + j;
                            t[j] = s2 * r8[4 * q + j] + sd_ * au[(i >> 4) % NV][i & 15];
                        }
                        *reinterpret_cast<f32x4*>(XHb + (size_t)(2 * e + q) * 1024 + lane16) = t;
                    }
                }
            } else {                                                 // G2: g = sigmoid(a_G)
#pragma unroll
                for (int i = 0; i < LW; ++i) {
                    const float gv = sigmoid_f(au[(i >> 4) % NV][i & 15]);
                    if constexpr (NS == 1) gqp[i >> 3].p[0][i & 7] = (__bf16)gv; else gqf[i] = gv;
                }
            }
        } else {
            // A1 / G1: the weight gradient that needs no dh / dq  (dWd = dpre_a^T x2,  dWgd = dpre_g^T x1)
            Frag<NS> xn[G::KU];
            tile_frags(tile_of(bf), valid, xn);
            accumulate(xn);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // B1: h is in the exchange buffer

        // ================= phase 2
        if (role == 3) {                                             // G2: dh, dq of the whole tile
            const uint8_t* ty = Tdy(bf);
#pragma unroll
            for (int e = 0; e < G::E4; ++e) {
                float dy8[8], dh8[8], dq8[8], hx[8];
                tile_lane_vals8<IO>(ty, m, h, e, dy8);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(XHb + (size_t)(2 * e + q) * 1024 + lane16);
                    hx[4 * q] = t[0]; hx[4 * q + 1] = t[1]; hx[4 * q + 2] = t[2]; hx[4 * q + 3] = t[3];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float gt;
                    if constexpr (NS == 1) gt = (float)gqp[e].p[0][j]; else gt = gqf[8 * e + j];
                    const float dyp = gs * dy8[j];
                    if (gate_add) {
                        dh8[j] = dyp;
                        dq8[j] = dyp * gt * (1.0f - gt);
                    } else {
                        dh8[j] = dyp * gt;
                        dq8[j] = dh8[j] * hx[j] * (1.0f - gt);
                    }
                }
                stage_lane_vals8<IO>(Tdh, m, h, e, dh8);
                stage_lane_vals8<IO>(Tdq, m, h, e, dq8);
            }
        } else if (role == 2) {                                      // G1: dx1 = Wgd^T dpre_g  (needs nothing from the others)
            f32x16 ax[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) ax[v] = zero16();
            project(Wdt, ax);
            uint8_t* tile = Tx1(bf);
#pragma unroll
            for (int e = 0; e < G::E4; ++e) {
                float o8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int i = 8 * e + j; o8[j] = ax[(i >> 4) % NV][i & 15]; }
                stage_lane_vals8<IO>(tile, m, h, e, o8);
            }
            n_store = store_tile(dxo, row0, tile);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // B2: the dh / dq tiles are complete

        // ================= phase 3
        if (role == 0) {                                             // A1: dx2 = s2*dh + Wd^T dpre_a
            f32x16 ax[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) ax[v] = zero16();
            project(Wdt, ax);
            uint8_t* tile = Tx2(bf);
#pragma unroll
            for (int e = 0; e < G::E4; ++e) {
                float o8[8], dh8[8];
                tile_lane_vals8<IO>(Tdh, m, h, e, dh8);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int i = 8 * e + j; o8[j] = ax[(i >> 4) % NV][i & 15] + s2 * dh8[j]; }
                stage_lane_vals8<IO>(tile, m, h, e, o8);
            }
            n_store = store_tile(dxo, row0, tile);
        } else if (role == 1 || role == 3) {                         // A2: dWu = z_a^T dh;  G2: dWgu = z_g^T dq
            Frag<NS> xn[G::KU];
            tile_frags(role == 1 ? Tdh : Tdq, valid, xn);
            accumulate(xn);
        }
        // Vector-memory operations retire in order.  Issued this iteration, oldest first: [P rows of it+1 (PPF)], the tile of
        // it+NBUF-1, the output stores, [P rows of it+1 (!PPF)].  The next iteration needs the tile of it+1 (this
        // iteration's when NBUF = 2, the previous one's otherwise) and the P rows; the stores may stay in flight.
        if constexpr (PPF) {
            wait_vm((NBUF > 2 ? n_tile : 0) + n_store);
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) pn[ks] = pnn[ks];
        } else {
            load_p(it + 1, pn);                                      // (consumed at the top of the next iteration: the compiler's wait)
            wait_vm(NPL + n_store + (NBUF > 2 ? n_tile : 0));
        }
        __builtin_amdgcn_s_barrier();                                // B3: next tiles landed; dh / dq / exchange free again
    }

    // ---- reduce the two row groups (fixed order) and emit this chunk's partial
    constexpr int NVAL = RT * NV * 16 + RT + NV;
    float* red = reinterpret_cast<float*>(smem) + (size_t)role * NVAL * 64;
    if (NRG > 1 && rg == 1) {
        int k = 0;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int nt = 0; nt < NV; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) red[(k++) * 64 + lane] = acc[ct][nt][i];
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) red[(k++) * 64 + lane] = csp[ct];
#pragma unroll
        for (int nt = 0; nt < NV; ++nt) red[(k++) * 64 + lane] = csx[nt];
    }
    __syncthreads();
    if (rg == 0) {
        if constexpr (NRG > 1) {
            int k = 0;
#pragma unroll
            for (int ct = 0; ct < RT; ++ct)
#pragma unroll
                for (int nt = 0; nt < NV; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[ct][nt][i] += red[(k++) * 64 + lane];
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) csp[ct] += red[(k++) * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < NV; ++nt) csx[nt] += red[(k++) * 64 + lane];
        }

        // (all four jobs have xcols = d: job j's block starts at j * RC * (PR*d + d + PR) floats -- wgrad_layout)
        const int xc = d, RC = a.wg.row_chunks, n0 = su * FE;
        float* part = a.wg.partial + (int64_t)role * RC * ((int64_t)PR * xc + xc + PR);
        float* tile = part + (int64_t)rc * PR * xc;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int nt = 0; nt < NV; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int crow = 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h;
                    tile[(int64_t)crow * xc + n0 + 32 * nt + m] = acc[ct][nt][i];
                }
        // column sums: the two half-waves hold disjoint row subsets of the same column
        float* psx = part + (int64_t)RC * PR * xc + (int64_t)rc * xc;
#pragma unroll
        for (int nt = 0; nt < NV; ++nt) {
            const float v = csx[nt] + __shfl_xor(csx[nt], 32);
            if (h == 0) psx[n0 + 32 * nt + m] = v;
        }
        if (su == 0) {
            float* psp = part + (int64_t)RC * PR * xc + (int64_t)RC * xc + (int64_t)rc * PR;
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) {
                const float v = csp[ct] + __shfl_xor(csp[ct], 32);
                if (h == 0) psp[32 * ct + m] = v;
            }
        }
    }
}

// ================================================================================================== pass 2, two waves per tile
// Same tile decomposition, but a row group is carried by TWO waves instead of four: the adapter-chain wave owns dWd and
// dWu (+ h, dx2), the gate-chain wave dWgd and dWgu (+ g, dh / dq, dx1).  With four roles each wave sat idle through two of
// the three phases of an iteration (the roles form a chain: projection -> h -> dh / dq -> dx2 / dWu / dWgu) and a
// workgroup finished one tile per ~7 k cycles; with two roles a wave has work in every phase and a workgroup of four
// waves (one per SIMD, the whole register file each: 192 accumulator registers + operands) carries two tiles at a time.
template <typename IO, int RT, int NRG, int NBUF>
__global__ __launch_bounds__(NRG * 128) void pet_gate_cols2_kernel(ColsArgs a) {
    using G = Geo4<IO>;
    using L = ColsLds<IO, NRG, NBUF>;
    constexpr int IR = 32 * NRG;
    constexpr int NS = G::NS, KT = 2 * RT, NV = G::NV, FE = G::FE, LW = G::LW;
    constexpr int SEG_B = 4 * RT * 1024;
    constexpr int PR = 32 * RT;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int b = blockIdx.x, xcd = b & 7, kq = b >> 3;
    const int sg = kq % a.GS, unit = (kq / a.GS) * 8 + xcd;
    const int rc = unit / a.NG, su = (unit % a.NG) * a.GS + sg;
    if (rc >= a.wg.row_chunks) return;
    const int64_t r_begin = (int64_t)rc * a.wg.rows_per_chunk;
    if (r_begin >= a.M) return;
    int64_t r_end = r_begin + a.wg.rows_per_chunk;
    if (r_end > a.M) r_end = a.M;
    const int n_it = (int)((r_end - r_begin + IR - 1) / IR);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave % NRG, chain = wave / NRG;                   // chain 0: adapter (A1 + A2), 1: gate (G1 + G2)
    const bool isA = chain == 0;
    const int m = lane & 31, h = lane >> 5;
    const int lane16 = lane * 16;
    const int d = a.d;
    const PackGeom pg = pack_geom(RT, d, NS);

    uint8_t* Wup = smem + (isA ? 0 : SEG_B);
    uint8_t* Wdt = smem + 2 * SEG_B + (isA ? 0 : SEG_B);
    uint8_t* rgb = smem + 4 * SEG_B + (size_t)rg * L::RG_B;
    auto Tx2 = [&](int bf) { return rgb + (size_t)(0 * L::NBUF + bf) * L::TILE_B; };
    auto Tdy = [&](int bf) { return rgb + (size_t)(1 * L::NBUF + bf) * L::TILE_B; };
    auto Tx1 = [&](int bf) { return rgb + (size_t)(2 * L::NBUF + bf) * L::TILE_B; };
    uint8_t* Tdh = rgb + (size_t)(3 * L::NBUF) * L::TILE_B;
    uint8_t* Tdq = Tdh + L::TILE_B;
    uint8_t* XHb = Tdq + L::TILE_B;
    float* sbias = reinterpret_cast<float*>(smem + 4 * SEG_B + NRG * L::RG_B);

    for (int k = wave; k < 16 * RT; k += 2 * NRG) {
        const int blk = k / (4 * RT), piece = k % (4 * RT);
        const uint8_t* pkx = (blk & 1) ? a.pk_g : a.pk_a;
        const uint8_t* src = pkx + (int64_t)(blk < 2 ? 1 : 3) * pg.pack_bytes + (int64_t)su * SEG_B + (size_t)piece * 1024;
        glds16(src + lane16, smem + (size_t)blk * SEG_B + (size_t)piece * 1024);
    }
    for (int i = tid; i < 2 * FE; i += NRG * 128) {
        const uint8_t* pkx = i < FE ? a.pk_a : a.pk_g;
        sbias[i] = reinterpret_cast<const float*>(pkx + pg.bias_off)[PR + su * FE + (i % FE)];
    }

    bf16x8 I0, I1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        I0[j] = (m == 8 * h + j) ? (__bf16)1.0f : (__bf16)0.0f;
        I1[j] = (m == 16 + 8 * h + j) ? (__bf16)1.0f : (__bf16)0.0f;
    }
    // chain A: x tile = x2, P rows = dpre_a (job 0) and z_a (job 1); chain G: x tile = x1 (+ the dy tile), dpre_g (job 2), z_g (job 3)
    const uint8_t* Xsrc = reinterpret_cast<const uint8_t*>(isA ? a.x2 : a.x1);
    const uint8_t* Ysrc = reinterpret_cast<const uint8_t*>(a.dy);
    const IO* Pdp = reinterpret_cast<const IO*>(isA ? a.dp_a : a.dp_g);
    const IO* Pz = reinterpret_cast<const IO*>(isA ? a.z_a : a.z_g);
    uint8_t* dxo = reinterpret_cast<uint8_t*>(isA ? a.dx2 : a.dx1);
    const float s2 = a.s2, sd_ = a.sd, gs = a.gs;
    const bool gate_add = (a.flags & PET_GATE_ADD) != 0;

    f32x16 accD[RT][NV], accU[RT][NV];           // down-weight job (P = dpre), up-weight job (P = z)
    float cspD[RT], cspU[RT], csxD[NV], csxU[NV];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) {
        cspD[ct] = cspU[ct] = 0.f;
#pragma unroll
        for (int nt = 0; nt < NV; ++nt) { accD[ct][nt] = zero16(); accU[ct][nt] = zero16(); }
    }
#pragma unroll
    for (int nt = 0; nt < NV; ++nt) csxD[nt] = csxU[nt] = 0.f;

    uint32_t voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tr = 8 * i + (lane >> 3);
        voff[i] = (uint32_t)tr * (uint32_t)(d * (int)sizeof(IO)) + (uint32_t)(((lane & 7) ^ swz(tr)) * 16);
    }
    auto load_tile = [&](const uint8_t* src, int64_t row0t, uint8_t* dst) {
        if (row0t + 32 <= a.M) {
            const uint8_t* base = src + row0t * d * (int64_t)sizeof(IO) + su * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16_row(base + voff[i], dst + i * 1024);
        } else {
            const RowLanes rl = row_lanes<IO>(row0t, a.M, d, 0, lane);
            glds_rows4(src, rl, su * 128, dst, 0);
        }
    };
    auto issue_tiles = [&](int it) {                                 // A: x2;  G: x1 and dy
        if (it >= n_it) return 0;
        const int64_t row0t = r_begin + (int64_t)it * IR + 32 * rg;
        if (isA) { load_tile(Xsrc, row0t, Tx2(it % NBUF)); return 4; }
        load_tile(Xsrc, row0t, Tx1(it % NBUF));
        load_tile(Ysrc, row0t, Tdy(it % NBUF));
        return 8;
    };
    auto store_tile = [&](uint8_t* base_out, int64_t row0t, const uint8_t* tile) {
        if (row0t + 32 <= a.M) {
            uint8_t* base = base_out + row0t * d * (int64_t)sizeof(IO) + su * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(tile + ((size_t)(8 * i + (lane >> 3)) * 8 + (lane & 7)) * 16);
                *reinterpret_cast<u32x4*>(base + voff[i]) = v;
            }
            return 4;
        }
        const RowLanes rl = row_lanes<IO>(row0t, a.M, d, 0, lane);
        store_rows4(base_out, rl, su * 128, tile, 0, lane);
        return rl.n_inst;
    };
    // P rows of the tile in registers, re-loaded IN PLACE for the next tile right after their last use (a second register set
    // for the next iteration does not fit beside the 192 accumulator registers)
    Frag<NS> pdp[KT], pz[KT];
    auto load_p = [&](int it, const IO* P, Frag<NS>* dst) {
        int64_t row = r_begin + (int64_t)it * IR + 32 * rg + m;
        if (row >= r_end) row = r_end - 1;
        const IO* p0 = P + row * (int64_t)PR + 8 * h;
        const bool ok = r_begin + (int64_t)it * IR + 32 * rg + m < r_end;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) { const Frag<NS> f = load_frag8(p0 + 16 * ks); dst[ks] = ok ? f : zero_frag<NS>(); }
    };
    // down job: the column sums of P (= bias gradient of the down projection) are wanted; up job: those of X (dh / dq)
    auto accumulate = [&](const uint8_t* tile, bool valid, const Frag<NS>* pp, f32x16 (&acc)[RT][NV], float* csp, float* csx, auto want_p) {
        constexpr bool WP = decltype(want_p)::value;
        Frag<NS> xt[NV][2];
#pragma unroll
        for (int nt = 0; nt < NV; ++nt) {
            Frag<NS> x0 = tile_bfrag4<IO>(tile, m, h, 2 * nt), x1f = tile_bfrag4<IO>(tile, m, h, 2 * nt + 1);
            x0 = valid ? x0 : zero_frag<NS>();
            x1f = valid ? x1f : zero_frag<NS>();
            const f32x16 t = transpose32<NS>(x0, x1f, I0, I1);
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i] = t[i]; if constexpr (!WP) csx[nt] += t[i]; }
            xt[nt][0] = frag_from_f32<NS>(v);
            xt[nt][1] = frag_from_f32<NS>(v + 8);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) {
            const f32x16 t = transpose32<NS>(pp[2 * ct], pp[2 * ct + 1], I0, I1);
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i] = t[i]; if constexpr (WP) csp[ct] += t[i]; }
            const Frag<NS> pt0 = frag_from_f32<NS>(v), pt1 = frag_from_f32<NS>(v + 8);
#pragma unroll
            for (int nt = 0; nt < NV; ++nt) acc[ct][nt] = mfma_ns<NS>(pt0, xt[nt][0], acc[ct][nt]);
#pragma unroll
            for (int nt = 0; nt < NV; ++nt) acc[ct][nt] = mfma_ns<NS>(pt1, xt[nt][1], acc[ct][nt]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto project = [&](const uint8_t* w, const Frag<NS>* pp, f32x16* out) {
        Frag<NS> cur[NV], nxt[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) cur[v] = wfrag<NS>(w, v * KT, lane);
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            if (ks + 1 < KT) {
#pragma unroll
                for (int v = 0; v < NV; ++v) nxt[v] = wfrag<NS>(w, v * KT + ks + 1, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < NV; ++v) out[v] = mfma_ns<NS>(cur[v], pp[ks], out[v]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < KT) {
#pragma unroll
                for (int v = 0; v < NV; ++v) cur[v] = nxt[v];
            }
        }
    };

#pragma unroll
    for (int k = 0; k < NBUF - 1; ++k) issue_tiles(k);
    load_p(0, Pdp, pdp);
    load_p(0, Pz, pz);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const float* bb = sbias + (isA ? 0 : FE) + LW * h;
    // one loop per chain (the two chains share nothing but the barriers): with both in one loop body every accumulator is a
    // phi of the two branches at each phase and the register allocator copies / spills them
    auto run = [&](auto chain_a) {
    constexpr bool CA = decltype(chain_a)::value;
    for (int it = 0; it < n_it; ++it) {
        const int bf = it % NBUF;
        const int64_t row0 = r_begin + (int64_t)it * IR + 32 * rg;
        const bool valid = row0 + m < r_end;
        const int n_tile = issue_tiles(it + NBUF - 1);
        int n_store = 0;
        Frag<1> gqp[NS == 1 ? LW / 8 : 1];
        float gqf[NS == 1 ? 1 : LW];

        // ================= phase 1: both up projections; A hands h over; the weight gradients that need no dh / dq; G's dx1
        {
            f32x16 au[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 tb = *reinterpret_cast<const f32x4*>(bb + 16 * v + 4 * q);
                    au[v][4 * q] = tb[0]; au[v][4 * q + 1] = tb[1]; au[v][4 * q + 2] = tb[2]; au[v][4 * q + 3] = tb[3];
                }
            }
            project(Wup, pz, au);
            if constexpr (CA) {
                const uint8_t* tx = Tx2(bf);
#pragma unroll
                for (int e = 0; e < G::E4; ++e) {
                    float r8[8];
                    tile_lane_vals8<IO>(tx, m, h, e, r8);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x4 t;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int i = 8 * e + 4 * q + j;
                            t[j] = s2 * r8[4 * q + j] + sd_ * au[(i >> 4) % NV][i & 15];
                        }
                        *reinterpret_cast<f32x4*>(XHb + (size_t)(2 * e + q) * 1024 + lane16) = t;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < LW; ++i) {
                    const float gv = sigmoid_f(au[(i >> 4) % NV][i & 15]);
                    if constexpr (NS == 1) gqp[i >> 3].p[0][i & 7] = (__bf16)gv; else gqf[i] = gv;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // B1: h is in the exchange buffer

        // ================= phase 2: G: dh, dq of the tile;  A: dWd = dpre_a^T x2 (independent of dh)
        if constexpr (CA) accumulate(Tx2(bf), valid, pdp, accD, cspD, csxD, std::true_type{}); else {
            const uint8_t* ty = Tdy(bf);
#pragma unroll
            for (int e = 0; e < G::E4; ++e) {
                float dy8[8], dh8[8], dq8[8], hx[8];
                tile_lane_vals8<IO>(ty, m, h, e, dy8);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(XHb + (size_t)(2 * e + q) * 1024 + lane16);
                    hx[4 * q] = t[0]; hx[4 * q + 1] = t[1]; hx[4 * q + 2] = t[2]; hx[4 * q + 3] = t[3];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float gt;
                    if constexpr (NS == 1) gt = (float)gqp[e].p[0][j]; else gt = gqf[8 * e + j];
                    const float dyp = gs * dy8[j];
                    if (gate_add) {
                        dh8[j] = dyp;
                        dq8[j] = dyp * gt * (1.0f - gt);
                    } else {
                        dh8[j] = dyp * gt;
                        dq8[j] = dh8[j] * hx[j] * (1.0f - gt);
                    }
                }
                stage_lane_vals8<IO>(Tdh, m, h, e, dh8);
                stage_lane_vals8<IO>(Tdq, m, h, e, dq8);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // B2: the dh / dq tiles are complete

        // ================= phase 3: A: dx2, dWu;  G: dWgd, dx1, dWgu
        if constexpr (CA) {
            accumulate(Tdh, valid, pz, accU, cspU, csxU, std::false_type{});
            load_p(it + 1, Pz, pz);
            f32x16 ax[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) ax[v] = zero16();
            project(Wdt, pdp, ax);
            load_p(it + 1, Pdp, pdp);
            uint8_t* tile = Tx2(bf);
#pragma unroll
            for (int e = 0; e < G::E4; ++e) {
                float o8[8], dh8[8];
                tile_lane_vals8<IO>(Tdh, m, h, e, dh8);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int i = 8 * e + j; o8[j] = ax[(i >> 4) % NV][i & 15] + s2 * dh8[j]; }
                stage_lane_vals8<IO>(tile, m, h, e, o8);
            }
            n_store = store_tile(dxo, row0, tile);
        } else {
            accumulate(Tdq, valid, pz, accU, cspU, csxU, std::false_type{});
            load_p(it + 1, Pz, pz);
            accumulate(Tx1(bf), valid, pdp, accD, cspD, csxD, std::true_type{});
            f32x16 ax[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) ax[v] = zero16();
            project(Wdt, pdp, ax);
            uint8_t* tile = Tx1(bf);
#pragma unroll
            for (int e = 0; e < G::E4; ++e) {
                float o8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int i = 8 * e + j; o8[j] = ax[(i >> 4) % NV][i & 15]; }
                stage_lane_vals8<IO>(tile, m, h, e, o8);
            }
            load_p(it + 1, Pdp, pdp);
            n_store = store_tile(dxo, row0, tile);
        }
        // in order on vmcnt, oldest first: tiles of it+NBUF-1, z rows of it+1, dpre rows of it+1, output stores.  The next
        // iteration needs the tiles of it+1 (this iteration's when NBUF = 2, the previous one's otherwise); the register loads
        // are waited for by the compiler's own counters at their first use.
        if constexpr (NBUF > 2) wait_vm(n_tile + 2 * KT * NS + n_store);
        else wait_vm(2 * KT * NS + n_store);
        __builtin_amdgcn_s_barrier();                                // B3
    }

    };
    if (isA) run(std::true_type{}); else run(std::false_type{});

    // ---- reduce the row groups (fixed order) and emit this chunk's partials: jobs 2*chain (down) and 2*chain + 1 (up)
    constexpr int NVAL = RT * NV * 16 + RT + NV;
    __syncthreads();
    float* redD = reinterpret_cast<float*>(smem) + (size_t)(2 * chain) * NVAL * 64;
    float* redU = redD + (size_t)NVAL * 64;
    auto spill_out = [&](float* red, f32x16 (&acc)[RT][NV], float* csp, float* csx) {
        int k = 0;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int nt = 0; nt < NV; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) red[(k++) * 64 + lane] = acc[ct][nt][i];
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) red[(k++) * 64 + lane] = csp[ct];
#pragma unroll
        for (int nt = 0; nt < NV; ++nt) red[(k++) * 64 + lane] = csx[nt];
    };
    auto add_in = [&](const float* red, f32x16 (&acc)[RT][NV], float* csp, float* csx) {
        int k = 0;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int nt = 0; nt < NV; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ct][nt][i] += red[(k++) * 64 + lane];
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) csp[ct] += red[(k++) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < NV; ++nt) csx[nt] += red[(k++) * 64 + lane];
    };
    auto emit = [&](int job, f32x16 (&acc)[RT][NV], float* csp, float* csx) {
        const int xc = d, RC = a.wg.row_chunks, n0 = su * FE;
        float* part = a.wg.partial + (int64_t)job * RC * ((int64_t)PR * xc + xc + PR);
        float* tile = part + (int64_t)rc * PR * xc;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int nt = 0; nt < NV; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int crow = 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h;
                    tile[(int64_t)crow * xc + n0 + 32 * nt + m] = acc[ct][nt][i];
                }
        float* psx = part + (int64_t)RC * PR * xc + (int64_t)rc * xc;
#pragma unroll
        for (int nt = 0; nt < NV; ++nt) {
            const float v = csx[nt] + __shfl_xor(csx[nt], 32);
            if (h == 0) psx[n0 + 32 * nt + m] = v;
        }
        if (su == 0) {
            float* psp = part + (int64_t)RC * PR * xc + (int64_t)RC * xc + (int64_t)rc * PR;
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) {
                const float v = csp[ct] + __shfl_xor(csp[ct], 32);
                if (h == 0) psp[32 * ct + m] = v;
            }
        }
    };
    if constexpr (NRG > 1) {
        static_assert(NRG == 2, "two row groups per workgroup");
        if (rg == 1) { spill_out(redD, accD, cspD, csxD); spill_out(redU, accU, cspU, csxU); }
        __syncthreads();
        if (rg == 0) { add_in(redD, accD, cspD, csxD); add_in(redU, accU, cspU, csxU); }
    }
    if (rg == 0) {
        emit(2 * chain, accD, cspD, csxD);
        emit(2 * chain + 1, accU, cspU, csxU);
    }
}

// ================================================================================================== host side
void gate_bwd3_plan(int64_t M, int d, int io_fp32, int* row_chunks, int64_t* rows_per_chunk, int* GS, int* NG) {
    const int S = d / (io_fp32 ? 32 : 64);
    int gs = 1;
    for (int g = 1; g <= 8 && g <= S; ++g) if (S % g == 0) gs = g;   // feature blocks that share an XCD's L2
    const int ng = S / gs;
    // one workgroup (512 threads, ~130 KiB of LDS) per CU and 32 CUs per XCD: floor(32 / gs) groups per XCD
    const int target_units = vlpet_tuning().bwd3_units;
    const int units = target_units > 0 ? target_units : 8 * (32 / gs);
    int64_t rc = units / ng;
    if (rc < 1) rc = 1;
    const int64_t blocks64 = (M + 63) / 64;
    if (rc > blocks64) rc = blocks64;
    const int64_t per = (blocks64 + rc - 1) / rc;                   // 64-row blocks per chunk
    rc = (blocks64 + per - 1) / per;
    *row_chunks = (int)rc; *rows_per_chunk = per * 64; *GS = gs; *NG = ng;
}
static inline bool bwd3_rt_ok(int RT) { return RT == 1 || RT == 3 || RT == 6; }

// Where the two-pass form is the default: r = 192, whose one-kernel row pass (pet_bwd_kernel<.., 6, gate>) spills 350
// registers (two-pass 358 us vs 451 us at M = 16,800).  For r <= 96 the previous split (pet_gate_bwd2 + wgrad) is faster on
// MI355X despite its 11 units of traffic (164 us vs 234 us at M = 28,000): pass 2 is latency-chain bound (see launch_cols_one).
// VLPET_BWD3 = 1 / 0 forces it on / off for every supported rank.
bool pet_gate_bwd3_applies(const PetBwdArgs& a) {
    const int force = vlpet_tuning().bwd3;
    if (force == 0) return false;
    if (!((a.flags & PET_GATE) && a.saved != nullptr && !drop_active(a.drop) && bwd3_rt_ok(a.RT))) return false;
    // small M (a strong-scaled rank: a few thousand rows): both forms are latency chains of a handful of workgroups, and the
    // two-pass one is the shorter (M = 3,500: 79 vs 88 us; M = 512: 74 vs 79 us; profiles/r02_kbench_two_pass_ab.txt)
    return force == 1 || a.RT == 6 || a.M <= 4096;
}

template <typename IO, int RT, int RG>
static hipError_t launch_dz_one(const PetBwdArgs& a, hipStream_t stream) {
    using L = DzLds<IO, RT, RG>;
    const size_t lds = L::bytes(a.d);
    auto kern = pet_gate_dz_kernel<IO, RT, RG>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int rows = RG * 32;
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.M + rows - 1) / rows)), dim3(RG * 128), lds, stream, a);
    return hipGetLastError();
}
template <typename IO, int RT>
static hipError_t launch_dz_rt(const PetBwdArgs& a, hipStream_t stream) {
    if constexpr (RT == 6) {
        return launch_dz_one<IO, RT, 2>(a, stream);      // r = 192: the 2 x 48 KiB weight ring leaves room for 64 rows
    } else {
        switch (pick_row_groups(a.M, 4, 2)) {
            case 4: return launch_dz_one<IO, RT, 4>(a, stream);
            case 3: if constexpr ((4 * RT) % 3 == 0) return launch_dz_one<IO, RT, 3>(a, stream);   // (else: falls through)
            default: return launch_dz_one<IO, RT, 2>(a, stream);
        }
    }
}
hipError_t launch_pet_gate_dz(const PetBwdArgs& a, int io_fp32, hipStream_t stream) {
    if (a.RT == 1) return io_fp32 ? launch_dz_rt<float, 1>(a, stream) : launch_dz_rt<__bf16, 1>(a, stream);
    if (a.RT == 3) return io_fp32 ? launch_dz_rt<float, 3>(a, stream) : launch_dz_rt<__bf16, 3>(a, stream);
    if (a.RT == 6) return io_fp32 ? launch_dz_rt<float, 6>(a, stream) : launch_dz_rt<__bf16, 6>(a, stream);
    return hipErrorInvalidValue;
}

template <typename IO, int RT, int NRG, int NBUF, bool PPF>
static hipError_t launch_cols_cfg(const ColsArgs& c, hipStream_t stream) {
    const size_t lds = ColsLds<IO, NRG, NBUF>::bytes(RT);
    auto kern = pet_gate_cols_kernel<IO, RT, NRG, NBUF, PPF>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int units = c.NG * c.wg.row_chunks;
    const unsigned grid = 8u * (unsigned)c.GS * (unsigned)((units + 7) / 8);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NRG * 256), lds, stream, c);
    return hipGetLastError();
}

// workgroup shape of pass 2.  Measured at M = 28,000, bf16, r = 96 (profiles/r02_kbench_two_pass_ab.txt): 8 waves (2 row
// groups x 4 roles, 256 registers each) spill 70-140 registers and take 290-410 us -- every scratch reload waits, in order,
// behind the tile prefetches; 4 waves (one per SIMD, no spill, P rows prefetched) take 189 us whether the tiles run one or
// two iterations ahead, i.e. the pass is bound by the three-phase dependency chain of an iteration (~7k cycles per 32 rows),
// not by memory.  Kept: the 4-wave shape (r > 32) and the 8-wave shape where it does not spill (r <= 32).
template <typename IO, int RT, int NRG, int NBUF>
static hipError_t launch_cols2_cfg(const ColsArgs& c, hipStream_t stream) {
    const size_t lds = ColsLds<IO, NRG, NBUF>::bytes(RT);
    auto kern = pet_gate_cols2_kernel<IO, RT, NRG, NBUF>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int units = c.NG * c.wg.row_chunks;
    const unsigned grid = 8u * (unsigned)c.GS * (unsigned)((units + 7) / 8);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NRG * 128), lds, stream, c);
    return hipGetLastError();
}

template <typename IO, int RT>
static hipError_t launch_cols_one(const ColsArgs& c, hipStream_t stream) {
    // VLPET_BWD3_FORM: 0 = four roles per row group (4 waves, one tile at a time); 1 (default for r <= 96) = two waves per row
    // group, two row groups per workgroup (198.8 -> 137.2 us at M = 28 k, 58.9 -> 55.0 at 3.5 k; profiles/r02_kbench_pass2_forms.txt)
    const int form = vlpet_tuning().bwd3_form;
    if constexpr (RT == 3) {
        if (form == 1) return launch_cols2_cfg<IO, RT, 2, 2>(c, stream);
        if (form == 2) return launch_cols2_cfg<IO, RT, 2, 3>(c, stream);
    }
    if constexpr (RT == 6) {
        return launch_cols_cfg<IO, RT, 1, 2, false>(c, stream);     // 96 KiB of weight fragments: room for two tile buffers
    } else if constexpr (RT == 1) {
        return launch_cols_cfg<IO, RT, 2, 2, false>(c, stream);
    } else {
        return launch_cols_cfg<IO, RT, 1, 3, true>(c, stream);
    }
}

// pass 2 + the partial reduction.  `g` = the four weight-gradient jobs as run_bwd builds them for launch_wgrad (job order
// dWd, dWu, dWgd, dWgu; partial = the workspace block), with row_chunks / rows_per_chunk from gate_bwd3_plan.
hipError_t launch_pet_gate_cols(const PetBwdArgs& a, const WgradArgs& g, int GS, int NG, int io_fp32, hipStream_t stream) {
    ColsArgs c;
    c.dy = a.dy; c.x1 = a.xg; c.x2 = a.res;
    c.z_a = a.z_a; c.z_g = a.z_g; c.dp_a = a.dp_a; c.dp_g = a.dp_g;
    c.dx1 = a.dxg; c.dx2 = a.dxa;
    c.pk_a = a.pk_a; c.pk_g = a.pk_g;
    c.M = a.M; c.d = a.d; c.RT = a.RT; c.s2 = a.s2; c.sd = a.sd; c.gs = a.gs; c.flags = a.flags;
    c.GS = GS; c.NG = NG; c.wg = g;
    hipError_t e;
    if (a.RT == 1) e = io_fp32 ? launch_cols_one<float, 1>(c, stream) : launch_cols_one<__bf16, 1>(c, stream);
    else if (a.RT == 3) e = io_fp32 ? launch_cols_one<float, 3>(c, stream) : launch_cols_one<__bf16, 3>(c, stream);
    else if (a.RT == 6) e = io_fp32 ? launch_cols_one<float, 6>(c, stream) : launch_cols_one<__bf16, 6>(c, stream);
    else return hipErrorInvalidValue;
    if (e != hipSuccess) return e;
    return launch_wgrad_finalize(g, stream);
}
