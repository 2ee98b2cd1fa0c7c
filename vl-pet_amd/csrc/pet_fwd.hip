// Fused PET forward (K1 encoder adapter+gate, K2 decoder value-parallel adapter, K3 LoRA delta):
//
//   out = ( s2*res + sd*( up_A( act( down_A(xa) ) ) ) )  [ (*|+) sigmoid( up_G( gelu_new( down_G(xg) ) ) ) ] * gs
//
// Reference op chains replaced: my_transformers/modeling_bart.py:1147-1155,1195-1209,1256-1257
// (K1), adapters/adapter_modeling.py:55-61 + adapter_controller.py:149-162 (K2),
// lora/controller.py:56-70 (K3, on top of the PyTorch base GEMM).
//
// Structure (pet32.h): workgroup = WAVES x 32 rows; each wave carries its 32 rows through the whole
// chain in registers on v_mfma_f32_32x32x16_bf16 (down-projection accumulators -> bias/gelu -> bf16 B
// fragments -> up-projection -> residual/gate epilogue).  Everything that comes from memory arrives by
// global_load_lds (no staging registers, whole 128-byte lines):
//   * weight fragments (shared by the waves, L2-resident): 2-slot ring, one stage ahead;
//   * each wave's own 32 x 128-byte row pieces (HBM): 3-slot ring, two stages ahead, source-swizzled so
//     the fragment reads are bank-conflict free.
// A stage ends with a counted s_waitcnt vmcnt(N) (only what the NEXT stage needs must have landed; the
// rows of the stage after next and this stage's output stores stay in flight) and a raw s_barrier.
// Outputs are staged in place in the residual tile and stored as whole lines.
// HBM traffic = read xa, read xg, re-read res (= xa for K1), write out.
#include "common.h"
#include "kernels.h"
#include "pet32.h"

template <typename IO, int RT, bool GATE, int WAVES>
struct FwdLds {
    static constexpr int NS = Geo4<IO>::NS;
    static constexpr int SEG_KB = 4 * RT;                         // one chain's weights per stage (KiB)
    static constexpr int SEG_FR = SEG_KB / NS;                    // ... in fragments
    static constexpr int W_B = SEG_KB * 1024 * (GATE ? 2 : 1);
    static constexpr int TILE_B = WAVES * 32 * 128;
    static constexpr int ROW_B = TILE_B * (GATE ? 2 : 1);
    static constexpr int NR = 3;                                  // row-ring slots
    static constexpr int ROW_OFF = 2 * W_B;
    static constexpr int BIAS_OFF = ROW_OFF + NR * ROW_B;
    static size_t bytes(int d) { return (size_t)BIAS_OFF + (size_t)2 * (32 * RT + d) * 4; }
};

template <typename IO, int RT, bool GATE, bool GATE_ADD, bool ACT_ID, bool DROP, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void pet_fwd_kernel(PetFwdArgs a) {
    using G = Geo4<IO>;
    using L = FwdLds<IO, RT, GATE, WAVES>;
    constexpr int NS = G::NS;
    constexpr int KT = 2 * RT;                   // k-steps (16) of the up projection
    constexpr int NTEN = GATE ? 2 : 1;           // row tensors in the down phase
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int trow = 32 * wave + m;
    const int d = a.d;
    const int64_t row0_wave = (int64_t)blockIdx.x * (WAVES * 32) + wave * 32;
    const int S = d / G::FE;                     // stages per phase
    const PackGeom pg = pack_geom(RT, d, NS);
    const uint8_t* pkA = a.pk_a;
    const uint8_t* pkG = GATE ? a.pk_g : a.pk_a;
    const uint8_t* xa = reinterpret_cast<const uint8_t*>(a.xa);
    const uint8_t* xg = reinterpret_cast<const uint8_t*>(a.xg);
    const uint8_t* res = reinterpret_cast<const uint8_t*>(a.res);
    uint8_t* out = reinterpret_cast<uint8_t*>(a.out);

    auto slot_w = [&](int j) { return smem + (size_t)j * L::W_B; };
    auto slot_ta = [&](int j) { return smem + L::ROW_OFF + (size_t)j * L::ROW_B; };
    auto slot_tg = [&](int j) { return smem + L::ROW_OFF + (size_t)j * L::ROW_B + L::TILE_B; };
    float* sb = reinterpret_cast<float*>(smem + L::BIAS_OFF);
    const int nb = 32 * RT + d;

    const RowLanes rl = row_lanes<IO>(row0_wave, a.M, d, wave, lane);
    const int lane16 = lane * 16;

    auto rows_count = [&](int s2) { return s2 < S ? 4 * NTEN : (s2 < 2 * S ? 4 : 0); };
    auto issue_rows = [&](int s2) {
        if (s2 >= 2 * S) return;
        const int j = s2 % L::NR;
        const bool up = s2 >= S;
        const int so = (up ? s2 - S : s2) * 128;          // 128 bytes of every row per stage, both dtypes
        glds_rows4(up ? res : xa, rl, so, slot_ta(j), wave);
        if constexpr (GATE) {
            if (!up) glds_rows4(xg, rl, so, slot_tg(j), wave);
        }
    };
    auto issue_w = [&](int s1) {
        if (s1 >= 2 * S) return;
        const bool up = s1 >= S;
        const int64_t woff = (up ? pg.pack_bytes : 0) + (int64_t)(up ? s1 - S : s1) * L::SEG_KB * 1024;
        uint8_t* dst = slot_w(s1 & 1);
        constexpr int KB = L::SEG_KB * NTEN;
        for (int k = wave; k < KB; k += WAVES) {
            const uint8_t* src = (k < L::SEG_KB ? pkA + woff + (size_t)k * 1024
                                                : pkG + woff + (size_t)(k - L::SEG_KB) * 1024) + lane16;
            glds16(src, dst + (size_t)k * 1024);
        }
    };

    issue_w(0);
    issue_rows(0);
    issue_rows(1);
    {   // biases -> LDS: [bdA(32RT) | buA(d) | bdG(32RT) | buG(d)]
        const float* ba = reinterpret_cast<const float*>(a.pk_a + pg.bias_off);
        for (int i = tid; i < nb; i += WAVES * 64) sb[i] = ba[i];
        if constexpr (GATE) {
            const float* bg = reinterpret_cast<const float*>(a.pk_g + pg.bias_off);
            for (int i = tid; i < nb; i += WAVES * 64) sb[nb + i] = bg[i];
        }
    }
    __syncthreads();            // drains everything issued so far (stage 0 weights, rows of stages 0 and 1)

    // ---- down projections: pre[c], register 8*sh + j of c-tile ct <-> c = 32ct + 16sh + 8h + j
    f32x16 accA[RT];
    f32x16 accG[GATE ? RT : 1];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) accA[ct] = zero16();
    if constexpr (GATE) {
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) accG[ct] = zero16();
    }
    int s = 0;
    for (; s < S; ++s) {
        issue_w(s + 1);
        issue_rows(s + 2);
        const uint8_t* w = slot_w(s & 1);
        const uint8_t* ta = slot_ta(s % L::NR);
        const uint8_t* tg = slot_tg(s % L::NR);
        // all fragment reads of a chain first (one LDS burst, counted lgkmcnt waits), then its MFMAs
        Frag<NS> bA[G::KU], wa[G::KU * RT];
#pragma unroll
        for (int u = 0; u < G::KU; ++u) {
            bA[u] = tile_bfrag4<IO>(ta, trow, h, u);
            if constexpr (DROP) {
                const int64_t grow = (row0_wave + m < a.M) ? row0_wave + m : a.M - 1;
                const uint64_t kp = *reinterpret_cast<const uint64_t*>(a.keep + grow * d + s * G::FE + 16 * u + 8 * h);
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = (float)bA[u].p[0][j];
                    if constexpr (NS == 2) v[j] += (float)bA[u].p[1][j];
                    v[j] = ((kp >> (8 * j)) & 0xff) ? v[j] * a.keep_scale : 0.f;
                }
                bA[u] = frag_from_f32<NS>(v);
            }
        }
#pragma unroll
        for (int i = 0; i < G::KU * RT; ++i) wa[i] = wfrag<NS>(w, i, lane);
        if constexpr (GATE) {
            Frag<NS> bG[G::KU], wg[G::KU * RT];
#pragma unroll
            for (int u = 0; u < G::KU; ++u) bG[u] = tile_bfrag4<IO>(tg, trow, h, u);
#pragma unroll
            for (int i = 0; i < G::KU * RT; ++i) wg[i] = wfrag<NS>(w, L::SEG_FR + i, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < G::KU; ++u) {
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) accA[ct] = mfma_ns<NS>(wa[u * RT + ct], bA[u], accA[ct]);
            }
#pragma unroll
            for (int u = 0; u < G::KU; ++u) {
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) accG[ct] = mfma_ns<NS>(wg[u * RT + ct], bG[u], accG[ct]);
            }
        } else {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < G::KU; ++u) {
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) accA[ct] = mfma_ns<NS>(wa[u * RT + ct], bA[u], accA[ct]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // next stage needs: its weights (issued first in this stage) and its rows (issued a stage earlier)
        wait_vm(rows_count(s + 2));
        __builtin_amdgcn_s_barrier();
    }

    // ---- bias + activation -> B fragments of the up projection (k-step 2ct+sh holds c = 32ct+16sh+8h+j)
    Frag<NS> zA[KT];
    Frag<NS> zG[GATE ? KT : 1];
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
                }
                zA[2 * ct + sh] = frag_from_f32<NS>(v);
                if constexpr (GATE) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = gelu_new_f(accG[ct][8 * sh + j] + bdG[32 * ct + 16 * sh + j]);
                    zG[2 * ct + sh] = frag_from_f32<NS>(v);
                }
            }
        }
    }

    // ---- up projections + epilogue: FE features per stage, LW contiguous per lane
    const float* buA = sb + 32 * RT + G::LW * h;
    const float* buG = sb + nb + 32 * RT + G::LW * h;
    const float gs = GATE ? a.gs : 1.0f;
    const float s2g = a.s2 * gs, sdg = a.sd * gs;      // gate scale folded into the linear part
    for (; s < 2 * S; ++s) {
        issue_w(s + 1);
        issue_rows(s + 2);
        const int su = s - S;
        const uint8_t* w = slot_w(s & 1);
        uint8_t* tr = slot_ta(s % L::NR);
        float o[G::LW];
        // accumulators start at the up-projection bias (the MFMA C operand), so the epilogue has no bias add
        f32x16 aA[G::NV], aG[G::NV];
#pragma unroll
        for (int v = 0; v < G::NV; ++v) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(buA + su * G::FE + 16 * v + 4 * q);
                aA[v][4 * q] = t[0]; aA[v][4 * q + 1] = t[1]; aA[v][4 * q + 2] = t[2]; aA[v][4 * q + 3] = t[3];
                if constexpr (GATE) {
                    const f32x4 tg = *reinterpret_cast<const f32x4*>(buG + su * G::FE + 16 * v + 4 * q);
                    aG[v][4 * q] = tg[0]; aG[v][4 * q + 1] = tg[1]; aG[v][4 * q + 2] = tg[2]; aG[v][4 * q + 3] = tg[3];
                }
            }
        }
        {
            Frag<NS> wa[G::NV * KT];
#pragma unroll
            for (int i = 0; i < G::NV * KT; ++i) wa[i] = wfrag<NS>(w, i, lane);
            if constexpr (GATE) {
                Frag<NS> wg[G::NV * KT];
#pragma unroll
                for (int i = 0; i < G::NV * KT; ++i) wg[i] = wfrag<NS>(w, L::SEG_FR + i, lane);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KT; ++ks) {
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) aG[v] = mfma_ns<NS>(wg[v * KT + ks], zG[ks], aG[v]);
                }
#pragma unroll
                for (int ks = 0; ks < KT; ++ks) {
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) aA[v] = mfma_ns<NS>(wa[v * KT + ks], zA[ks], aA[v]);
                }
            } else {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KT; ++ks) {
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) aA[v] = mfma_ns<NS>(wa[v * KT + ks], zA[ks], aA[v]);
                }
            }
        }
        float r[G::LW];
        tile_lane_vals4<IO>(tr, trow, h, r);
#pragma unroll
        for (int i = 0; i < G::LW; ++i) {
            float hv = s2g * r[i] + sdg * aA[i >> 4][i & 15];
            if constexpr (GATE) {
                const float gt = sigmoid_f(aG[i >> 4][i & 15]);
                hv = GATE_ADD ? hv + gs * gt : hv * gt;
            }
            o[i] = hv;
        }
        // stage the outputs in place (the wave's own rows of the residual tile), then whole-line stores
        stage_lane_vals4<IO>(tr, trow, h, o);
        store_rows4(out, rl, su * 128, tr, wave, lane);
        wait_vm(rows_count(s + 2) + rl.n_inst);
        __builtin_amdgcn_s_barrier();
    }
}

template <typename IO, int RT, bool GATE, bool GATE_ADD, bool ACT_ID, bool DROP, int WAVES>
static hipError_t launch_one(const PetFwdArgs& a, hipStream_t stream) {
    using L = FwdLds<IO, RT, GATE, WAVES>;
    const size_t lds = L::bytes(a.d);
    auto kern = pet_fwd_kernel<IO, RT, GATE, GATE_ADD, ACT_ID, DROP, WAVES>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int rows = WAVES * 32;
    const int blocks = (int)((a.M + rows - 1) / rows);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

template <typename IO, int RT, bool GATE, bool GATE_ADD, bool ACT_ID, bool DROP>
static hipError_t launch_waves(const PetFwdArgs& a, hipStream_t stream) {
    // 4 waves (128 rows) per workgroup unless the rings would not fit the 160 KiB LDS
    if constexpr (FwdLds<IO, RT, GATE, 4>::BIAS_OFF + 8 * 1024 <= 160 * 1024)
        return launch_one<IO, RT, GATE, GATE_ADD, ACT_ID, DROP, 4>(a, stream);
    else
        return launch_one<IO, RT, GATE, GATE_ADD, ACT_ID, DROP, 2>(a, stream);
}

template <typename IO, int RT>
static hipError_t launch_rt(const PetFwdArgs& a, hipStream_t stream) {
    const bool gate = a.flags & PET_GATE, add = a.flags & PET_GATE_ADD, act_id = a.flags & PET_ACT_IDENTITY;
    const bool drop = a.keep != nullptr;
    if (gate) return add ? launch_waves<IO, RT, true, true, false, false>(a, stream)
                         : launch_waves<IO, RT, true, false, false, false>(a, stream);
    if (act_id) return drop ? launch_waves<IO, RT, false, false, true, true>(a, stream)
                            : launch_waves<IO, RT, false, false, true, false>(a, stream);
    return launch_waves<IO, RT, false, false, false, false>(a, stream);
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

hipError_t launch_pet_fwd(const PetFwdArgs& a, int io_fp32, hipStream_t stream) {
    return io_fp32 ? launch_io<float>(a, stream) : launch_io<__bf16>(a, stream);
}
