// Fused PET forward (K1 encoder adapter+gate, K2 decoder value-parallel adapter, K3 LoRA delta):
//
//   out = ( s2*res + sd*( up_A( act( down_A(xa) ) ) ) )  [ (*|+) sigmoid( up_G( gelu_new( down_G(xg) ) ) ) ] * gs
//
// Reference op chains replaced: my_transformers/modeling_bart.py:1147-1155,1195-1209,1256-1257
// (K1), adapters/adapter_modeling.py:55-61 + adapter_controller.py:149-162 (K2),
// lora/controller.py:56-70 (K3, on top of the PyTorch base GEMM).
//
// v3 structure (see pet16.h): workgroup = WAVES x 16 rows; each wave carries its 16 rows through
// the whole chain in registers (down-projection accumulators -> bias/gelu -> bf16 B fragments ->
// up-projection -> residual/gate epilogue).  8 waves per CU = 2 per SIMD, so one wave's MFMAs overlap
// the other's VALU epilogue and LDS traffic.  Everything that comes from memory arrives through a
// two-slot LDS ring filled by global_load_lds one stage ahead: the pre-packed weight fragments
// (shared by all waves) and each wave's own 16 x 128-byte row pieces (coalesced full lines,
// source-swizzled so the fragment reads are bank-conflict free).  Outputs are staged through LDS and
// stored as whole 128-byte lines.  HBM traffic = read xa (=res), read xg, write out.
#include "common.h"
#include "kernels.h"
#include "pet16.h"

template <typename IO, int RT, bool GATE, int WAVES>
struct FwdLds {
    static constexpr int NS = Geo<IO>::NS;
    static constexpr int SEG_KB = 4 * RT;                         // one chain's weights per stage (KiB)
    static constexpr int SEG_FR = SEG_KB / NS;                    // ... in fragments
    static constexpr int W_B = SEG_KB * 1024 * (GATE ? 2 : 1);
    static constexpr int TILE_B = WAVES * 16 * 128;
    static constexpr int SLOT_B = W_B + TILE_B * (GATE ? 2 : 1);
    static constexpr int STAGING_OFF = 2 * SLOT_B;
    static constexpr int BIAS_OFF = STAGING_OFF + TILE_B;
    static size_t bytes(int d) { return (size_t)BIAS_OFF + (size_t)2 * (32 * RT + d) * 4; }
};

template <typename IO, int RT, bool GATE, bool ACT_ID, bool DROP, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void pet_fwd_kernel(PetFwdArgs a) {
    using G = Geo<IO>;
    using L = FwdLds<IO, RT, GATE, WAVES>;
    constexpr int NS = G::NS;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int trow = 16 * wave + m;
    const int d = a.d;
    const int64_t row0_wave = (int64_t)blockIdx.x * (WAVES * 16) + wave * 16;
    const int S = d / G::FE;                     // stages per phase
    const PackGeom pg = pack_geom(RT, d, NS);
    const uint8_t* pkA = a.pk_a;
    const uint8_t* pkG = GATE ? a.pk_g : a.pk_a;
    const IO* xa = reinterpret_cast<const IO*>(a.xa);
    const IO* xg = reinterpret_cast<const IO*>(a.xg);
    const IO* res = reinterpret_cast<const IO*>(a.res);
    IO* out = reinterpret_cast<IO*>(a.out);

    auto slot_w = [&](int j) { return smem + (size_t)j * L::SLOT_B; };
    auto slot_ta = [&](int j) { return smem + (size_t)j * L::SLOT_B + L::W_B; };
    auto slot_tg = [&](int j) { return smem + (size_t)j * L::SLOT_B + L::W_B + L::TILE_B; };
    uint8_t* staging = smem + L::STAGING_OFF;
    float* sb = reinterpret_cast<float*>(smem + L::BIAS_OFF);
    const int nb = 32 * RT + d;

    // issue everything stage s needs into ring slot s & 1 (s in [0, 2S): down phase then up phase)
    auto issue = [&](int s) {
        if (s >= 2 * S) return;
        const int j = s & 1;
        const bool up = s >= S;
        const int ss = up ? s - S : s;
        const int64_t woff = (up ? pg.pack_bytes : 0) + (int64_t)ss * L::SEG_KB * 1024;
        glds_weights<WAVES>(pkA + woff, pkG + woff, L::SEG_KB, GATE ? L::SEG_KB : 0, slot_w(j), wave, lane);
        glds_rows<IO>(up ? res : xa, row0_wave, a.M, d, ss * G::FE, slot_ta(j), wave, lane);
        if constexpr (GATE) {
            if (!up) glds_rows<IO>(xg, row0_wave, a.M, d, ss * G::FE, slot_tg(j), wave, lane);
        }
    };

    issue(0);
    {   // biases -> LDS: [bdA(32RT) | buA(d) | bdG(32RT) | buG(d)]
        const float* ba = reinterpret_cast<const float*>(a.pk_a + pg.bias_off);
        for (int i = tid; i < nb; i += WAVES * 64) sb[i] = ba[i];
        if constexpr (GATE) {
            const float* bg = reinterpret_cast<const float*>(a.pk_g + pg.bias_off);
            for (int i = tid; i < nb; i += WAVES * 64) sb[nb + i] = bg[i];
        }
    }
    __syncthreads();

    // ---- down projections: pre[c] for c = 32K + 8g + 4e + rho
    f32x4 accA[RT][2];
    f32x4 accG[GATE ? RT : 1][2];
#pragma unroll
    for (int K = 0; K < RT; ++K) { accA[K][0] = zero4(); accA[K][1] = zero4(); }
    if constexpr (GATE) {
#pragma unroll
        for (int K = 0; K < RT; ++K) { accG[K][0] = zero4(); accG[K][1] = zero4(); }
    }
    int s = 0;
    for (; s < S; ++s) {
        issue(s + 1);
        const uint8_t* w = slot_w(s & 1);
        const uint8_t* ta = slot_ta(s & 1);
        const uint8_t* tg = slot_tg(s & 1);
#pragma unroll
        for (int u = 0; u < G::KS; ++u) {
            Frag<NS> bA = tile_bfrag<IO>(ta, trow, g, u);
            if constexpr (DROP) {
                const int64_t grow = (row0_wave + m < a.M) ? row0_wave + m : a.M - 1;
                const uint64_t kp = *reinterpret_cast<const uint64_t*>(a.keep + grow * d + s * G::FE + 32 * u + 8 * g);
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = (float)bA.p[0][j];
                    if constexpr (NS == 2) v[j] += (float)bA.p[1][j];
                    v[j] = ((kp >> (8 * j)) & 0xff) ? v[j] * a.keep_scale : 0.f;
                }
                bA = frag_from_f32<NS>(v);
            }
#pragma unroll
            for (int K = 0; K < RT; ++K) {
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    accA[K][e] = mfma16_ns<NS>(wfrag<NS>(w, (u * RT + K) * 2 + e, lane), bA, accA[K][e]);
            }
            if constexpr (GATE) {
                const Frag<NS> bG = tile_bfrag<IO>(tg, trow, g, u);
#pragma unroll
                for (int K = 0; K < RT; ++K) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        accG[K][e] = mfma16_ns<NS>(wfrag<NS>(w, L::SEG_FR + (u * RT + K) * 2 + e, lane), bG, accG[K][e]);
                }
            }
        }
        __syncthreads();
    }

    // ---- bias + activation -> B fragments (k-step K holds c = 32K + 8g + j, j = 4e + rho)
    Frag<NS> zA[RT];
    Frag<NS> zG[GATE ? RT : 1];
    {
        const float* bdA = sb + 8 * g;
        const float* bdG = sb + nb + 8 * g;
#pragma unroll
        for (int K = 0; K < RT; ++K) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float pre = accA[K][j >> 2][j & 3] + bdA[32 * K + j];
                v[j] = ACT_ID ? pre : gelu_new_f(pre);
            }
            zA[K] = frag_from_f32<NS>(v);
            if constexpr (GATE) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = gelu_new_f(accG[K][j >> 2][j & 3] + bdG[32 * K + j]);
                zG[K] = frag_from_f32<NS>(v);
            }
        }
    }

    // ---- up projections + epilogue: FE features per stage, LW contiguous per lane
    const float* buA = sb + 32 * RT + G::LW * g;
    const float* buG = sb + nb + 32 * RT + G::LW * g;
    const float s2 = a.s2, sd_ = a.sd, gs = a.gs;
    const bool gate_add = (a.flags & PET_GATE_ADD) != 0;
    for (; s < 2 * S; ++s) {
        issue(s + 1);
        const int su = s - S;
        const uint8_t* w = slot_w(s & 1);
        const uint8_t* tr = slot_ta(s & 1);
        float r[G::LW], o[G::LW];
        tile_lane_vals<IO>(tr, trow, g, r);
#pragma unroll
        for (int q = 0; q < G::NQ; ++q) {
            f32x4 aA = zero4(), aG = zero4();
#pragma unroll
            for (int K = 0; K < RT; ++K) aA = mfma16_ns<NS>(wfrag<NS>(w, q * RT + K, lane), zA[K], aA);
            if constexpr (GATE) {
#pragma unroll
                for (int K = 0; K < RT; ++K)
                    aG = mfma16_ns<NS>(wfrag<NS>(w, L::SEG_FR + q * RT + K, lane), zG[K], aG);
            }
#pragma unroll
            for (int rho = 0; rho < 4; ++rho) {
                const int i = 4 * q + rho;
                float hv = s2 * r[i] + sd_ * (aA[rho] + buA[su * G::FE + i]);
                if constexpr (GATE) {
                    const float gt = sigmoid_f(aG[rho] + buG[su * G::FE + i]);
                    hv = gate_add ? hv + gt : hv * gt;
                    hv *= gs;
                }
                o[i] = hv;
            }
        }
        stage_lane_vals<IO>(staging, trow, g, o);
        store_rows<IO>(out, row0_wave, a.M, d, su * G::FE, staging, wave, lane);
        __syncthreads();
    }
}

template <typename IO, int RT, bool GATE, bool ACT_ID, bool DROP, int WAVES>
static hipError_t launch_one(const PetFwdArgs& a, hipStream_t stream) {
    using L = FwdLds<IO, RT, GATE, WAVES>;
    const size_t lds = L::bytes(a.d);
    auto kern = pet_fwd_kernel<IO, RT, GATE, ACT_ID, DROP, WAVES>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int rows = WAVES * 16;
    const int blocks = (int)((a.M + rows - 1) / rows);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

template <typename IO, int RT, bool GATE, bool ACT_ID, bool DROP>
static hipError_t launch_waves(const PetFwdArgs& a, hipStream_t stream) {
    // 8 waves (128 rows) per workgroup unless the two-slot ring would not fit the 160 KiB LDS
    if constexpr (FwdLds<IO, RT, GATE, 8>::BIAS_OFF + 16 * 1024 <= 160 * 1024)
        return launch_one<IO, RT, GATE, ACT_ID, DROP, 8>(a, stream);
    else
        return launch_one<IO, RT, GATE, ACT_ID, DROP, 4>(a, stream);
}

template <typename IO, int RT>
static hipError_t launch_rt(const PetFwdArgs& a, hipStream_t stream) {
    const bool gate = a.flags & PET_GATE, act_id = a.flags & PET_ACT_IDENTITY, drop = a.keep != nullptr;
    if (gate) return launch_waves<IO, RT, true, false, false>(a, stream);
    if (act_id) return drop ? launch_waves<IO, RT, false, true, true>(a, stream)
                            : launch_waves<IO, RT, false, true, false>(a, stream);
    return launch_waves<IO, RT, false, false, false>(a, stream);
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
