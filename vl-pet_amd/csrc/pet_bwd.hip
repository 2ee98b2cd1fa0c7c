// Fused PET backward, row-parallel part (same v3 structure as pet_fwd.hip, see pet16.h):
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
// Stage stream (each stage = one fill of a ring slot: <= 8RT KiB of weight fragments + <= 2 row tiles):
//   gate:     [down A|G + xa,xg] x S   [ (up A|G + res,dy), (up_t A|G) ] x S   [down_t A|G + dh] x S
//   no gate:  [down A + xa] x S        [ up_t A + dy ] x S                      [down_t A] x S
#include "common.h"
#include "kernels.h"
#include "pet16.h"

template <typename IO, int RT, bool GATE, int WAVES>
struct BwdLds {
    static constexpr int NS = Geo<IO>::NS;
    static constexpr int SEG_KB = 4 * RT;
    static constexpr int SEG_FR = SEG_KB / NS;
    static constexpr int W_B = SEG_KB * 1024 * (GATE ? 2 : 1);
    static constexpr int TILE_B = WAVES * 16 * 128;
    static constexpr int SLOT_B = W_B + TILE_B * (GATE ? 2 : 1);
    static constexpr int STAGING_OFF = 2 * SLOT_B;
    static constexpr int BIAS_OFF = STAGING_OFF + TILE_B * (GATE ? 2 : 1);
    static size_t bytes(int d) { return (size_t)BIAS_OFF + (size_t)2 * (32 * RT + d) * 4; }
};

template <typename IO, int RT, bool GATE, bool ACT_ID, bool DROP, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void pet_bwd_kernel(PetBwdArgs a) {
    using G = Geo<IO>;
    using L = BwdLds<IO, RT, GATE, WAVES>;
    constexpr int NS = G::NS;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int trow = 16 * wave + m;
    const int d = a.d;
    const int64_t row0_wave = (int64_t)blockIdx.x * (WAVES * 16) + wave * 16;
    const int64_t grow_raw = row0_wave + m;
    const bool row_ok = grow_raw < a.M;
    const int64_t grow = row_ok ? grow_raw : a.M - 1;
    const int S = d / G::FE;
    const PackGeom pg = pack_geom(RT, d, NS);
    const uint8_t* pkA = a.pk_a;
    const uint8_t* pkG = GATE ? a.pk_g : a.pk_a;
    const IO* xa = reinterpret_cast<const IO*>(a.xa);
    const IO* xg = reinterpret_cast<const IO*>(a.xg);
    const IO* res = reinterpret_cast<const IO*>(a.res);
    const IO* dy = reinterpret_cast<const IO*>(a.dy);
    IO* DH = reinterpret_cast<IO*>(a.dh);
    IO* DQ = reinterpret_cast<IO*>(a.dq);

    auto slot_w = [&](int j) { return smem + (size_t)j * L::SLOT_B; };
    auto slot_t0 = [&](int j) { return smem + (size_t)j * L::SLOT_B + L::W_B; };
    auto slot_t1 = [&](int j) { return smem + (size_t)j * L::SLOT_B + L::W_B + L::TILE_B; };
    uint8_t* stg0 = smem + L::STAGING_OFF;
    uint8_t* stg1 = smem + L::STAGING_OFF + L::TILE_B;
    float* sb = reinterpret_cast<float*>(smem + L::BIAS_OFF);
    const int nb = 32 * RT + d;
    const int NU = GATE ? 2 * S : S;            // stages of the middle phase
    const int total = S + NU + S;

    auto issue = [&](int s) {
        if (s >= total) return;
        const int j = s & 1;
        int pack, ss;
        const IO* t0 = nullptr;
        const IO* t1 = nullptr;
        if (s < S) { pack = 0; ss = s; t0 = xa; if constexpr (GATE) t1 = xg; }
        else if (s < S + NU) {
            const int u = s - S;
            if constexpr (GATE) {
                ss = u >> 1;
                if (u & 1) { pack = 2; }
                else { pack = 1; t0 = res; t1 = dy; }
            } else { ss = u; pack = 2; t0 = dy; }
        } else { pack = 3; ss = s - S - NU; if constexpr (GATE) t0 = DH; }
        const int64_t woff = (int64_t)pack * pg.pack_bytes + (int64_t)ss * L::SEG_KB * 1024;
        glds_weights<WAVES>(pkA + woff, pkG + woff, L::SEG_KB, GATE ? L::SEG_KB : 0, slot_w(j), wave, lane);
        if (t0) glds_rows<IO>(t0, row0_wave, a.M, d, ss * G::FE, slot_t0(j), wave, lane);
        if constexpr (GATE) {
            if (t1) glds_rows<IO>(t1, row0_wave, a.M, d, ss * G::FE, slot_t1(j), wave, lane);
        }
    };

    issue(0);
    {
        const float* ba = reinterpret_cast<const float*>(a.pk_a + pg.bias_off);
        for (int i = tid; i < nb; i += WAVES * 64) sb[i] = ba[i];
        if constexpr (GATE) {
            const float* bg = reinterpret_cast<const float*>(a.pk_g + pg.bias_off);
            for (int i = tid; i < nb; i += WAVES * 64) sb[nb + i] = bg[i];
        }
    }
    __syncthreads();

    // ---- phase 1: recompute the bottleneck pre-activations of both chains
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
#pragma unroll
        for (int u = 0; u < G::KS; ++u) {
            Frag<NS> bA = tile_bfrag<IO>(slot_t0(s & 1), trow, g, u);
            if constexpr (DROP) {
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
                const Frag<NS> bG = tile_bfrag<IO>(slot_t1(s & 1), trow, g, u);
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

    // z = act(pre) as B fragments; accA / accG are overwritten with act'(pre)
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
                accA[K][j >> 2][j & 3] = ACT_ID ? 1.0f : gelu_new_grad_f(pre);
            }
            zA[K] = frag_from_f32<NS>(v);
            if constexpr (GATE) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pre = accG[K][j >> 2][j & 3] + bdG[32 * K + j];
                    v[j] = gelu_new_f(pre);
                    accG[K][j >> 2][j & 3] = gelu_new_grad_f(pre);
                }
                zG[K] = frag_from_f32<NS>(v);
            }
        }
    }

    // ---- phase 2: elementwise backward per feature group, then contraction over features
    const float* buA = sb + 32 * RT + G::LW * g;
    const float* buG = sb + nb + 32 * RT + G::LW * g;
    const float s2 = a.s2, sd_ = a.sd, gs = a.gs;
    const bool gate_add = (a.flags & PET_GATE_ADD) != 0;
    f32x4 dzA[RT][2];
    f32x4 dzG[GATE ? RT : 1][2];
#pragma unroll
    for (int K = 0; K < RT; ++K) { dzA[K][0] = zero4(); dzA[K][1] = zero4(); }
    if constexpr (GATE) {
#pragma unroll
        for (int K = 0; K < RT; ++K) { dzG[K][0] = zero4(); dzG[K][1] = zero4(); }
    }
    for (int su = 0; su < S; ++su) {
        Frag<NS> dfA[G::E2], dfG[G::E2];
        if constexpr (GATE) {
            issue(s + 1);
            const uint8_t* w = slot_w(s & 1);
            float r[G::LW], dyv[G::LW], dh[G::LW], dq[G::LW], dd[G::LW];
            tile_lane_vals<IO>(slot_t0(s & 1), trow, g, r);
            tile_lane_vals<IO>(slot_t1(s & 1), trow, g, dyv);
#pragma unroll
            for (int q = 0; q < G::NQ; ++q) {
                f32x4 aA = zero4(), aG = zero4();
#pragma unroll
                for (int K = 0; K < RT; ++K) aA = mfma16_ns<NS>(wfrag<NS>(w, q * RT + K, lane), zA[K], aA);
#pragma unroll
                for (int K = 0; K < RT; ++K) aG = mfma16_ns<NS>(wfrag<NS>(w, L::SEG_FR + q * RT + K, lane), zG[K], aG);
#pragma unroll
                for (int rho = 0; rho < 4; ++rho) {
                    const int i = 4 * q + rho;
                    const float hv = s2 * r[i] + sd_ * (aA[rho] + buA[su * G::FE + i]);
                    const float gt = sigmoid_f(aG[rho] + buG[su * G::FE + i]);
                    const float dyp = gs * dyv[i];
                    dh[i] = gate_add ? dyp : dyp * gt;
                    const float dg = gate_add ? dyp : dyp * hv;
                    dq[i] = dg * gt * (1.0f - gt);
                    dd[i] = sd_ * dh[i];
                }
            }
            stage_lane_vals<IO>(stg0, trow, g, dh);
            stage_lane_vals<IO>(stg1, trow, g, dq);
            store_rows<IO>(DH, row0_wave, a.M, d, su * G::FE, stg0, wave, lane);
            store_rows<IO>(DQ, row0_wave, a.M, d, su * G::FE, stg1, wave, lane);
#pragma unroll
            for (int e2 = 0; e2 < G::E2; ++e2) {
                dfA[e2] = frag_from_f32<NS>(dd + 8 * e2);
                dfG[e2] = frag_from_f32<NS>(dq + 8 * e2);
            }
            __syncthreads();
            ++s;
        }
        {
            issue(s + 1);
            const uint8_t* w = slot_w(s & 1);
            if constexpr (!GATE) {
                float dyv[G::LW];
                tile_lane_vals<IO>(slot_t0(s & 1), trow, g, dyv);
#pragma unroll
                for (int i = 0; i < G::LW; ++i) dyv[i] *= sd_;
#pragma unroll
                for (int e2 = 0; e2 < G::E2; ++e2) dfA[e2] = frag_from_f32<NS>(dyv + 8 * e2);
            }
#pragma unroll
            for (int e2 = 0; e2 < G::E2; ++e2) {
#pragma unroll
                for (int K = 0; K < RT; ++K) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        dzA[K][e] = mfma16_ns<NS>(wfrag<NS>(w, (e2 * RT + K) * 2 + e, lane), dfA[e2], dzA[K][e]);
                }
                if constexpr (GATE) {
#pragma unroll
                    for (int K = 0; K < RT; ++K) {
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                            dzG[K][e] = mfma16_ns<NS>(wfrag<NS>(w, L::SEG_FR + (e2 * RT + K) * 2 + e, lane), dfG[e2], dzG[K][e]);
                    }
                }
            }
            __syncthreads();
            ++s;
        }
    }

    // ---- dpre = dz * act'(pre); row-major side products for the weight-gradient kernel
    const int ldz = 32 * RT;
    Frag<NS> dpA[RT];
    Frag<NS> dpG[GATE ? RT : 1];
#pragma unroll
    for (int K = 0; K < RT; ++K) {
        float v[8], zv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = dzA[K][j >> 2][j & 3] * accA[K][j >> 2][j & 3];
        dpA[K] = frag_from_f32<NS>(v);
        if (row_ok) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                zv[j] = (float)zA[K].p[0][j];
                if constexpr (NS == 2) zv[j] += (float)zA[K].p[1][j];
            }
            store8_f32(reinterpret_cast<IO*>(a.z_a) + grow * ldz + 32 * K + 8 * g, zv);
            store8_f32(reinterpret_cast<IO*>(a.dp_a) + grow * ldz + 32 * K + 8 * g, v);
        }
        if constexpr (GATE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = dzG[K][j >> 2][j & 3] * accG[K][j >> 2][j & 3];
            dpG[K] = frag_from_f32<NS>(v);
            if (row_ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    zv[j] = (float)zG[K].p[0][j];
                    if constexpr (NS == 2) zv[j] += (float)zG[K].p[1][j];
                }
                store8_f32(reinterpret_cast<IO*>(a.z_g) + grow * ldz + 32 * K + 8 * g, zv);
                store8_f32(reinterpret_cast<IO*>(a.dp_g) + grow * ldz + 32 * K + 8 * g, v);
            }
        }
    }

    // ---- phase 3: input gradients
    IO* dxa = reinterpret_cast<IO*>(a.dxa);
    IO* dxg = reinterpret_cast<IO*>(a.dxg);
    for (int su = 0; su < S; ++su, ++s) {
        issue(s + 1);
        const uint8_t* w = slot_w(s & 1);
        float oa[G::LW], og[G::LW], dhv[G::LW];
        if constexpr (GATE) tile_lane_vals<IO>(slot_t0(s & 1), trow, g, dhv);
        uint64_t kp[2] = {0, 0};
        if constexpr (DROP) {
            // keep mask of the lane's LW features (uint8 each)
            const uint8_t* kr = a.keep + grow * d + su * G::FE + G::LW * g;
            kp[0] = *reinterpret_cast<const uint64_t*>(kr);
            if constexpr (G::LW == 16) kp[1] = *reinterpret_cast<const uint64_t*>(kr + 8);
        }
#pragma unroll
        for (int q = 0; q < G::NQ; ++q) {
            f32x4 aA = zero4(), aG = zero4();
#pragma unroll
            for (int K = 0; K < RT; ++K) aA = mfma16_ns<NS>(wfrag<NS>(w, q * RT + K, lane), dpA[K], aA);
            if constexpr (GATE) {
#pragma unroll
                for (int K = 0; K < RT; ++K) aG = mfma16_ns<NS>(wfrag<NS>(w, L::SEG_FR + q * RT + K, lane), dpG[K], aG);
            }
#pragma unroll
            for (int rho = 0; rho < 4; ++rho) {
                const int i = 4 * q + rho;
                float v = aA[rho];
                if constexpr (GATE) v += s2 * dhv[i];
                if constexpr (DROP) v = ((kp[i >> 3] >> (8 * (i & 7))) & 0xff) ? v * a.keep_scale : 0.f;
                oa[i] = v;
                og[i] = aG[rho];
            }
        }
        stage_lane_vals<IO>(stg0, trow, g, oa);
        store_rows<IO>(dxa, row0_wave, a.M, d, su * G::FE, stg0, wave, lane);
        if constexpr (GATE) {
            stage_lane_vals<IO>(stg1, trow, g, og);
            store_rows<IO>(dxg, row0_wave, a.M, d, su * G::FE, stg1, wave, lane);
        }
        __syncthreads();
    }
}

template <typename IO, int RT, bool GATE, bool ACT_ID, bool DROP, int WAVES>
static hipError_t launch_one(const PetBwdArgs& a, hipStream_t stream) {
    using L = BwdLds<IO, RT, GATE, WAVES>;
    const size_t lds = L::bytes(a.d);
    auto kern = pet_bwd_kernel<IO, RT, GATE, ACT_ID, DROP, WAVES>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int rows = WAVES * 16;
    const int blocks = (int)((a.M + rows - 1) / rows);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

template <typename IO, int RT, bool GATE, bool ACT_ID, bool DROP>
static hipError_t launch_waves(const PetBwdArgs& a, hipStream_t stream) {
    if constexpr (BwdLds<IO, RT, GATE, 8>::BIAS_OFF + 16 * 1024 <= 160 * 1024)
        return launch_one<IO, RT, GATE, ACT_ID, DROP, 8>(a, stream);
    else
        return launch_one<IO, RT, GATE, ACT_ID, DROP, 4>(a, stream);
}

template <typename IO, int RT>
static hipError_t launch_rt(const PetBwdArgs& a, hipStream_t stream) {
    const bool gate = a.flags & PET_GATE, act_id = a.flags & PET_ACT_IDENTITY, drop = a.keep != nullptr;
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
