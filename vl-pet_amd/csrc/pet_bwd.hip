// Fused PET backward, row-parallel part (same tiling as pet_fwd.hip):
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
#include "common.h"
#include "kernels.h"
#include "pet_phases.h"

template <int NS, int RT, bool GATE>
struct BwdCtx {
    static constexpr int FB = NS * 1024;
    static constexpr int STAGE_B = 4 * RT * FB;
    static constexpr int HALF_B = 2 * RT * FB;
    const uint8_t* pk_a;
    const uint8_t* pk_g;
    uint8_t* smem;
    int64_t pack_bytes;
    int tid, T, NT;
    __device__ __forceinline__ uint8_t* buf(int i) const { return smem + i * STAGE_B; }
    // stage stream: [down A: T] [down G: T] [per n-tile: (up A|up G), (up_t A|up_t G)] [per n-tile: (down_t A|down_t G)]
    // without gate:  [down A: T] [per n-tile: up_t A] [per n-tile: down_t A]
    __device__ __forceinline__ StageDesc stage(int s) const {
        StageDesc r{pk_a, 0, pk_a, 0};
        if (s < T) { r.p0 = pk_a + (int64_t)s * STAGE_B; r.u0 = STAGE_B / 16; return r; }
        s -= T;
        if constexpr (GATE) {
            if (s < T) { r.p0 = pk_g + (int64_t)s * STAGE_B; r.u0 = STAGE_B / 16; return r; }
            s -= T;
            if (s < 2 * NT) {
                const int nt = s >> 1;
                const int64_t off = ((s & 1) ? 2 : 1) * pack_bytes + (int64_t)nt * HALF_B;
                r.p0 = pk_a + off; r.u0 = HALF_B / 16; r.p1 = pk_g + off; r.u1 = HALF_B / 16;
                return r;
            }
            s -= 2 * NT;
        } else {
            if (s < NT) { r.p0 = pk_a + 2 * pack_bytes + (int64_t)s * HALF_B; r.u0 = HALF_B / 16; return r; }
            s -= NT;
        }
        if (s < NT) {
            const int64_t off = 3 * pack_bytes + (int64_t)s * HALF_B;
            r.p0 = pk_a + off; r.u0 = HALF_B / 16;
            if constexpr (GATE) { r.p1 = pk_g + off; r.u1 = HALF_B / 16; }
        }
        return r;
    }
};

template <typename IO, int RT, bool GATE, bool ACT_ID, bool DROP>
__global__ __launch_bounds__(VLPET_THREADS) void pet_bwd_kernel(PetBwdArgs a) {
    constexpr int NS = IoTraits<IO>::NS;
    constexpr int KT = 2 * RT;
    constexpr int MAXU = RT * NS;
    using Ctx = BwdCtx<NS, RT, GATE>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int d = a.d;
    const int64_t row_raw = (int64_t)blockIdx.x * VLPET_ROWS_PER_WG + wave * 32 + m;
    const bool row_ok = row_raw < a.M;
    const int64_t row = row_ok ? row_raw : a.M - 1;

    const PackGeom g = pack_geom(RT, d, NS);
    Ctx c;
    c.pk_a = a.pk_a; c.pk_g = a.pk_g; c.smem = smem; c.pack_bytes = g.pack_bytes;
    c.tid = tid; c.T = d / 64; c.NT = d / 32;

    float* sb = reinterpret_cast<float*>(smem + 2 * Ctx::STAGE_B);
    const int nb = 32 * RT + d;
    {
        const float* ba = reinterpret_cast<const float*>(a.pk_a + g.bias_off);
        for (int i = tid; i < nb; i += VLPET_THREADS) sb[i] = ba[i];
        if constexpr (GATE) {
            const float* bg = reinterpret_cast<const float*>(a.pk_g + g.bias_off);
            for (int i = tid; i < nb; i += VLPET_THREADS) sb[nb + i] = bg[i];
        }
    }
    {
        StageRegs<MAXU> sr;
        const StageDesc s0 = c.stage(0);
        stage_load<MAXU>(sr, s0.p0, s0.u0, s0.p1, s0.u1, tid);
        stage_store<MAXU>(sr, c.buf(0), s0.u0 + s0.u1, tid);
    }
    __syncthreads();

    int s = 0;
    // ---- recompute the bottleneck activations (+ gelu') of both chains
    const IO* xa = reinterpret_cast<const IO*>(a.xa) + row * d + 32 * h;
    const uint8_t* keeprow = DROP ? a.keep + row * d + 32 * h : nullptr;
    Frag<NS> zA[KT];
    f32x16 gpA[RT];
    down_phase<IO, RT, ACT_ID, true, DROP>(c, s, xa, keeprow, a.keep_scale, sb + 8 * h, lane, zA, gpA);
    Frag<NS> zG[GATE ? KT : 1];
    f32x16 gpG[GATE ? RT : 1];
    if constexpr (GATE) {
        const IO* xg = reinterpret_cast<const IO*>(a.xg) + row * d + 32 * h;
        down_phase<IO, RT, false, true, false>(c, s, xg, nullptr, 1.f, sb + nb + 8 * h, lane, zG, gpG);
    }

    const IO* dy = reinterpret_cast<const IO*>(a.dy) + row * d;
    const float* sbu = sb + 32 * RT;
    const float* sbgu = sb + nb + 32 * RT;
    const float s2 = a.s2, sd = a.sd, gs = a.gs;
    const bool gate_add = (a.flags & PET_GATE_ADD) != 0;

    f32x16 dzA[RT];
    f32x16 dzG[GATE ? RT : 1];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) dzA[ct] = zero16();
    if constexpr (GATE) {
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) dzG[ct] = zero16();
    }

    // ---- per n-tile: (gate) up projections + elementwise backward, then contraction over features
    for (int nt = 0; nt < c.NT; ++nt) {
        const int f0 = 64 * (nt >> 1) + 32 * h + 16 * (nt & 1);
        Frag<NS> dfA[2], dfG[2];
        float dyv[16];
        load8_f32(dy + f0, dyv);
        load8_f32(dy + f0 + 8, dyv + 8);
        if constexpr (GATE) {
            StageRegs<MAXU> sr;
            const StageDesc nx = c.stage(s + 1);
            stage_load<MAXU>(sr, nx.p0, nx.u0, nx.p1, nx.u1, tid);
            const IO* res = reinterpret_cast<const IO*>(a.res) + row * d;
            float r[16];
            load8_f32(res + f0, r);
            load8_f32(res + f0 + 8, r + 8);
            const uint8_t* b = c.buf(s & 1);
            f32x16 aA = zero16(), aG = zero16();
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) aA = mfma_ns<NS>(lds_frag<NS>(b, ks, lane), zA[ks], aA);
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) aG = mfma_ns<NS>(lds_frag<NS>(b, KT + ks, lane), zG[ks], aG);
            float dh[16], dq[16], dd[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float hv = s2 * r[i] + sd * (aA[i] + sbu[f0 + i]);
                const float gt = sigmoid_f(aG[i] + sbgu[f0 + i]);
                const float dyp = gs * dyv[i];
                dh[i] = gate_add ? dyp : dyp * gt;
                const float dg = gate_add ? dyp : dyp * hv;
                dq[i] = dg * gt * (1.0f - gt);
                dd[i] = sd * dh[i];
            }
            if (row_ok) {
                IO* DH = reinterpret_cast<IO*>(a.dh) + row * d + f0;
                IO* DQ = reinterpret_cast<IO*>(a.dq) + row * d + f0;
                store8_f32(DH, dh); store8_f32(DH + 8, dh + 8);
                store8_f32(DQ, dq); store8_f32(DQ + 8, dq + 8);
            }
            dfA[0] = frag_from_f32<NS>(dd); dfA[1] = frag_from_f32<NS>(dd + 8);
            dfG[0] = frag_from_f32<NS>(dq); dfG[1] = frag_from_f32<NS>(dq + 8);
            stage_store<MAXU>(sr, c.buf((s + 1) & 1), nx.u0 + nx.u1, tid);
            __syncthreads();
            ++s;
        } else {
            float dd[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) dd[i] = sd * dyv[i];
            dfA[0] = frag_from_f32<NS>(dd); dfA[1] = frag_from_f32<NS>(dd + 8);
        }
        {
            StageRegs<MAXU> sr;
            const StageDesc nx = c.stage(s + 1);
            stage_load<MAXU>(sr, nx.p0, nx.u0, nx.p1, nx.u1, tid);
            const uint8_t* b = c.buf(s & 1);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int ct = 0; ct < RT; ++ct)
                    dzA[ct] = mfma_ns<NS>(lds_frag<NS>(b, e * RT + ct, lane), dfA[e], dzA[ct]);
            }
            if constexpr (GATE) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
#pragma unroll
                    for (int ct = 0; ct < RT; ++ct)
                        dzG[ct] = mfma_ns<NS>(lds_frag<NS>(b, 2 * RT + e * RT + ct, lane), dfG[e], dzG[ct]);
                }
            }
            stage_store<MAXU>(sr, c.buf((s + 1) & 1), nx.u0 + nx.u1, tid);
            __syncthreads();
            ++s;
        }
    }

    // ---- dpre = dz * act'(pre); write the row-major side products for the weight gradients
    const int ldz = 32 * RT;
    Frag<NS> dpA[KT];
    Frag<NS> dpG[GATE ? KT : 1];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) {
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = dzA[ct][8 * sh + j] * gpA[ct][8 * sh + j];
            dpA[2 * ct + sh] = frag_from_f32<NS>(v);
            if (row_ok) {
                const int col = 32 * ct + 16 * sh + 8 * h;
                float zv[8];
                frag_to_f32<NS>(zA[2 * ct + sh], zv);
                store8_f32(reinterpret_cast<IO*>(a.z_a) + row * ldz + col, zv);
                store8_f32(reinterpret_cast<IO*>(a.dp_a) + row * ldz + col, v);
            }
        }
    }
    if constexpr (GATE) {
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) {
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = dzG[ct][8 * sh + j] * gpG[ct][8 * sh + j];
                dpG[2 * ct + sh] = frag_from_f32<NS>(v);
                if (row_ok) {
                    const int col = 32 * ct + 16 * sh + 8 * h;
                    float zv[8];
                    frag_to_f32<NS>(zG[2 * ct + sh], zv);
                    store8_f32(reinterpret_cast<IO*>(a.z_g) + row * ldz + col, zv);
                    store8_f32(reinterpret_cast<IO*>(a.dp_g) + row * ldz + col, v);
                }
            }
        }
    }

    // ---- input gradients, n-tile by n-tile
    for (int nt = 0; nt < c.NT; ++nt) {
        StageRegs<MAXU> sr;
        const StageDesc nx = c.stage(s + 1);
        stage_load<MAXU>(sr, nx.p0, nx.u0, nx.p1, nx.u1, tid);
        const int f0 = 64 * (nt >> 1) + 32 * h + 16 * (nt & 1);
        float dhv[16];
        if constexpr (GATE) {
            // s2*dh re-read: this lane wrote exactly these 16 elements in the n-tile loop above
            const IO* DH = reinterpret_cast<const IO*>(a.dh) + row * d + f0;
            load8_f32(DH, dhv);
            load8_f32(DH + 8, dhv + 8);
        }
        uint64_t kp0 = 0, kp1 = 0;
        if constexpr (DROP) {
            kp0 = *reinterpret_cast<const uint64_t*>(a.keep + row * d + f0);
            kp1 = *reinterpret_cast<const uint64_t*>(a.keep + row * d + f0 + 8);
        }
        const uint8_t* b = c.buf(s & 1);
        f32x16 aA = zero16(), aG = zero16();
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) aA = mfma_ns<NS>(lds_frag<NS>(b, ks, lane), dpA[ks], aA);
        if constexpr (GATE) {
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) aG = mfma_ns<NS>(lds_frag<NS>(b, KT + ks, lane), dpG[ks], aG);
        }
        float oa[16], og[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = aA[i];
            if constexpr (GATE) v += s2 * dhv[i];
            if constexpr (DROP) {
                const uint64_t k = i < 8 ? kp0 : kp1;
                v = ((k >> (8 * (i & 7))) & 0xff) ? v * a.keep_scale : 0.f;
            }
            oa[i] = v;
            og[i] = aG[i];
        }
        if (row_ok) {
            IO* dxa = reinterpret_cast<IO*>(a.dxa) + row * d + f0;
            store8_f32(dxa, oa); store8_f32(dxa + 8, oa + 8);
            if constexpr (GATE) {
                IO* dxg = reinterpret_cast<IO*>(a.dxg) + row * d + f0;
                store8_f32(dxg, og); store8_f32(dxg + 8, og + 8);
            }
        }
        stage_store<MAXU>(sr, c.buf((s + 1) & 1), nx.u0 + nx.u1, tid);
        __syncthreads();
        ++s;
    }
}

template <typename IO, int RT, bool GATE, bool ACT_ID, bool DROP>
static hipError_t launch_one(const PetBwdArgs& a, hipStream_t stream) {
    constexpr int NS = IoTraits<IO>::NS;
    const size_t lds = 2 * (size_t)BwdCtx<NS, RT, GATE>::STAGE_B + (size_t)2 * (32 * RT + a.d) * 4;
    auto kern = pet_bwd_kernel<IO, RT, GATE, ACT_ID, DROP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int blocks = (int)((a.M + VLPET_ROWS_PER_WG - 1) / VLPET_ROWS_PER_WG);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(VLPET_THREADS), lds, stream, a);
    return hipGetLastError();
}

template <typename IO, int RT>
static hipError_t launch_rt(const PetBwdArgs& a, hipStream_t stream) {
    const bool gate = a.flags & PET_GATE, act_id = a.flags & PET_ACT_IDENTITY, drop = a.keep != nullptr;
    if (gate) return launch_one<IO, RT, true, false, false>(a, stream);
    if (act_id) return drop ? launch_one<IO, RT, false, true, true>(a, stream)
                            : launch_one<IO, RT, false, true, false>(a, stream);
    return launch_one<IO, RT, false, false, false>(a, stream);
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
