// FFN activation + dropout of the frozen backbone, the step between fc1 and fc2 in the sublayer that K1 / K5 close:
//     out = dropout(act(x), p)           act = gelu (erf form, F.gelu) | gelu_new (tanh form) | relu
// (my_transformers/modeling_bart.py:1264-1265, 1750-1756; my_transformers/modeling_t5.py:262-265).  In the reference this
// is two elementwise passes forward (activation, dropout) and two backward, and autograd keeps x, act(x) and a byte mask
// alive; here one pass each way over [M, d_ff], only x is kept, and the mask comes from the same counter-based generator
// as K5 / K3 (rng.h): group of 8 consecutive elements -> one Philox call, regenerated in the backward.
// Pure streaming: 2 B in + 2 B out per element forward, 4 B in + 2 B out backward (bf16).
#include "common.h"
#include "kernels.h"
#include "rng.h"
#include "vec8.h"

// The pass is as much instruction-bound as memory-bound (an element has ~30 VALU slots at the HBM rate), so the
// transcendental forms are the cheap ones: one v_exp_f32 and one v_rcp_f32 per element.
//   erf form:  Phi(x) = 1 - h (x >= 0), h (x < 0),  h = 0.5 * poly(t) * exp(-x^2 / 2),  t = 1 / (1 + p |x| / sqrt(2))
//              (Abramowitz & Stegun 7.1.26: |error of erf| <= 1.5e-7), and the same exponential is the density of the derivative;
//   tanh form: 0.5 (1 + tanh(u)) = 1 / (1 + exp(-2 u)).
struct GeluParts { float cdf, e; };           // Phi(x) and exp(-x^2 / 2)
__device__ __forceinline__ GeluParts gelu_erf_parts(float x) {
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f * 0.7071067811865476f, ax, 1.0f));
    const float e = __builtin_amdgcn_exp2f(-0.5f * 1.4426950408889634f * x * x);
    float q = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    q = __builtin_fmaf(q, t, 1.421413741f);
    q = __builtin_fmaf(q, t, -0.284496736f);
    q = __builtin_fmaf(q, t, 0.254829592f);
    const float h = 0.5f * q * t * e;
    return {x >= 0.f ? 1.0f - h : h, e};
}
__device__ __forceinline__ float sigmoid_fast(float v) {      // 1 / (1 + exp(-v)); exp overflow -> 1 / inf = 0
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}

template <int ACT> __device__ __forceinline__ float act_f(float x) {
    if constexpr (ACT == VLPET_ACT_RELU) return x > 0.f ? x : 0.f;
    else if constexpr (ACT == VLPET_ACT_GELU_NEW) {
        const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
        return x * sigmoid_fast(2.0f * u);
    } else return x * gelu_erf_parts(x).cdf;
}
template <int ACT> __device__ __forceinline__ float act_df(float x) {
    if constexpr (ACT == VLPET_ACT_RELU) return x > 0.f ? 1.f : 0.f;
    else if constexpr (ACT == VLPET_ACT_GELU_NEW) {
        const float x2 = x * x;
        const float u = 0.7978845608028654f * (x + 0.044715f * x * x2);
        const float sg = sigmoid_fast(2.0f * u);              // = 0.5 (1 + tanh u);  1 - tanh^2 u = 4 sg (1 - sg)
        return sg + x * 2.0f * sg * (1.0f - sg) * 0.7978845608028654f * (1.0f + 0.134145f * x2);
    } else {
        const GeluParts g = gelu_erf_parts(x);
        return g.cdf + x * 0.3989422804014327f * g.e;
    }
}

template <typename IO, int ACT, bool DROP, bool BWD>
__global__ __launch_bounds__(256) void act_dropout_kernel(ActDropArgs a) {
    const int64_t groups = a.n >> 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint64_t seed = DROP ? vlpet_eff_seed(a.seed, a.seed_ctr) : 0;
    // two groups per thread and iteration, both loads (four in the backward) issued before the arithmetic of either
    for (int64_t g0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g0 < groups; g0 += 2 * stride) {
        const int64_t g1 = g0 + stride;
        const bool two = g1 < groups;
        Vec8<IO> x[2], dy[2];
        x[0].load(a.x, g0);
        if (two) x[1].load(a.x, g1);
        if constexpr (BWD) {
            dy[0].load(a.dy, g0);
            if (two) dy[1].load(a.dy, g1);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const int64_t g = u == 0 ? g0 : g1;
            Vec8<IO> o;
            uint32_t bits = 0xffu;
            if constexpr (DROP) bits = keep8(g, seed, a.thr);
            if constexpr (BWD) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float k = ((bits >> j) & 1u) ? a.keep_scale : 0.f;
                    o.set(j, dy[u].get(j) * k * act_df<ACT>(x[u].get(j)));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float k = ((bits >> j) & 1u) ? a.keep_scale : 0.f;
                    o.set(j, act_f<ACT>(x[u].get(j)) * k);
                }
                if (a.keep_out != nullptr) drop_export8(a.keep_out, g << 3, bits);
            }
            o.store(a.out, g);
        }
    }
}

template <typename IO, int ACT>
static hipError_t launch_act(const ActDropArgs& a, bool bwd, hipStream_t stream) {
    const int64_t groups = a.n >> 3;
    if (groups == 0) return hipSuccess;
    int64_t blocks = (groups + 255) / 256;
    const int64_t cap = 256 * 16;                     // 16 workgroups per CU, grid-stride beyond
    if (blocks > cap) blocks = cap;
    const bool drop = a.thr != 0;
    const dim3 grid((unsigned)blocks), blk(256);
    if (bwd) {
        if (drop) hipLaunchKernelGGL((act_dropout_kernel<IO, ACT, true, true>), grid, blk, 0, stream, a);
        else hipLaunchKernelGGL((act_dropout_kernel<IO, ACT, false, true>), grid, blk, 0, stream, a);
    } else {
        if (drop) hipLaunchKernelGGL((act_dropout_kernel<IO, ACT, true, false>), grid, blk, 0, stream, a);
        else hipLaunchKernelGGL((act_dropout_kernel<IO, ACT, false, false>), grid, blk, 0, stream, a);
    }
    return hipGetLastError();
}

template <typename IO>
static hipError_t launch_act_io(const ActDropArgs& a, bool bwd, hipStream_t stream) {
    switch (a.act) {
        case VLPET_ACT_GELU: return launch_act<IO, VLPET_ACT_GELU>(a, bwd, stream);
        case VLPET_ACT_GELU_NEW: return launch_act<IO, VLPET_ACT_GELU_NEW>(a, bwd, stream);
        case VLPET_ACT_RELU: return launch_act<IO, VLPET_ACT_RELU>(a, bwd, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_act_dropout(const ActDropArgs& a, bool bwd, int io_fp32, hipStream_t stream) {
    return io_fp32 ? launch_act_io<float>(a, bwd, stream) : launch_act_io<__bf16>(a, bwd, stream);
}

// ------------------------------------------------------------------------------------------------ joint-encoder input assembly
// x = dropout(cat([a, v], dim = 1), p): a [B, La, d] (the embedded, layer-normed text), v [B, Lv, d] (the visual embedding, K4's output),
// x [B, La + Lv, d] -- src/modeling_bart.py:804-820 (JointEncoder.forward: inputs_embeds = cat([inputs_embeds, vis_embeds], dim=1), then
// F.dropout); T5: src/modeling_t5.py:263, 300.  The library chain is a concatenation pass, a dropout pass that also writes a byte mask,
// and in the backward a masked-scale pass plus one copy per slice of the gradient; here one pass each way, the mask regenerated
// (rng.h, keyed by the element index of x).  BWD: dx -> da, dv (either may be nullptr: no gradient wanted).
struct CatDropArgs {
    const void* a; const void* v; void* x;      // forward: a, v in, x out.  backward: x = dx in, a / v = da / dv out
    int64_t B; int La, Lv, d;
    uint32_t thr; float keep_scale; uint64_t seed; const uint64_t* seed_ctr;
};
template <typename IO, bool DROP, bool BWD>
__global__ __launch_bounds__(256) void cat_dropout_kernel(CatDropArgs c) {
    const int gpr = c.d >> 3, L = c.La + c.Lv;                     // groups of 8 elements per row
    const int64_t groups = c.B * (int64_t)L * gpr;
    const uint64_t seed = DROP ? vlpet_eff_seed(c.seed, c.seed_ctr) : 0;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = g / gpr;
        const int col = (int)(g - row * gpr);
        const int64_t b = row / L;
        const int t = (int)(row - b * L);
        const bool text = t < c.La;
        const int64_t sg = text ? (b * c.La + t) * (int64_t)gpr + col : (b * c.Lv + (t - c.La)) * (int64_t)gpr + col;     // group index inside a / v
        void* side = const_cast<void*>(text ? c.a : c.v);
        uint32_t bits = 0xffu;
        if constexpr (DROP) bits = keep8(g, seed, c.thr);
        Vec8<IO> in, o;
        if constexpr (BWD) {
            if (side == nullptr) continue;
            in.load(c.x, g);
        } else in.load(side, sg);
#pragma unroll
        for (int j = 0; j < 8; ++j) o.set(j, ((bits >> j) & 1u) ? in.get(j) * c.keep_scale : 0.f);
        if constexpr (BWD) o.store(side, sg);
        else o.store(c.x, g);
    }
}
hipError_t launch_cat_dropout(const void* a, const void* v, void* x, int64_t B, int La, int Lv, int d, uint32_t thr, float keep_scale,
                              uint64_t seed, const uint64_t* seed_ctr, bool bwd, int io_fp32, hipStream_t stream) {
    CatDropArgs c{a, v, x, B, La, Lv, d, thr, keep_scale, seed, seed_ctr};
    const int64_t groups = B * (int64_t)(La + Lv) * (d >> 3);
    if (groups == 0) return hipSuccess;
    int64_t blocks = (groups + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const dim3 grid((unsigned)blocks), blk(256);
    const bool drop = thr != 0;
#define VLPET_CD(IO) \
    if (bwd) { if (drop) hipLaunchKernelGGL((cat_dropout_kernel<IO, true, true>), grid, blk, 0, stream, c); \
               else hipLaunchKernelGGL((cat_dropout_kernel<IO, false, true>), grid, blk, 0, stream, c); } \
    else { if (drop) hipLaunchKernelGGL((cat_dropout_kernel<IO, true, false>), grid, blk, 0, stream, c); \
           else hipLaunchKernelGGL((cat_dropout_kernel<IO, false, false>), grid, blk, 0, stream, c); }
    if (io_fp32) { VLPET_CD(float) } else { VLPET_CD(__bf16) }
#undef VLPET_CD
    return hipGetLastError();
}
