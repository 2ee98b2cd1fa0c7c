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

template <int ACT> __device__ __forceinline__ float act_f(float x) {
    if constexpr (ACT == VLPET_ACT_RELU) return x > 0.f ? x : 0.f;
    else if constexpr (ACT == VLPET_ACT_GELU_NEW) {
        const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
        return 0.5f * x * (1.0f + tanhf(u));
    } else return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
}
template <int ACT> __device__ __forceinline__ float act_df(float x) {
    if constexpr (ACT == VLPET_ACT_RELU) return x > 0.f ? 1.f : 0.f;
    else if constexpr (ACT == VLPET_ACT_GELU_NEW) {
        const float x2 = x * x;
        const float u = 0.7978845608028654f * (x + 0.044715f * x * x2);
        const float t = tanhf(u);
        return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 0.134145f * x2);
    } else {
        const float cdf = 0.5f * (1.0f + erff(x * 0.7071067811865476f));
        return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
    }
}

template <typename IO, int ACT, bool DROP, bool BWD>
__global__ __launch_bounds__(256) void act_dropout_kernel(ActDropArgs a) {
    const int64_t groups = a.n >> 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        Vec8<IO> x, o;
        x.load(a.x, g);
        uint32_t bits = 0xffu;
        if constexpr (DROP) bits = keep8(g, a.seed, a.thr);
        if constexpr (BWD) {
            Vec8<IO> dy;
            dy.load(a.dy, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float k = ((bits >> j) & 1u) ? a.keep_scale : 0.f;
                o.set(j, dy.get(j) * k * act_df<ACT>(x.get(j)));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float k = ((bits >> j) & 1u) ? a.keep_scale : 0.f;
                o.set(j, act_f<ACT>(x.get(j)) * k);
            }
            if (a.keep_out != nullptr) drop_export8(a.keep_out, g << 3, bits);
        }
        o.store(a.out, g);
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
