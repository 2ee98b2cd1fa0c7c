// Fused optimizer step over the flat trainable buffer: global-norm clip + AdamW in two
// launches instead of ~230 per-tensor kernels.  Reference semantics: torch.nn.utils.clip_grad_norm_(params, 5.0)
// (multitask.py:279-300) followed by transformers' AdamW as configured in trainer_base.py:633-701
// (eps 1e-6, weight decay 0.01 except bias / LayerNorm.weight, bias correction on):
//     g   <- g * grad_scale * min(1, max_norm / (||g * grad_scale|| + 1e-6))
//     m   <- b1 m + (1-b1) g ;  v <- b2 v + (1-b2) g^2
//     p   <- p - lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)  ;  p <- p - lr * wd * p      (decay after the update)
// (transformers.optimization.AdamW order; torch.optim.AdamW decays before the update -- selectable).
// HBM-bound streaming kernels: 16-byte accesses, one pass over p, g, m, v (+1 byte/element decay mask).
#include "common.h"
#include "kernels.h"

constexpr int OPT_THREADS = 256;

int optim_blocks(int64_t n) {
    const int64_t need = (n / 4 + OPT_THREADS - 1) / OPT_THREADS;
    const int64_t cap = 1024;
    return (int)(need < 1 ? 1 : (need < cap ? need : cap));
}

__global__ __launch_bounds__(OPT_THREADS) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* partials) {
    __shared__ float red[OPT_THREADS / 64];
    float s = 0.f;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * OPT_THREADS) {
        const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) { const float t = g[n4 * 4 + threadIdx.x]; s += t * t; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < OPT_THREADS / 64; ++w) t += red[w];
        partials[blockIdx.x] = t;
    }
}

// SLICED: per-parameter step counts / activity (transformers.AdamW keeps state['step'] per parameter and skips a
// parameter whose grad is None -- no decay, no moment update: per-task adapters / LoRA under multitask.py:296-297's
// grads = None).  slice_of[i] = parameter index of element i; slice_bc[2k], [2k+1] = 1 - b1^t_k, sqrt(1 - b2^t_k) of
// parameter k, or <= 0 when it received no gradient this step (the element is left untouched).
template <bool SLICED>
__global__ __launch_bounds__(OPT_THREADS) void adamw_kernel(AdamwArgs a) {
    // every workgroup reduces the (<= 1024) partial sums itself: no host round trip, no third launch
    __shared__ float red[OPT_THREADS / 64];
    __shared__ float clip_s;
    float s = 0.f;
    for (int i = threadIdx.x; i < a.n_partials; i += OPT_THREADS) s += a.partials[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < OPT_THREADS / 64; ++w) t += red[w];
        const float norm = sqrtf(t) * a.grad_scale;
        float c = a.max_norm > 0.f ? a.max_norm / (norm + 1e-6f) : 1.0f;
        clip_s = a.grad_scale * (c < 1.0f ? c : 1.0f);
        if (blockIdx.x == 0 && a.norm_out) *a.norm_out = norm;
    }
    __syncthreads();
    const float gscale = clip_s;
    const float b1 = a.beta1, b2 = a.beta2, lr = a.lr, eps = a.eps;
    auto upd = [&](float& p, float g, float& m, float& v, uint8_t dec, int sl) {
        float c1 = a.bias_c1, c2 = a.bias_c2_sqrt;
        if constexpr (SLICED) {
            c1 = a.slice_bc[2 * sl]; c2 = a.slice_bc[2 * sl + 1];
            if (!(c1 > 0.f)) return;                  // no gradient this step: parameter, moments and step untouched
        }
        g *= gscale;
        const float wd = dec ? a.weight_decay : 0.f;
        if (a.decay_first) p -= lr * wd * p;
        m = b1 * m + (1.f - b1) * g;
        v = b2 * v + (1.f - b2) * g * g;
        p -= (lr * c2 / c1) * m / (sqrtf(v) + eps * (a.eps_scaled ? c2 : 1.0f));
        if (!a.decay_first) p -= lr * wd * p;
    };
    const int64_t n4 = a.n / 4;
    for (int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * OPT_THREADS) {
        f32x4 p = reinterpret_cast<f32x4*>(a.p)[i], g = reinterpret_cast<f32x4*>(a.g)[i];
        f32x4 m = reinterpret_cast<f32x4*>(a.m)[i], v = reinterpret_cast<f32x4*>(a.v)[i];
        const uint32_t dk = a.decay ? reinterpret_cast<const uint32_t*>(a.decay)[i] : 0x01010101u;
        int sl[4] = {0, 0, 0, 0};
        if constexpr (SLICED) {
            const auto t = reinterpret_cast<const __attribute__((ext_vector_type(4))) int*>(a.slice_of)[i];
            sl[0] = t[0]; sl[1] = t[1]; sl[2] = t[2]; sl[3] = t[3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pj = p[j], mj = m[j], vj = v[j];
            upd(pj, g[j], mj, vj, (uint8_t)(dk >> (8 * j)), sl[j]);
            p[j] = pj; m[j] = mj; v[j] = vj;
        }
        reinterpret_cast<f32x4*>(a.p)[i] = p;
        reinterpret_cast<f32x4*>(a.m)[i] = m;
        reinterpret_cast<f32x4*>(a.v)[i] = v;
        if (a.zero_grad) { const f32x4 z = {0.f, 0.f, 0.f, 0.f}; reinterpret_cast<f32x4*>(a.g)[i] = z; }
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(a.n - n4 * 4)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        upd(a.p[i], a.g[i], a.m[i], a.v[i], a.decay ? a.decay[i] : (uint8_t)1, SLICED ? a.slice_of[i] : 0);
        if (a.zero_grad) a.g[i] = 0.f;
    }
}

hipError_t launch_sumsq(const float* g, int64_t n, float* partials, hipStream_t stream) {
    hipLaunchKernelGGL(sumsq_kernel, dim3(optim_blocks(n)), dim3(OPT_THREADS), 0, stream, g, n, partials);
    return hipGetLastError();
}
hipError_t launch_adamw(const AdamwArgs& a, hipStream_t stream) {
    if (a.slice_of != nullptr)
        hipLaunchKernelGGL(adamw_kernel<true>, dim3(optim_blocks(a.n)), dim3(OPT_THREADS), 0, stream, a);
    else
        hipLaunchKernelGGL(adamw_kernel<false>, dim3(optim_blocks(a.n)), dim3(OPT_THREADS), 0, stream, a);
    return hipGetLastError();
}

// dst += src over n IO-dtype elements (n % 8 == 0, 16-byte aligned): the fallback of the K1 backward's dx1 accumulation for
// the kernel forms that do not add in their epilogue
template <typename IO>
__global__ __launch_bounds__(OPT_THREADS) void add_inplace_kernel(IO* dst, const IO* src, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n8; i += (int64_t)gridDim.x * OPT_THREADS) {
        float a[8], b[8];
        load8_f32(dst + 8 * i, a);
        load8_f32(src + 8 * i, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += b[j];
        store8_f32(dst + 8 * i, a);
    }
}
hipError_t launch_add_inplace(void* dst, const void* src, int64_t n, int io_fp32, hipStream_t stream) {
    const int64_t n8 = n / 8;
    int64_t blocks = (n8 + OPT_THREADS - 1) / OPT_THREADS;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    if (io_fp32) hipLaunchKernelGGL(add_inplace_kernel<float>, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, stream,
                                    reinterpret_cast<float*>(dst), reinterpret_cast<const float*>(src), n8);
    else hipLaunchKernelGGL(add_inplace_kernel<__bf16>, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, stream,
                            reinterpret_cast<__bf16*>(dst), reinterpret_cast<const __bf16*>(src), n8);
    return hipGetLastError();
}

// out = src[0] + ... + src[n - 1] over `len` IO-dtype elements (len % 8 == 0, 16-byte aligned), fp32 accumulation, one rounding: the
// gradient of a tensor that n consumers read (the encoder output feeding every decoder layer's cross-attention,
// my_transformers/modeling_bart.py:2300-2330) -- autograd sums such gradients pairwise, n - 1 passes of
// three row units each; this is one pass of n + 1 units.  Every source's piece is in flight before the first add.
template <typename IO, int N>
__global__ __launch_bounds__(OPT_THREADS) void sum_n_kernel(SumNArgs a, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n8; i += (int64_t)gridDim.x * OPT_THREADS) {
        float v[N][8];
#pragma unroll
        for (int k = 0; k < N; ++k) load8_f32(reinterpret_cast<const IO*>(a.src[k]) + 8 * i, v[k]);
#pragma unroll
        for (int k = 1; k < N; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[0][j] += v[k][j];
        store8_f32(reinterpret_cast<IO*>(a.out) + 8 * i, v[0]);
    }
}
template <typename IO>
static void sum_n_launch(const SumNArgs& a, int n, int64_t n8, unsigned blocks, hipStream_t s) {
    switch (n) {
        case 2: hipLaunchKernelGGL((sum_n_kernel<IO, 2>), dim3(blocks), dim3(OPT_THREADS), 0, s, a, n8); break;
        case 3: hipLaunchKernelGGL((sum_n_kernel<IO, 3>), dim3(blocks), dim3(OPT_THREADS), 0, s, a, n8); break;
        case 4: hipLaunchKernelGGL((sum_n_kernel<IO, 4>), dim3(blocks), dim3(OPT_THREADS), 0, s, a, n8); break;
        case 5: hipLaunchKernelGGL((sum_n_kernel<IO, 5>), dim3(blocks), dim3(OPT_THREADS), 0, s, a, n8); break;
        case 6: hipLaunchKernelGGL((sum_n_kernel<IO, 6>), dim3(blocks), dim3(OPT_THREADS), 0, s, a, n8); break;
        case 7: hipLaunchKernelGGL((sum_n_kernel<IO, 7>), dim3(blocks), dim3(OPT_THREADS), 0, s, a, n8); break;
        default: hipLaunchKernelGGL((sum_n_kernel<IO, 8>), dim3(blocks), dim3(OPT_THREADS), 0, s, a, n8); break;
    }
}
// n in 2 .. 8 per launch; the caller chains launches for more sources (out may be src[0])
hipError_t launch_sum_n(const SumNArgs& a, int n, int64_t len, int io_fp32, hipStream_t stream) {
    const int64_t n8 = len / 8;
    int64_t blocks = (n8 + OPT_THREADS - 1) / OPT_THREADS;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (io_fp32) sum_n_launch<float>(a, n, n8, (unsigned)blocks, stream);
    else sum_n_launch<__bf16>(a, n, n8, (unsigned)blocks, stream);
    return hipGetLastError();
}
