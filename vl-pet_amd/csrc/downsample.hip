// CLIP-grid down-sampling ahead of K4: AdaptiveMaxPool2d (s_in x s_in -> s_out x s_out) over the token grid of
// x [n_img, s_in*s_in, dim] (channel-contiguous) -> out [n_img, s_out*s_out, dim], cast to the IO dtype on the way
// (src/modeling_bart.py:556-581; the NLVR two-image form is the same memory layout with n_img = 2B).
// PyTorch's adaptive window: rows floor(i*s_in/s_out) .. ceil((i+1)*s_in/s_out) - 1.
// HBM-bound gather: a thread owns 8 channels of one output token; every load / store is 16 or 32 contiguous bytes
// and a wave covers 512 contiguous channels.
#include "common.h"
#include "kernels.h"

template <typename IN, typename OUT>
__global__ __launch_bounds__(256) void downsample_kernel(PoolArgs a) {
    const int cgroups = a.dim / 8;
    const int64_t total = a.n_img * a.s_out * a.s_out * cgroups;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int cg = (int)(idx % cgroups);
        const int64_t tok = idx / cgroups;
        const int oj = (int)(tok % a.s_out), oi = (int)((tok / a.s_out) % a.s_out);
        const int64_t img = tok / ((int64_t)a.s_out * a.s_out);
        const int h0 = (oi * a.s_in) / a.s_out, h1 = ((oi + 1) * a.s_in + a.s_out - 1) / a.s_out;
        const int w0 = (oj * a.s_in) / a.s_out, w1 = ((oj + 1) * a.s_in + a.s_out - 1) / a.s_out;
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
        const IN* base = reinterpret_cast<const IN*>(a.x) + img * (int64_t)a.s_in * a.s_in * a.dim + cg * 8;
        for (int hh = h0; hh < h1; ++hh) {
            for (int ww = w0; ww < w1; ++ww) {
                const IN* p = base + (int64_t)(hh * a.s_in + ww) * a.dim;
                float v[8];
                if constexpr (sizeof(IN) == 4) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
                } else {
                    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = (v[j] > m[j] || v[j] != v[j]) ? v[j] : m[j];     // NaN propagates like torch
            }
        }
        OUT* q = reinterpret_cast<OUT*>(a.out) + tok * (int64_t)a.dim + cg * 8;
        if constexpr (sizeof(OUT) == 4) {
            const f32x4 lo = {m[0], m[1], m[2], m[3]}, hi = {m[4], m[5], m[6], m[7]};
            *reinterpret_cast<f32x4*>(q) = lo; *reinterpret_cast<f32x4*>(q + 4) = hi;
        } else {
            bf16x8 t;
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = (__bf16)m[j];
            *reinterpret_cast<bf16x8*>(q) = t;
        }
    }
}

hipError_t launch_downsample(const PoolArgs& a, int in_fp32, int out_fp32, hipStream_t stream) {
    const int64_t total = a.n_img * a.s_out * a.s_out * (a.dim / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const dim3 g((unsigned)blocks), b(256);
    if (in_fp32 && out_fp32) hipLaunchKernelGGL((downsample_kernel<float, float>), g, b, 0, stream, a);
    else if (in_fp32) hipLaunchKernelGGL((downsample_kernel<float, __bf16>), g, b, 0, stream, a);
    else if (out_fp32) hipLaunchKernelGGL((downsample_kernel<__bf16, float>), g, b, 0, stream, a);
    else hipLaunchKernelGGL((downsample_kernel<__bf16, __bf16>), g, b, 0, stream, a);
    return hipGetLastError();
}
