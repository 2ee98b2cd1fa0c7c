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

// The same through LDS: a workgroup owns (image, 256-channel slab), reads the slab of every input token ONCE (whole 1 KiB rows), and
// forms the outputs from the LDS copy.  The gather kernel above asks the memory system for every window separately: at 7 -> 6 each
// input token is requested by up to four windows (2.9x the bytes through L2 / TA: 98 us for 274 MB at B = 500).  fp32 in, bf16 / fp32 out.
template <typename OUT>
__global__ __launch_bounds__(256) void downsample_slab_kernel(PoolArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tile[];          // [s_in * s_in][256]
    const int slabs = a.dim / 256;
    const int64_t img = blockIdx.x / slabs;
    const int slab = (int)(blockIdx.x % slabs);
    const int ntok = a.s_in * a.s_in, nout = a.s_out * a.s_out;
    const int tid = threadIdx.x;
    const float* src = reinterpret_cast<const float*>(a.x) + img * (int64_t)ntok * a.dim + slab * 256;
    for (int i = tid; i < ntok * 64; i += 256) {                          // 64 float4 per token row
        const int t = i >> 6, c4 = i & 63;
        reinterpret_cast<f32x4*>(tile)[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (int64_t)t * a.dim + 4 * c4));      // (every input token is read once)
    }
    __syncthreads();
    const int cg = tid & 31;                                              // 8 channels
    for (int o = tid >> 5; o < nout; o += 8) {
        const int oi = o / a.s_out, oj = o % a.s_out;
        const int h0 = (oi * a.s_in) / a.s_out, h1 = ((oi + 1) * a.s_in + a.s_out - 1) / a.s_out;
        const int w0 = (oj * a.s_in) / a.s_out, w1 = ((oj + 1) * a.s_in + a.s_out - 1) / a.s_out;
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
        for (int hh = h0; hh < h1; ++hh)
            for (int ww = w0; ww < w1; ++ww) {
                const f32x4* p = reinterpret_cast<const f32x4*>(tile + (hh * a.s_in + ww) * 256 + cg * 8);
                const f32x4 lo = p[0], hi = p[1];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    m[j] = (lo[j] > m[j] || lo[j] != lo[j]) ? lo[j] : m[j];          // NaN propagates like torch
                    m[4 + j] = (hi[j] > m[4 + j] || hi[j] != hi[j]) ? hi[j] : m[4 + j];
                }
            }
        OUT* q = reinterpret_cast<OUT*>(a.out) + (img * nout + o) * (int64_t)a.dim + slab * 256 + cg * 8;
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
    const size_t slab_lds = (size_t)a.s_in * a.s_in * 256 * 4;
    if (in_fp32 && a.dim % 256 == 0 && slab_lds <= 64 * 1024 && a.n_img * (a.dim / 256) <= 0x7fffffff) {
        const dim3 g((unsigned)(a.n_img * (a.dim / 256))), b(256);
        if (out_fp32) hipLaunchKernelGGL((downsample_slab_kernel<float>), g, b, slab_lds, stream, a);
        else hipLaunchKernelGGL((downsample_slab_kernel<__bf16>), g, b, slab_lds, stream, a);
        return hipGetLastError();
    }
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
