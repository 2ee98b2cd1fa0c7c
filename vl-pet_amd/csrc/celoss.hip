// Token-level cross entropy over the LM-head logits, the last op of the training forward and the first of the backward:
//     loss[n] = logsumexp(logits[n, :V]) - logits[n, label[n]]       (0 where label[n] < 0: ignore_index = -100)
//     dlogits[n, v] = dloss[n] * (softmax(logits[n])[v] - [v == label[n]])
// (src/modeling_bart.py:1574-1586 and src/modeling_t5.py:680-694: CrossEntropyLoss(ignore_index=-100, reduction='none') on
// lm_logits.view(-1, V)).  The reference's chain on [N, V] (N = B * T_dec, V = 50,465: 126 M logits per VQA step) is a
// cast to fp32, log_softmax, nll, and in the backward the softmax gradient and a cast back -- five passes over the
// largest tensor of the step.  Here: one read forward (online max / sum-of-exponentials per row, the row's log-sum-exp
// kept), one read + one write backward, arithmetic in fp32 on the IO-dtype logits (the same rounding points as the
// reference's .float() path).  A workgroup owns a row; rows are `ld` elements apart (ld % 8 == 0: the host pads the
// LM-head weight so that every row starts on a 16-byte boundary), columns [V, ld) are padding and get zero gradient.
#include "common.h"
#include "kernels.h"
#include "vec8.h"

#define CE_THREADS 256

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <typename IO>
__global__ __launch_bounds__(CE_THREADS) void ce_fwd_kernel(CeArgs a) {
    __shared__ float sm[CE_THREADS / 64], ss[CE_THREADS / 64];
    const int64_t n = blockIdx.x;
    const int64_t lab = a.labels[n];
    if (lab < 0 || lab >= a.V) {                     // ignored token (or an id outside the vocabulary: treated as ignored)
        if (threadIdx.x == 0) {
            a.loss[n] = 0.f; a.lse[n] = 0.f;
            if (a.bad != nullptr && lab != -100) atomicAdd(a.bad, 1u);      // not ignore_index: a vocabulary / tokenizer mismatch
        }
        return;
    }
    const IO* row = reinterpret_cast<const IO*>(a.logits) + n * (int64_t)a.ld;
    const int groups = (a.V + 7) >> 3;
    float m = -INFINITY, s = 0.f;
    for (int g = threadIdx.x; g < groups; g += CE_THREADS) {
        Vec8<IO> x;
        x.load(row, g);
        float v[8], cm = -INFINITY;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = (8 * g + j < a.V) ? x.get(j) : -INFINITY;
            cm = fmaxf(cm, v[j]);
        }
        const float mn = fmaxf(m, cm);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += __expf(v[j] - mn);
        s = s * __expf(m - mn) + acc;                // (m = -inf on the first group: exp(-inf) = 0, s = 0)
        m = mn;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float wm = wave_max(m);
    const float wsum = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - wm));
    if (lane == 0) { sm[wave] = wm; ss[wave] = wsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = sm[0];
#pragma unroll
        for (int w = 1; w < CE_THREADS / 64; ++w) M = fmaxf(M, sm[w]);
        float S = 0.f;
#pragma unroll
        for (int w = 0; w < CE_THREADS / 64; ++w) S += ss[w] * __expf(sm[w] - M);
        const float lse = M + __logf(S);
        a.lse[n] = lse;
        a.loss[n] = lse - (float)row[lab];
    }
}

template <typename IO>
__global__ __launch_bounds__(CE_THREADS) void ce_bwd_kernel(CeArgs a) {
    const int64_t n = blockIdx.x;
    const int64_t lab = a.labels[n];
    const bool live = lab >= 0 && lab < a.V;
    const float g = live ? a.dloss[n] : 0.f;
    const IO* row = reinterpret_cast<const IO*>(a.logits) + n * (int64_t)a.ld;
    IO* out = reinterpret_cast<IO*>(a.dlogits) + n * (int64_t)a.ld;
    const int groups = a.ld >> 3;
    if (g == 0.f) {                                  // ignored token or zero weight: a row of zeros, nothing read
        Vec8<IO> z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z.set(j, 0.f);
        for (int gi = threadIdx.x; gi < groups; gi += CE_THREADS) z.store(out, gi);
        return;
    }
    const float lse = a.lse[n];
    for (int gi = threadIdx.x; gi < groups; gi += CE_THREADS) {
        Vec8<IO> x, o;
        x.load(row, gi);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int v = 8 * gi + j;
            float p = v < a.V ? __expf(x.get(j) - lse) : 0.f;
            if (v == (int)lab) p -= 1.0f;
            o.set(j, g * p);
        }
        o.store(out, gi);
    }
}

hipError_t launch_ce(const CeArgs& a, bool bwd, int io_fp32, hipStream_t stream) {
    if (a.N <= 0) return hipSuccess;
    const dim3 grid((unsigned)a.N), blk(CE_THREADS);
    if (bwd) {
        if (io_fp32) hipLaunchKernelGGL(ce_bwd_kernel<float>, grid, blk, 0, stream, a);
        else hipLaunchKernelGGL(ce_bwd_kernel<__bf16>, grid, blk, 0, stream, a);
    } else {
        if (io_fp32) hipLaunchKernelGGL(ce_fwd_kernel<float>, grid, blk, 0, stream, a);
        else hipLaunchKernelGGL(ce_fwd_kernel<__bf16>, grid, blk, 0, stream, a);
    }
    return hipGetLastError();
}
