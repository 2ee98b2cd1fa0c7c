// Row kernels for the other three granularity gates of the encoder adapter:
//
//   small   (use_encoder_adapter_gating_small_xy_cat):  g = mean_S sigmoid(w . [x1 ; h] + b)   -- one scalar per sample
//   middleX (use_encoder_adapter_gating_middle_xy_add): g = sigmoid(w . (x1 + h) + b)          -- one scalar per token
//   middleY (use_encoder_adapter_gating_middle_ia3_add): y = h + h*z  /  h + 1 + z             -- one vector z in R^d
//
// (my_transformers/modeling_bart.py:1210-1231, 1326-1347; T5: my_transformers/modeling_t5.py:391-403, 807-819), where
// h = s2*x2 + sd*up(gelu_new(cat_i down_i(x2))) comes from the fused K1 kernel in adapter-only mode.  The [M, d]
// passes are these HBM-bound kernels (one wave per row, whole-line accesses, wave reductions); the O(M) scalar
// algebra between them (sigmoid of a row scalar, the sequence mean) stays with the host library.
//
//   ROW_DOT    : s[row] = sum_j a[row,j]*va[j] + c[row,j]*vc[j]      (va == null: s[row] = sum_j a[row,j]*c[row,j])
//   ROW_AFFINE : o1[row,:] = a[row,:]*ra[row] + rb[row]
//   ROW_BWD    : o1 = ra[row]*a + rb[row]*vc  (dh);  o2 = rb[row]*va  (dx1);  partials: sum_rows rb*c (dwa), rb*e (dwc)
//                (a = dy, c = x1, e = h)
//   VEC_FWD    : o1[row,:] = a[row,:]*va[:] + vc[:]
//   VEC_BWD    : o1 = a*va (dh, a = dy);  partials: sum_rows a*c (c = h), sum_rows a
#include "common.h"
#include "kernels.h"
#include "rowops.h"

constexpr int RG_WAVES = 4;

template <int NP, int E>
__device__ __forceinline__ void emit_partials(float (&p0)[NP][E], float (&p1)[NP][E], float* dst, int d, int pieces,
                                              int lane, int wave) {
    __shared__ float acc[RG_WAVES][2][64 * 8];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int p = lane + 64 * k;
#pragma unroll
        for (int j = 0; j < E; ++j) { acc[wave][0][lane * 8 + j] = p0[k][j]; acc[wave][1][lane * 8 + j] = p1[k][j]; }
        __syncthreads();
        if (wave == 0 && p < pieces) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int w = 0; w < RG_WAVES; ++w) { s0 += acc[w][0][lane * 8 + j]; s1 += acc[w][1][lane * 8 + j]; }
                dst[p * E + j] = s0; dst[d + p * E + j] = s1;
            }
        }
        __syncthreads();
    }
}

template <typename IO, int NP, int OP>
__global__ __launch_bounds__(RG_WAVES * 64) void rowgate_kernel(RowArgs a) {
    using P = Piece<IO>;
    constexpr int E = P::E;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int d = a.d, pieces = d / E;
    const uint8_t* A = reinterpret_cast<const uint8_t*>(a.a);
    const uint8_t* C = reinterpret_cast<const uint8_t*>(a.c);
    const uint8_t* Ein = reinterpret_cast<const uint8_t*>(a.e);
    uint8_t* O1 = reinterpret_cast<uint8_t*>(a.o1);
    uint8_t* O2 = reinterpret_cast<uint8_t*>(a.o2);
    // per-feature vectors of this lane's pieces
    float va[NP][E], vc[NP][E];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int p = lane + 64 * k;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            va[k][j] = (a.va && p < pieces) ? a.va[p * E + j] : 0.f;
            vc[k][j] = (a.vc && p < pieces) ? a.vc[p * E + j] : 0.f;
        }
    }
    float p0[NP][E], p1[NP][E];
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int j = 0; j < E; ++j) { p0[k][j] = 0.f; p1[k][j] = 0.f; }

    for (int64_t row = (int64_t)blockIdx.x * RG_WAVES + wave; row < a.M; row += (int64_t)gridDim.x * RG_WAVES) {
        const int64_t rb = row * d * (int64_t)sizeof(IO);
        if constexpr (OP == ROW_DOT) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = lane + 64 * k;
                if (p < pieces) {
                    float x[E], y[E];
                    P::load(A + rb + p * 16, x);
                    if (C) P::load(C + rb + p * 16, y);
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        if (a.va) s += x[j] * va[k][j] + (C ? y[j] * vc[k][j] : 0.f);
                        else s += x[j] * y[j];
                    }
                }
            }
            s = wave_sum(s);
            if (lane == 0) a.rs[row] = s;
        } else if constexpr (OP == ROW_AFFINE) {
            const float al = a.ra[row], ga = a.rb ? a.rb[row] : 0.f;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = lane + 64 * k;
                if (p < pieces) {
                    float x[E];
                    P::load(A + rb + p * 16, x);
#pragma unroll
                    for (int j = 0; j < E; ++j) x[j] = x[j] * al + ga;
                    P::store(O1 + rb + p * 16, x);
                }
            }
        } else if constexpr (OP == ROW_BWD) {
            const float al = a.ra[row], be = a.rb[row];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = lane + 64 * k;
                if (p < pieces) {
                    float dy[E], x1[E], h[E], o[E];
                    P::load(A + rb + p * 16, dy);
                    P::load(C + rb + p * 16, x1);
                    P::load(Ein + rb + p * 16, h);
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        o[j] = al * dy[j] + be * vc[k][j];
                        p0[k][j] += be * x1[j];
                        p1[k][j] += be * h[j];
                    }
                    P::store(O1 + rb + p * 16, o);
#pragma unroll
                    for (int j = 0; j < E; ++j) o[j] = be * va[k][j];
                    P::store(O2 + rb + p * 16, o);
                }
            }
        } else if constexpr (OP == VEC_FWD) {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = lane + 64 * k;
                if (p < pieces) {
                    float x[E];
                    P::load(A + rb + p * 16, x);
#pragma unroll
                    for (int j = 0; j < E; ++j) x[j] = x[j] * va[k][j] + vc[k][j];
                    P::store(O1 + rb + p * 16, x);
                }
            }
        } else {    // VEC_BWD
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = lane + 64 * k;
                if (p < pieces) {
                    float dy[E], h[E], o[E];
                    P::load(A + rb + p * 16, dy);
                    P::load(C + rb + p * 16, h);
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        o[j] = dy[j] * va[k][j];
                        p0[k][j] += dy[j] * h[j];
                        p1[k][j] += dy[j];
                    }
                    P::store(O1 + rb + p * 16, o);
                }
            }
        }
    }
    if constexpr (OP == ROW_BWD || OP == VEC_BWD) {
        emit_partials<NP, E>(p0, p1, a.part + (size_t)blockIdx.x * 2 * d, d, pieces, lane, wave);
    }
}

int rowgate_blocks(int64_t M) {
    const int64_t need = (M + RG_WAVES - 1) / RG_WAVES;
    const int64_t cap = 256 * 8;
    return (int)(need < cap ? need : cap);
}

template <typename IO, int NP>
static hipError_t launch_np(const RowArgs& a, int op, hipStream_t stream) {
    const dim3 g(rowgate_blocks(a.M)), b(RG_WAVES * 64);
    switch (op) {
        case ROW_DOT: hipLaunchKernelGGL((rowgate_kernel<IO, NP, ROW_DOT>), g, b, 0, stream, a); break;
        case ROW_AFFINE: hipLaunchKernelGGL((rowgate_kernel<IO, NP, ROW_AFFINE>), g, b, 0, stream, a); break;
        case ROW_BWD: hipLaunchKernelGGL((rowgate_kernel<IO, NP, ROW_BWD>), g, b, 0, stream, a); break;
        case VEC_FWD: hipLaunchKernelGGL((rowgate_kernel<IO, NP, VEC_FWD>), g, b, 0, stream, a); break;
        case VEC_BWD: hipLaunchKernelGGL((rowgate_kernel<IO, NP, VEC_BWD>), g, b, 0, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <typename IO>
static hipError_t launch_io(const RowArgs& a, int op, hipStream_t stream) {
    const int pieces = a.d / Piece<IO>::E;
    const int np = (pieces + 63) / 64;
    if (np <= 1) return launch_np<IO, 1>(a, op, stream);
    if (np <= 2) return launch_np<IO, 2>(a, op, stream);
    if (np <= 3) return launch_np<IO, 3>(a, op, stream);
    if (np <= 4) return launch_np<IO, 4>(a, op, stream);
    return hipErrorInvalidValue;
}

hipError_t launch_rowgate(const RowArgs& a, int op, int io_fp32, hipStream_t stream) {
    return io_fp32 ? launch_io<float>(a, op, stream) : launch_io<__bf16>(a, op, stream);
}
