// Shared device helpers for the VL-PET hot-path kernels (gfx950 / CDNA4 only).
//
// Conventions (see tests/packing_spec.py for the layout specification):
//   * one wavefront (64 lanes) owns 32 activation rows; lane (m = lane & 31, h = lane >> 5);
//   * every contraction is v_mfma_f32_32x32x16_bf16 in swapped form (weights = A operand,
//     activation rows = B operand), so results land as "16 values of row m per lane";
//   * IO type __bf16  -> one bf16 plane per operand (NS = 1);
//     IO type float   -> operands split into bf16 hi + lo planes (NS = 2) and three MFMAs
//                        (hi*hi + hi*lo + lo*hi, fp32 accumulate): ~2^-16 relative product error,
//                        i.e. tighter than the TF32 matmuls the reference ran with on A100.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define VLPET_THREADS 256
#define VLPET_WAVES 4
#define VLPET_ROWS_PER_WG 128

template <typename IO> struct IoTraits;
template <> struct IoTraits<__bf16> { static constexpr int NS = 1; };
template <> struct IoTraits<float> { static constexpr int NS = 2; };

template <int NS> struct Frag { bf16x8 p[NS]; };

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int NS>
__device__ __forceinline__ f32x16 mfma_ns(const Frag<NS>& a, const Frag<NS>& b, f32x16 c) {
    if constexpr (NS == 2) {
        c = mfma32(a.p[1], b.p[0], c);
        c = mfma32(a.p[0], b.p[1], c);
    }
    return mfma32(a.p[0], b.p[0], c);
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// ---------------------------------------------------------------- activations
// HF NewGELUActivation (tanh form): 0.5x(1+tanh(k(x+0.044715x^3))) = x * sigmoid(2k(...)).
#define VLPET_GELU_K 0.7978845608028654f
__device__ __forceinline__ float gelu_new_f(float x) {
    float u = VLPET_GELU_K * (x + 0.044715f * x * x * x);
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * u));
}
__device__ __forceinline__ float gelu_new_grad_f(float x) {
    float x2 = x * x;
    float u = VLPET_GELU_K * (x + 0.044715f * x * x2);
    float s = __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * u));
    float du = VLPET_GELU_K * (1.0f + 3.0f * 0.044715f * x2);
    return s + x * s * (1.0f - s) * 2.0f * du;
}
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ---------------------------------------------------------------- conversions
__device__ __forceinline__ void split_bf16(float v, __bf16& hi, __bf16& lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

// 8 fp32 values -> operand fragment (1 or 2 planes)
template <int NS>
__device__ __forceinline__ Frag<NS> frag_from_f32(const float* v) {
    Frag<NS> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if constexpr (NS == 1) {
            f.p[0][j] = (__bf16)v[j];
        } else {
            __bf16 hi, lo;
            split_bf16(v[j], hi, lo);
            f.p[0][j] = hi;
            f.p[1][j] = lo;
        }
    }
    return f;
}

// load 8 contiguous IO elements (16-byte aligned for bf16, 32-byte for fp32) as a fragment
__device__ __forceinline__ Frag<1> load_frag8(const __bf16* p) {
    Frag<1> f;
    f.p[0] = *reinterpret_cast<const bf16x8*>(p);
    return f;
}
__device__ __forceinline__ Frag<2> load_frag8(const float* p) {
    f32x4 a = reinterpret_cast<const f32x4*>(p)[0];
    f32x4 b = reinterpret_cast<const f32x4*>(p)[1];
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return frag_from_f32<2>(v);
}

// load / store 8 contiguous IO elements as fp32
__device__ __forceinline__ void load8_f32(const __bf16* p, float* v) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
}
__device__ __forceinline__ void load8_f32(const float* p, float* v) {
    f32x4 a = reinterpret_cast<const f32x4*>(p)[0];
    f32x4 b = reinterpret_cast<const f32x4*>(p)[1];
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void store8_f32(__bf16* p, const float* v) {
    bf16x8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x8*>(p) = t;
}
__device__ __forceinline__ void store8_f32(float* p, const float* v) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    reinterpret_cast<f32x4*>(p)[0] = a;
    reinterpret_cast<f32x4*>(p)[1] = b;
}

// ---------------------------------------------------------------- packed-pair layout
// One packed (down [r,d], up [d,r]) pair = four fragment packs of NF = d/16*RT fragments each
// (RT = padded bottleneck / 32), followed by the fp32 biases:
//   [down | up | up_t | down_t] each NF * NS KiB, then bias_down[32*RT], bias_up[d]  (fp32)
struct PackGeom {
    int64_t pack_bytes;   // bytes of one of the four packs
    int64_t bias_off;     // byte offset of bias_down
    int64_t total_bytes;
};
__host__ __device__ inline PackGeom pack_geom(int RT, int d, int NS) {
    PackGeom g;
    g.pack_bytes = (int64_t)(d / 16) * RT * NS * 1024;
    g.bias_off = 4 * g.pack_bytes;
    g.total_bytes = g.bias_off + (int64_t)(32 * RT + d) * 4;
    g.total_bytes = (g.total_bytes + 255) / 256 * 256;
    return g;
}

// Register staging of the weight-fragment stream: global -> VGPR now, VGPR -> LDS after the
// current stage's MFMAs (loads overlap compute; one __syncthreads per stage).
template <int MAXU> struct StageRegs { u32x4 v[MAXU]; };

template <int MAXU>
__device__ __forceinline__ void stage_load(StageRegs<MAXU>& r, const uint8_t* s0, int u0,
                                           const uint8_t* s1, int u1, int tid) {
    // branch-free: out-of-range units re-read unit 0 of segment 0 (never stored)
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        const int q = tid + VLPET_THREADS * i;
        const u32x4* src = reinterpret_cast<const u32x4*>(q < u0 ? s0 : s1);
        int idx = q < u0 ? q : q - u0;
        if (q >= u0 + u1) { src = reinterpret_cast<const u32x4*>(s0); idx = 0; }
        r.v[i] = src[idx];
    }
}
template <int MAXU>
__device__ __forceinline__ void stage_store(const StageRegs<MAXU>& r, uint8_t* lds, int total, int tid) {
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        const int q = tid + VLPET_THREADS * i;
        if (q < total) reinterpret_cast<u32x4*>(lds)[q] = r.v[i];
    }
}

template <int NS>
__device__ __forceinline__ Frag<NS> lds_frag(const uint8_t* buf, int frag, int lane) {
    Frag<NS> f;
#pragma unroll
    for (int p = 0; p < NS; ++p)
        f.p[p] = *reinterpret_cast<const bf16x8*>(buf + ((size_t)(frag * NS + p) * 64 + lane) * 16);
    return f;
}
