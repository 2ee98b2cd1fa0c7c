// Shared pieces of the HBM-bound row kernels (tail.hip, rowgate.hip): one wave per activation row, a lane
// owns the 16-byte pieces lane, lane + 64, ... of the row (whole 128-byte lines per 8 lanes).
#pragma once
#include "common.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename IO> struct Piece {
    static constexpr int E = 16 / (int)sizeof(IO);     // elements per 16-byte piece (8 bf16 / 4 fp32)
    static __device__ __forceinline__ void load(const void* p, float* v) {
        if constexpr (E == 8) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
        } else {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = a[j];
        }
    }
    // the same piece kept raw (a prefetched row lives in registers unconverted)
    static __device__ __forceinline__ u32x4 load_raw(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
    static __device__ __forceinline__ void from_raw(const u32x4& r, float* v) {
        if constexpr (E == 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned int u = r[i];
                v[2 * i] = __builtin_bit_cast(float, u << 16);
                v[2 * i + 1] = __builtin_bit_cast(float, u & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const unsigned int u = r[i]; v[i] = __builtin_bit_cast(float, u); }
        }
    }
    static __device__ __forceinline__ void store(void* p, const float* v) {
        if constexpr (E == 8) {
            bf16x8 a;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = (__bf16)v[j];
            *reinterpret_cast<bf16x8*>(p) = a;
        } else {
            const f32x4 a = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(p) = a;
        }
    }
};
