// Shared pieces of the HBM-bound row kernels (tail.hip, rowgate.hip): one wave per activation row, a lane
// owns the 16-byte pieces lane, lane + 64, ... of the row (whole 128-byte lines per 8 lanes).
#pragma once
#include "common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// PB = bytes of a piece: 16 (8 bf16 / 4 fp32), or 8 for bf16 rows whose 16-byte pieces leave lanes idle (round 5: d = 768 is 96
// 16-byte pieces -- lanes 32..63 own one piece, lanes 0..31 two, every per-lane array is sized for two -- but 192 8-byte pieces =
// three per lane exactly: 12 elements per lane instead of 16 slots, a quarter fewer registers and no idle half wave).
// two sums at once (the shuffles of the two chains interleave)
__device__ __forceinline__ void wave_sum2(float a, float b, float& ra, float& rb) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    ra = a; rb = b;
}

template <typename IO, int PB = 16> struct Piece {
    static constexpr int E = PB / (int)sizeof(IO);     // elements per piece
    static constexpr int B = PB;
    static_assert(PB == 16 || (PB == 8 && sizeof(IO) == 2), "8-byte pieces are the bf16 form");
    using Raw = typename std::conditional<PB == 16, u32x4, u32x2>::type;
    static __device__ __forceinline__ void load(const void* p, float* v) {
        static_assert(PB == 16, "");
        if constexpr (E == 8) {          // (non-temporal: the row kernels read every row tensor once per launch)
            const bf16x8 a = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
        } else {
            const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = a[j];
        }
    }
    // the same piece kept raw (a prefetched row lives in registers unconverted)
    static __device__ __forceinline__ Raw load_raw(const void* p) { return *reinterpret_cast<const Raw*>(p); }
    // the same load with the non-temporal policy: a row tensor one wave reads once (tools/bw_probe.hip: cold streams read at
    // 4.5 TB/s with it against 2.4-2.6 TB/s with the default policy)
    static __device__ __forceinline__ Raw load_raw_nt(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const Raw*>(p)); }
    static __device__ __forceinline__ void from_raw(const Raw& r, float* v) {
        if constexpr (sizeof(IO) == 2) {
#pragma unroll
            for (int i = 0; i < PB / 4; ++i) {
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
        if constexpr (PB == 8) {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
            const bf16x4_t a = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *reinterpret_cast<bf16x4_t*>(p) = a;
        } else if constexpr (E == 8) {
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
