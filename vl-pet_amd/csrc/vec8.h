// Eight consecutive elements of a row-major IO tensor as one per-lane unit (16 B of bf16, 32 B of fp32): the streaming
// kernels outside the MFMA path (actdrop.hip, celoss.hip) walk their tensors in these groups.
#pragma once
#include "common.h"

template <typename IO> struct Vec8;
template <> struct Vec8<__bf16> {
    bf16x8 v;
    // (non-temporal, round 5: every tensor these kernels walk is read once per launch; cold streams read at 4.5 TB/s with the nt policy
    //  against 2.4-2.6 with the default one -- tools/bw_probe.hip, and 50 -> 36 us on the K5 forward at 28,000 cold rows)
    __device__ __forceinline__ void load(const void* p, int64_t g) { v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p) + g); }
    __device__ __forceinline__ void store(void* p, int64_t g) const { reinterpret_cast<bf16x8*>(p)[g] = v; }
    __device__ __forceinline__ float get(int j) const { return (float)v[j]; }
    __device__ __forceinline__ void set(int j, float f) { v[j] = (__bf16)f; }
};
template <> struct Vec8<float> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const void* p, int64_t g) { a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + 2 * g); b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + 2 * g + 1); }
    __device__ __forceinline__ void store(void* p, int64_t g) const { reinterpret_cast<f32x4*>(p)[2 * g] = a; reinterpret_cast<f32x4*>(p)[2 * g + 1] = b; }
    __device__ __forceinline__ float get(int j) const { return j < 4 ? a[j] : b[j - 4]; }
    __device__ __forceinline__ void set(int j, float f) { if (j < 4) a[j] = f; else b[j - 4] = f; }
};

