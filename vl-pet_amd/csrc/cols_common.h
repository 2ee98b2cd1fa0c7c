// Helpers shared by the column-parallel K1 backward kernels (pet_cols.hip, pet_cols6.hip): compile-time loops, inline-asm LDS
// accesses with hand-placed waits (see trread.h for why), counted vmcnt waits, bf16 packing, and the LDS swizzles.
#pragma once
#include <type_traits>
#include <utility>
#include "pet16.h"
#include "kernels.h"
#include "trread.h"

template <typename F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

template <int OFF> __device__ __forceinline__ void lds_read16(u32x4& o, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(o) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_write16(uint32_t addr, const u32x4& v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(OFF) : "memory");
}
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
template <int OFF> __device__ __forceinline__ void lds_read8(u32x2& o, uint32_t addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(o) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_write8(uint32_t addr, const u32x2& v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lgkm_fence(u32x4& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) :: "memory"); }
__device__ __forceinline__ void lgkm_tie(u32x4& a) { asm volatile("" : "+v"(a) :: "memory"); }
__device__ __forceinline__ bf16x8 as_bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// transpose read with separate addresses for the two 4-row halves (their rows carry different swizzles)
template <int OFF> __device__ __forceinline__ void tr_read2(TrOp& o, uint32_t alo, uint32_t ahi) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(o.lo) : "v"(alo), "n"(OFF) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(o.hi) : "v"(ahi), "n"(OFF) : "memory");
}
// LDS images (all swizzles are applied on the SOURCE side of the global_load_lds: slot = 16-byte unit of a row):
//   row tiles [32 rows x 128 B]: slot ^= fsw(row), fsw = (q & 1) << 2 | q >> 1 with q = (row >> 1) & 7 -- bit 2 alternates every two
//     rows (the four rows of a transpose read fall into four 64-byte bank windows), the other two bits make the 16 rows of a
//     ds_read_b128 lane group distinct (the row-per-lane reads of dy, x2, dh);
//   bottleneck tiles [32 rows x 64*RT B]: slot ^= (row >> 2) & 3 -- the four rows of a transpose read share it (their native
//     conflict-free pattern is kept), the rows 4 apart of a ds_read_b128 lane group (B fragments, lane = row) do not.
__device__ __forceinline__ int fsw(int row) { const int q = (row >> 1) & 7; return ((q & 1) << 2) | (q >> 1); }
__device__ __forceinline__ int gsw(int row) { return (row >> 2) & 3; }

// at most n vector-memory operations of this wave still in flight (n is wave-uniform)
__device__ __forceinline__ void vm_wait(int n) {
#define VLPET_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        VLPET_VMW(0) VLPET_VMW(1) VLPET_VMW(2) VLPET_VMW(3) VLPET_VMW(4) VLPET_VMW(5) VLPET_VMW(6) VLPET_VMW(7)
        VLPET_VMW(8) VLPET_VMW(9) VLPET_VMW(10) VLPET_VMW(11) VLPET_VMW(12) VLPET_VMW(13) VLPET_VMW(14) VLPET_VMW(15)
        VLPET_VMW(16) VLPET_VMW(17) VLPET_VMW(18) VLPET_VMW(19) VLPET_VMW(20) VLPET_VMW(21) VLPET_VMW(22) VLPET_VMW(23)
        VLPET_VMW(24) VLPET_VMW(25) VLPET_VMW(26) VLPET_VMW(27) VLPET_VMW(28) VLPET_VMW(29) VLPET_VMW(30)
        default: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;       // more than 30: stricter is safe
    }
#undef VLPET_VMW
}

__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bf_at(const u32x4& v, int j) { return (j & 1) ? bf_hi(v[j >> 1]) : bf_lo(v[j >> 1]); }
__device__ __forceinline__ u32x4 pack8(const float* v) {
    bf16x8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (__bf16)v[j];
    return __builtin_bit_cast(u32x4, t);
}
// sigmoid with one v_exp_f32 and one v_rcp_f32
__device__ __forceinline__ float sigm(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}


// Workgroup -> (group, member) map of the column-parallel passes.  A group = the U workgroups that re-read the same bottleneck
// rows (the column blocks of a row chunk); block b runs on XCD b % 8, and a group's members should share an XCD's L2.  Each XCD
// has 32 CUs = G = 32 / U whole groups (slots j = b / 8 < U * G); the 32 - U * G CUs per XCD that are left over take the
// members of a few more groups, spread over the XCDs (those groups re-read their bottleneck rows through several L2s -- a few
// MB -- but at d = 768 they turn 240 busy CUs into 252: two more row chunks, one step less per workgroup).
__host__ __device__ inline int cols_groups_max(int U) { const int G = 32 / U; return 8 * G + ((32 - U * G) * 8) / U; }
__host__ __device__ inline void cols_decode(int b, int U, int& group, int& member) {
    const int G = 32 / U, main = U * G, x = b & 7, j = b >> 3;
    if (j < main) { member = j % U; group = (j / U) * 8 + x; }
    else { const int l = (j - main) * 8 + x; group = 8 * G + l / U; member = l % U; }
}
__host__ __device__ inline unsigned cols_grid(int U, int ngroups) {
    const int G = 32 / U;
    if (ngroups <= 8 * G) return 8u * (unsigned)U * (unsigned)((ngroups + 7) / 8);
    return 8u * (unsigned)(U * G + ((ngroups - 8 * G) * U + 7) / 8);
}
