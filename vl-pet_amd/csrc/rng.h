// Counter-based dropout masks shared by the K5 tail (tail.hip) and the K3 LoRA kernels (pet_fwd / pet_bwd / wgrad).
//
// Philox-4x32 (7 rounds; Salmon et al., "Parallel random numbers: as easy as 1, 2, 3") keyed by the call's 64-bit
// seed, counter = index of the 8-element group of the row-major [M, d] tensor; element j of the group keeps iff its
// 16-bit lane >= thr = round(p * 65536).  The mask depends on (seed, element index) only -- not on the IO dtype, the
// kernel or the workgroup geometry -- so the forward, the backward rows kernel and the weight-gradient kernel
// regenerate the same mask and nothing is stored.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void philox7(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t* o) {
    uint32_t c2 = 0x5bd1e995u, c3 = 0x2545f491u;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        // one 64-bit product per multiplier (v_mad_u64_u32) instead of a v_mul_hi_u32 / v_mul_lo_u32 pair: the 32-bit integer
        // multiplies are the slow VALU ops of the generator (28 -> 14 per group of 8 elements)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
// Seeds under graph replay.  A captured train step replays its kernel arguments, so a per-call seed passed by value would give
// every step the same masks.  vlpet_set_seed_counter (include/vlpet_hip.h) hands the library the address of a 64-bit device
// counter the trainer bumps once per step; every dropout-carrying kernel reads it ONCE in its prologue (a scalar load) and
// mixes it into its call's seed.  No counter (nullptr): the seed is used as passed.  Forward and backward of a step see the same
// counter value, so regenerated masks still match.
__device__ __forceinline__ uint64_t vlpet_eff_seed(uint64_t seed, const uint64_t* ctr) {
    return ctr != nullptr ? seed + *ctr * 0x9E3779B97F4A7C15ull : seed;
}
// keep flags (bit j = element j of the 8-element group kept)
__device__ __forceinline__ uint32_t keep8(int64_t group, uint64_t seed, uint32_t thr) {
    uint32_t o[4];
    philox7((uint32_t)group, (uint32_t)((uint64_t)group >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
    uint32_t bits = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t u = (o[j >> 1] >> (16 * (j & 1))) & 0xffffu;
        bits |= (u >= thr ? 1u : 0u) << j;
    }
    return bits;
}

// Where a kernel's dropout mask comes from: an explicit 0/1 byte mask (keep != nullptr; "bring your own mask"), the packed
// mask an earlier kernel of the same call chain left behind (bits != nullptr: 1 bit per element, see drop_pos), or the
// generator above (thr != 0).  None of them: no dropout (the kernels' DROP template flag is then false).
struct DropSpec {
    const uint8_t* keep;     // [M, d] uint8, 1 = keep, or nullptr
    uint8_t* keep_out;       // forward only: optional [M, d] 0/1 export of the mask that was applied (parity tests)
    const uint8_t* bits;     // [M, d/8] packed mask written by the forward (training form), or nullptr
    uint8_t* bits_out;       // forward only: where to leave the packed mask for the backward, or nullptr
    uint64_t seed;
    uint32_t thr;            // drop iff 16-bit uniform < thr
    float keep_scale;        // 1 / (1 - p)
    const uint64_t* seed_ctr;   // optional device step counter mixed into `seed` (vlpet_eff_seed); kernels resolve it once: drop_resolved
};
// the spec with the step counter folded into the seed (one scalar load: call it in the kernel prologue, never inside a loop)
__device__ __forceinline__ DropSpec drop_resolved(const DropSpec& s) {
    DropSpec r = s;
    if (s.thr != 0 && s.keep == nullptr && s.bits == nullptr) r.seed = vlpet_eff_seed(s.seed, s.seed_ctr);
    r.seed_ctr = nullptr;
    return r;
}
static inline bool drop_active(const DropSpec& s) { return s.keep != nullptr || s.bits != nullptr || s.thr != 0; }

// Byte position, inside a row's d/8 mask bytes, of 8-element group G (elements 8G .. 8G+7).  Not the natural order: inside
// every block of 8 groups the even groups come first -- lane (row, h) of the forward kernels owns groups 2u + h of a
// 64-feature stage, so with this order its four mask bytes are one aligned dword (one store per stage instead of four).
__host__ __device__ inline int drop_pos(int G) { return (G & ~7) + ((G & 1) << 2) + ((G >> 1) & 3); }

// keep flags of elements f0 .. f0+7 (f0 % 8 == 0) of row `row` of the row-major [M, d] tensor
__device__ __forceinline__ uint32_t drop_bits8(const DropSpec& s, int64_t row, int f0, int d) {
    if (s.keep != nullptr) {
        const uint64_t kp = *reinterpret_cast<const uint64_t*>(s.keep + row * d + f0);
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) bits |= (((kp >> (8 * j)) & 0xff) ? 1u : 0u) << j;
        return bits;
    }
    if (s.bits != nullptr) return s.bits[row * (int64_t)(d >> 3) + drop_pos(f0 >> 3)];
    return keep8((row * d + f0) >> 3, s.seed, s.thr);
}
__device__ __forceinline__ void drop_export8(uint8_t* keep_out, int64_t e0, uint32_t bits) {
    uint64_t v = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) v |= (uint64_t)((bits >> j) & 1u) << (8 * j);
    *reinterpret_cast<uint64_t*>(keep_out + e0) = v;
}
