// v3 building blocks: 16-row waves on v_mfma_f32_16x16x32_bf16, every global read through LDS by
// global_load_lds (no staging VGPRs, fully coalesced 128-byte lines), outputs staged through LDS
// and stored as whole lines.  Layout specification: tests/packing_spec.py (section v3) and
// tests/test_layout_model16.py.
//
//   lane (m = lane & 15, g = lane >> 4) owns activation row m of its wave's 16 rows;
//   a *stage* moves 128 bytes of every row: FE = 64 features (bf16 IO) or 32 (fp32 IO).
#pragma once
#include "common.h"

typedef __attribute__((address_space(1))) const void gmem_cv;
typedef __attribute__((address_space(3))) void lmem_v;

template <typename IO> struct Geo {
    static constexpr int NS = IoTraits<IO>::NS;
    static constexpr int FE = 64 / NS;      // features per stage
    static constexpr int KS = FE / 32;      // MFMA k-steps per down-phase stage
    static constexpr int NQ = FE / 16;      // 16-feature n-tiles per up-phase stage
    static constexpr int LW = FE / 4;       // contiguous output features per lane
    static constexpr int E2 = LW / 8;       // k-steps of the feature contraction (backward) per stage
    static constexpr int EPP = 16 / (int)sizeof(IO);   // elements per 16-byte piece
};

__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <int NS>
__device__ __forceinline__ f32x4 mfma16_ns(const Frag<NS>& a, const Frag<NS>& b, f32x4 c) {
    if constexpr (NS == 2) {
        c = mfma16(a.p[1], b.p[0], c);
        c = mfma16(a.p[0], b.p[1], c);
    }
    return mfma16(a.p[0], b.p[0], c);
}

// async global -> LDS, 16 bytes per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gmem_cv*)gsrc, (lmem_v*)lds_wave_base, 16, 0, 0);
}
// the same for data that ONE workgroup reads ONCE (activation rows): cache-policy bits of the build (aux 2 = nt; MI355X_MICROARCH.md
// "nt-weights": issue -> landed -18 % for a stream a single CU reads once, -6 % end to end when every CU re-reads it -- so never on
// the packed weights).  Measured here (profiles/r02_k1bench_nt_row_loads.txt, r02_bench_instep_nt_row_loads_ab.txt): with cold inputs
// K1 forward 70.6 -> 57.9 us, backward rows 97.0 -> 85.5, weight gradients 84.6 -> 60.9 at M = 28,000 (warm, i.e. a microbenchmark
// re-reading its own previous iteration from the Infinity Cache, 3-10 % slower); in the training step 23.3 k -> 23.7 k samples/s.
// -DVLPET_ROW_AUX=0 builds the default-policy variant for A/B.
#ifndef VLPET_ROW_AUX
#define VLPET_ROW_AUX 2
#endif
__device__ __forceinline__ void glds16_row(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gmem_cv*)gsrc, (lmem_v*)lds_wave_base, 16, 0, VLPET_ROW_AUX);
}

// ---- row tiles: [rows][8 slots of 16 B]; slot = piece ^ swz(row), swizzle applied on the SOURCE side
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// the wave's own 16 rows x 128 bytes of a row-major [M, d] tensor -> its part of the tile (2 instructions)
template <typename IO>
__device__ __forceinline__ void glds_rows(const IO* base, int64_t row0_wave, int64_t M, int d, int feat0,
                                          uint8_t* tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int trow = 16 * wave + 8 * i + (lane >> 3);        // row inside the workgroup tile
        int64_t grow = row0_wave + 8 * i + (lane >> 3);
        if (grow >= M) grow = M - 1;
        const int piece = (lane & 7) ^ swz(trow);
        const IO* src = base + grow * d + feat0 + piece * Geo<IO>::EPP;
        glds16(src, tile + (size_t)(16 * wave + 8 * i) * 128);
    }
}

__device__ __forceinline__ const uint8_t* tile_piece(const uint8_t* tile, int trow, int piece) {
    return tile + ((size_t)trow * 8 + (piece ^ swz(trow))) * 16;
}

// B fragment of down-phase k-step u: 8 consecutive features at stage-local offset 32u + 8g
template <typename IO>
__device__ __forceinline__ Frag<Geo<IO>::NS> tile_bfrag(const uint8_t* tile, int trow, int g, int u) {
    if constexpr (Geo<IO>::NS == 1) {
        Frag<1> f;
        f.p[0] = *reinterpret_cast<const bf16x8*>(tile_piece(tile, trow, 4 * u + g));
        return f;
    } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(tile_piece(tile, trow, 2 * g));
        const f32x4 b = *reinterpret_cast<const f32x4*>(tile_piece(tile, trow, 2 * g + 1));
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        return frag_from_f32<2>(v);
    }
}

// the lane's LW contiguous features (stage-local offset LW*g) as fp32: pieces 2g, 2g+1 for both dtypes
template <typename IO>
__device__ __forceinline__ void tile_lane_vals(const uint8_t* tile, int trow, int g, float* v) {
    if constexpr (Geo<IO>::NS == 1) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(tile_piece(tile, trow, 2 * g));
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(tile_piece(tile, trow, 2 * g + 1));
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = (float)a[j]; v[8 + j] = (float)b[j]; }
    } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(tile_piece(tile, trow, 2 * g));
        const f32x4 b = *reinterpret_cast<const f32x4*>(tile_piece(tile, trow, 2 * g + 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
    }
}

// write the lane's LW outputs into the wave's staging tile (pieces 2g, 2g+1 of row m) ...
template <typename IO>
__device__ __forceinline__ void stage_lane_vals(uint8_t* tile, int trow, int g, const float* v) {
    uint8_t* p0 = const_cast<uint8_t*>(tile_piece(tile, trow, 2 * g));
    uint8_t* p1 = const_cast<uint8_t*>(tile_piece(tile, trow, 2 * g + 1));
    if constexpr (Geo<IO>::NS == 1) {
        bf16x8 a, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)v[j]; b[j] = (__bf16)v[8 + j]; }
        *reinterpret_cast<bf16x8*>(p0) = a;
        *reinterpret_cast<bf16x8*>(p1) = b;
    } else {
        const f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4*>(p0) = a;
        *reinterpret_cast<f32x4*>(p1) = b;
    }
}
// ... and stream the wave's 16 x 128 B out as whole lines (8 lanes per row)
template <typename IO>
__device__ __forceinline__ void store_rows(IO* base, int64_t row0_wave, int64_t M, int d, int feat0,
                                           const uint8_t* tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int trow = 16 * wave + 8 * i + (lane >> 3);
        const int64_t grow = row0_wave + 8 * i + (lane >> 3);
        const int piece = (lane & 7) ^ swz(trow);
        const u32x4 v = *reinterpret_cast<const u32x4*>(tile + ((size_t)trow * 8 + (lane & 7)) * 16);
        if (grow < M)
            *reinterpret_cast<u32x4*>(base + grow * d + feat0 + piece * Geo<IO>::EPP) = v;
    }
}

// weight stage: `kb` KiB from up to two segments, spread over the workgroup's waves (1 KiB per instruction)
template <int WAVES>
__device__ __forceinline__ void glds_weights(const uint8_t* seg0, const uint8_t* seg1, int kb0, int kb1,
                                             uint8_t* dst, int wave, int lane) {
    const int total = kb0 + kb1;
    for (int k = wave; k < total; k += WAVES) {
        const uint8_t* src = k < kb0 ? seg0 + (size_t)k * 1024 : seg1 + (size_t)(k - kb0) * 1024;
        glds16(src + lane * 16, dst + (size_t)k * 1024);
    }
}

template <int NS>
__device__ __forceinline__ Frag<NS> wfrag(const uint8_t* w, int frag, int lane) { return lds_frag<NS>(w, frag, lane); }
