// v4 building blocks: 32-row waves on v_mfma_f32_32x32x16_bf16 (half the LDS fragment bytes per flop
// of the 16x16x32 form, which measured LDS-bound) on the v3 memory system: every global read arrives by
// global_load_lds as whole 128-byte lines, outputs leave as whole lines.  Layout specification:
// tests/packing_spec.py (section v4), checked lane by lane in tests/test_layout_model32.py.
//
//   lane (m = lane & 31, h = lane >> 5) owns activation row m of its wave's 32 rows;
//   a *stage* moves 128 bytes of every row: FE = 64 features (bf16 IO) or 32 (fp32 IO);
//   MFMA k-slot (h, j) of k-step u is stage-local feature 16u + 8h + j;
//   after the up phase a lane holds LW = FE/2 contiguous features (64 bytes).
#pragma once
#include "pet16.h"      // glds16, swz, tile_piece, wfrag, zero4 ...

template <typename IO> struct Geo4 {
    static constexpr int NS = IoTraits<IO>::NS;
    static constexpr int FE = 64 / NS;      // features per stage
    static constexpr int KU = FE / 16;      // MFMA k-steps per down-phase stage
    static constexpr int NV = FE / 32;      // 32-feature n-tiles per up-phase stage
    static constexpr int LW = FE / 2;       // contiguous output features per lane
    static constexpr int E4 = FE / 16;      // k-steps of the feature contraction (backward) per stage
    static constexpr int EPP = 16 / (int)sizeof(IO);
};

// B fragment of down-phase k-step u: 8 consecutive features at stage-local offset 16u + 8h
template <typename IO>
__device__ __forceinline__ Frag<Geo4<IO>::NS> tile_bfrag4(const uint8_t* tile, int trow, int h, int u) {
    if constexpr (Geo4<IO>::NS == 1) {
        Frag<1> f;
        f.p[0] = *reinterpret_cast<const bf16x8*>(tile_piece(tile, trow, 2 * u + h));
        return f;
    } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(tile_piece(tile, trow, 2 * (2 * u + h)));
        const f32x4 b = *reinterpret_cast<const f32x4*>(tile_piece(tile, trow, 2 * (2 * u + h) + 1));
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        return frag_from_f32<2>(v);
    }
}

// the lane's LW contiguous features (64 bytes = pieces 4h .. 4h+3 for both dtypes) as fp32
template <typename IO>
__device__ __forceinline__ void tile_lane_vals4(const uint8_t* tile, int trow, int h, float* v) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint8_t* src = tile_piece(tile, trow, 4 * h + p);
        if constexpr (Geo4<IO>::NS == 1) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(src);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[8 * p + j] = (float)a[j];
        } else {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * p + j] = a[j];
        }
    }
}

template <typename IO>
__device__ __forceinline__ void stage_lane_vals4(uint8_t* tile, int trow, int h, const float* v) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        uint8_t* dst = const_cast<uint8_t*>(tile_piece(tile, trow, 4 * h + p));
        if constexpr (Geo4<IO>::NS == 1) {
            bf16x8 a;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = (__bf16)v[8 * p + j];
            *reinterpret_cast<bf16x8*>(dst) = a;
        } else {
            const f32x4 a = {v[4 * p], v[4 * p + 1], v[4 * p + 2], v[4 * p + 3]};
            *reinterpret_cast<f32x4*>(dst) = a;
        }
    }
}

// elements 8e .. 8e+7 of the lane's LW contiguous features (bf16: piece 4h+e; fp32: pieces 4h+2e, 4h+2e+1) -- lets an
// epilogue walk its 64 bytes in MFMA-fragment-sized steps with 8 live values instead of LW
template <typename IO>
__device__ __forceinline__ void tile_lane_vals8(const uint8_t* tile, int trow, int h, int e, float* v) {
    if constexpr (Geo4<IO>::NS == 1) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(tile_piece(tile, trow, 4 * h + e));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
    } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(tile_piece(tile, trow, 4 * h + 2 * e));
        const f32x4 b = *reinterpret_cast<const f32x4*>(tile_piece(tile, trow, 4 * h + 2 * e + 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
    }
}
template <typename IO>
__device__ __forceinline__ void stage_lane_vals8(uint8_t* tile, int trow, int h, int e, const float* v) {
    if constexpr (Geo4<IO>::NS == 1) {
        bf16x8 a;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (__bf16)v[j];
        *reinterpret_cast<bf16x8*>(const_cast<uint8_t*>(tile_piece(tile, trow, 4 * h + e))) = a;
    } else {
        const f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4*>(const_cast<uint8_t*>(tile_piece(tile, trow, 4 * h + 2 * e))) = a;
        *reinterpret_cast<f32x4*>(const_cast<uint8_t*>(tile_piece(tile, trow, 4 * h + 2 * e + 1))) = b;
    }
}

// per-lane addressing of the four row-piece instructions of a wave (8 rows x 128 B each)
struct RowLanes {
    int64_t off[4];     // byte offset of the lane's 16-byte piece inside a row-major [M, d] tensor (stage excluded)
    bool ok[4];         // row exists
    int n_inst;         // instructions whose 8 rows are not all past the end (wave-uniform)
};
template <typename IO>
__device__ __forceinline__ RowLanes row_lanes(int64_t row0_wave, int64_t M, int d, int wave, int lane) {
    RowLanes r;
    r.n_inst = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tr = 32 * wave + 8 * i + (lane >> 3);
        const int64_t grow_raw = row0_wave + 8 * i + (lane >> 3);
        const int64_t grow = grow_raw < M ? grow_raw : M - 1;
        const int piece = (lane & 7) ^ swz(tr);
        r.off[i] = (grow * d) * (int64_t)sizeof(IO) + piece * 16;
        r.ok[i] = grow_raw < M;
        if (row0_wave + 8 * i < M) ++r.n_inst;
    }
    return r;
}
// wave's 32 x 128 B of stage `so` bytes -> its rows of `tile`
__device__ __forceinline__ void glds_rows4(const uint8_t* base, const RowLanes& rl, int so, uint8_t* tile, int wave) {
    uint8_t* dst = tile + (size_t)(32 * wave) * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16_row(base + rl.off[i] + so, dst + i * 1024);
}
// the wave's rows of `tile` (slot order) -> whole-line stores
__device__ __forceinline__ void store_rows4(uint8_t* base, const RowLanes& rl, int so, const uint8_t* tile,
                                            int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(tile + ((size_t)(32 * wave + 8 * i + (lane >> 3)) * 8 + (lane & 7)) * 16);
        if (rl.ok[i]) *reinterpret_cast<u32x4*>(base + rl.off[i] + so) = v;
    }
}

__device__ __forceinline__ void wait_vm(int n) {
    // counted wait: at most n vector-memory operations of this wave may still be in flight; LDS drained
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13) lgkmcnt(0)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15) lgkmcnt(0)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); break;   // n > 16: stricter is safe
    }
}

// fp32 bias vectors -> LDS with all loads of a thread in flight before the first is waited for (a plain copy loop
// compiles to load -> vmcnt(0) -> ds_write per trip: one global latency per 256 floats, paid in every prologue)
template <int THREADS>
__device__ __forceinline__ void copy_bias(float* dst, const float* src, int n, int tid) {
    for (int base = 0; base < n; base += 4 * THREADS) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = base + k * THREADS + tid;
            v[k] = src[i < n ? i : n - 1];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = base + k * THREADS + tid;
            if (i < n) dst[i] = v[k];
        }
    }
}
