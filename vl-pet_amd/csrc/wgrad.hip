// Column-parallel weight gradients for the PET hot path:
//     Out[c, n] = scale * sum_m P[m, c] * X[m, n]        P: [M, 32*RT] (z or dpre), X: [M, xcols]
// plus column sums of X and P (the bias gradients).  Grid = (64-column slice of X, row chunk, job);
// up to four jobs per launch (dWd, dWu, dWgd, dWgu of one K1 call).
//
// Both operands arrive row-major, but the contraction index is the row m, so each wave loads its
// 32 rows naturally (lane = row, 16/32 contiguous bytes) as an MFMA *A* fragment and multiplies it
// by an identity B fragment: the product comes back in the C/D layout, i.e. transposed
// (lane = column, registers = rows) -- exactly the A / B operand shape of the m-contraction.
// No LDS transpose, no strided global access.  The next 32-row block is loaded while the current one
// is in the MFMAs (register double buffer).  Row-chunk partials are written to a workspace and summed
// by wgrad_finalize (deterministic: no atomics).
#include "pet16.h"      // common.h + glds16
#include <cstdlib>
#include <type_traits>
#include <mutex>
#include <vector>
#include "kernels.h"

size_t wgrad_workspace_bytes(int njobs, int RT, int xcols_max, int row_chunks) {
    const int64_t PR = 32 * RT;
    return (size_t)njobs * row_chunks * (PR * xcols_max + xcols_max + PR) * sizeof(float);
}

void wgrad_plan(int64_t M, int njobs, int xcols_max, int* row_chunks, int64_t* rows_per_chunk) {
    // The streaming kernel's workgroups are (job, 256-column slab, row chunk); the row chunks are what the partial-sum
    // workspace (and the finalize pass) scales with, so the plan takes as few as fill the chip: by default one workgroup
    // per CU (VLPET_WGRAD_WGS overrides; the 64-column-slice kernels of the fp32 / masked paths get 4x as many workgroups
    // from the same plan).
    const int slabs = (xcols_max + 255) / 256;
    int64_t blocks128 = (M + 127) / 128;
    const int64_t target = vlpet_tuning().wgrad_wgs > 0 ? vlpet_tuning().wgrad_wgs : 256;
    int64_t rc = (target + (int64_t)slabs * njobs / 2) / ((int64_t)slabs * njobs);
    if (rc < 1) rc = 1;
    // ... and never more row chunks than put ONE workgroup on every CU of an XCD: the grid is numbered XCD-aware (wg_grid:
    // 8 x nslice x ceil(groups / 8), groups = chunks x jobs), so one chunk too many adds a whole group row -- with 4 jobs and 3
    // slabs 21 chunks are 84 groups = 33 workgroups on four of the 32-CU XCDs, and the CU that carries two of them finishes
    // late: cold-input timings (tools/k1bench.py K1BENCH_COLD=1, profiles/r02_wgrad_cold_crossover.txt) jump by 25-40 %
    // exactly where the plan says 21 (M = 24,000: 94.6 us, 31,616: 116.9) against 20 (28,000: 82.9, 33,200: 93.4).
    // Applies to the default target only (an explicit VLPET_WGRAD_WGS is taken as given).
    const bool explicit_target = vlpet_tuning().wgrad_wgs > 0;
    if (!explicit_target) {
        const int64_t per_xcd = 32 / (slabs < 32 ? slabs : 32);              // (job, chunk) groups per XCD at one workgroup per CU
        const int64_t rc_max = per_xcd > 0 ? (8 * per_xcd) / njobs : 1;
        if (rc_max >= 1 && rc > rc_max) rc = rc_max;
    }
    if (rc > blocks128) rc = blocks128;
    int64_t per = (blocks128 + rc - 1) / rc;          // 128-row blocks per chunk
    rc = (blocks128 + per - 1) / per;
    *row_chunks = (int)rc;
    *rows_per_chunk = per * 128;
}

// XCD-aware workgroup numbering.  The slices of one (row chunk, job) group all read the same P rows, and consecutive
// workgroup ids are dealt round-robin to the 8 XCDs (block b runs on XCD b % 8, each with its own L2): with the natural
// (slice, chunk, job) grid the 12 slices of a group land on 8 different L2s and P is fetched 8 times over (measured L2 hit
// rate 24 %).  Here a 1-D grid is decoded so that all slices of a group share b % 8.
struct WgId { int slice, rc, job; bool ok; };
__device__ __forceinline__ WgId wg_decode(const WgradArgs& a, int nslice) {
    const int b = blockIdx.x, xcd = b & 7, k = b >> 3;
    WgId w;
    w.slice = k % nslice;
    const int g = (k / nslice) * 8 + xcd;
    w.ok = g < a.row_chunks * a.njobs;
    w.rc = g % a.row_chunks;
    w.job = g / a.row_chunks;
    return w;
}
static inline unsigned wg_grid(const WgradArgs& a, int nslice) {
    const int groups = a.row_chunks * a.njobs;
    return 8u * (unsigned)nslice * (unsigned)((groups + 7) / 8);
}

// 8 contiguous IO elements, unconverted (what a prefetched load holds)
template <typename IO> struct Raw8;
template <> struct Raw8<__bf16> { bf16x8 v; };
template <> struct Raw8<float> { f32x4 a, b; };

__device__ __forceinline__ void raw_load(const __bf16* p, Raw8<__bf16>& r) { r.v = *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void raw_load(const float* p, Raw8<float>& r) {
    r.a = reinterpret_cast<const f32x4*>(p)[0];
    r.b = reinterpret_cast<const f32x4*>(p)[1];
}
__device__ __forceinline__ void raw_f32(const Raw8<__bf16>& r, float* v) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)r.v[j];
}
__device__ __forceinline__ void raw_f32(const Raw8<float>& r, float* v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = r.a[j]; v[4 + j] = r.b[j]; }
}

template <typename IO>
__device__ __forceinline__ Frag<IoTraits<IO>::NS> raw_frag8(const Raw8<IO>& r, bool valid, bool drop, uint32_t kb,
                                                            float keep_scale) {
    constexpr int NS = IoTraits<IO>::NS;
    if constexpr (NS == 1) {
        if (!drop) {
            Frag<1> f;
            bf16x8 z;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.0f;
            f.p[0] = valid ? r.v : z;
            return f;
        }
    }
    float v[8];
    raw_f32(r, v);
    if (drop) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ((kb >> j) & 1u) ? v[j] * keep_scale : 0.f;
    }
    if (!valid) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    return frag_from_f32<NS>(v);
}

// transpose a [32 rows x 32 cols] tile given as two natural fragments (cols 0-15, 16-31)
template <int NS>
__device__ __forceinline__ f32x16 transpose_tile(const Frag<NS>& lo16, const Frag<NS>& hi16, bf16x8 I0, bf16x8 I1) {
    f32x16 t = zero16();
#pragma unroll
    for (int p = 0; p < NS; ++p) {
        t = mfma32(lo16.p[p], I0, t);
        t = mfma32(hi16.p[p], I1, t);
    }
    return t;
}

template <typename IO, int RT>
__global__ __launch_bounds__(VLPET_THREADS) void wgrad_kernel(WgradArgs a) {
    constexpr int NS = IoTraits<IO>::NS;
    constexpr int KT = 2 * RT;
    constexpr int PR = 32 * RT;
    constexpr int NV = RT * 2 * 16 + RT + 2;     // per-lane values reduced across the 4 waves
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* red = reinterpret_cast<float*>(smem);

    const WgId wg = wg_decode(a, a.nslice);
    if (!wg.ok) return;                         // uniform per block, before any barrier
    const int jb = wg.job;
    const IO* P = reinterpret_cast<const IO*>(a.job[jb].P);
    const IO* X = reinterpret_cast<const IO*>(a.job[jb].X);
    const bool has_drop = a.job[jb].has_drop != 0;
    const DropSpec drop = drop_resolved(a.job[jb].drop);
    const int ldp = a.job[jb].ldp, ldx = a.job[jb].ldx, xc = a.job[jb].xcols;
    const int n0 = wg.slice * 64;
    if (n0 >= xc) return;
    const int rc = wg.rc;
    const bool first_slice = wg.slice == 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar registers and scalar branches for everything derived from it
    const int m = lane & 31, h = lane >> 5;
    const int64_t r_begin = (int64_t)rc * a.rows_per_chunk;
    int64_t r_end = r_begin + a.rows_per_chunk;
    if (r_end > a.M) r_end = a.M;

    bf16x8 I0, I1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        I0[j] = (m == 8 * h + j) ? (__bf16)1.0f : (__bf16)0.0f;
        I1[j] = (m == 16 + 8 * h + j) ? (__bf16)1.0f : (__bf16)0.0f;
    }

    f32x16 acc[RT][2];
    float csp[RT], csx[2];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) { acc[ct][0] = zero16(); acc[ct][1] = zero16(); csp[ct] = 0.f; }
    csx[0] = csx[1] = 0.f;

    Raw8<IO> rp[KT], rx[4];
    auto load_block = [&](int64_t rb) {
        int64_t row = rb + m;
        if (row >= r_end) row = r_end - 1;
        const IO* pr = P + row * ldp + 8 * h;
        const IO* xr = X + row * ldx + n0 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) raw_load(pr + 16 * ks, rp[ks]);
#pragma unroll
        for (int q = 0; q < 4; ++q) raw_load(xr + 16 * q, rx[q]);
    };

    int64_t rb = r_begin + 32 * wave;
    if (rb < r_end) load_block(rb);
#pragma unroll 1
    for (; rb < r_end; rb += 128) {
        const bool valid = rb + m < r_end;
        Frag<NS> pn[KT], xn[4];
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) pn[ks] = raw_frag8<IO>(rp[ks], valid, false, 0u, 1.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t kb = 0;
            if (has_drop) {
                int64_t row = rb + m;
                if (row >= r_end) row = r_end - 1;
                kb = drop_bits8(drop, row, n0 + 16 * q + 8 * h, ldx);
            }
            xn[q] = raw_frag8<IO>(rx[q], valid, has_drop, kb, drop.keep_scale);
        }
        if (rb + 128 < r_end) load_block(rb + 128);          // prefetch: in flight during the MFMAs below
        Frag<NS> xt[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x16 t = transpose_tile<NS>(xn[2 * nt], xn[2 * nt + 1], I0, I1);
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i] = t[i]; csx[nt] += t[i]; }
            xt[nt][0] = frag_from_f32<NS>(v);
            xt[nt][1] = frag_from_f32<NS>(v + 8);
        }
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) {
            f32x16 t = transpose_tile<NS>(pn[2 * ct], pn[2 * ct + 1], I0, I1);
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i] = t[i]; csp[ct] += t[i]; }
            Frag<NS> pt0 = frag_from_f32<NS>(v), pt1 = frag_from_f32<NS>(v + 8);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                acc[ct][nt] = mfma_ns<NS>(pt0, xt[nt][0], acc[ct][nt]);
                acc[ct][nt] = mfma_ns<NS>(pt1, xt[nt][1], acc[ct][nt]);
            }
        }
    }

    // ---- reduce the four waves (fixed order) and emit this chunk's partial
    if (wave > 0) {
        float* dst = red + (size_t)(wave - 1) * NV * 64;
        int k = 0;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) dst[(k++) * 64 + lane] = acc[ct][nt][i];
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) dst[(k++) * 64 + lane] = csp[ct];
        dst[(k++) * 64 + lane] = csx[0];
        dst[(k++) * 64 + lane] = csx[1];
    }
    __syncthreads();
    if (wave == 0) {
        for (int w = 0; w < 3; ++w) {
            const float* src = red + (size_t)w * NV * 64;
            int k = 0;
#pragma unroll
            for (int ct = 0; ct < RT; ++ct)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[ct][nt][i] += src[(k++) * 64 + lane];
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) csp[ct] += src[(k++) * 64 + lane];
            csx[0] += src[(k++) * 64 + lane];
            csx[1] += src[(k++) * 64 + lane];
        }
        const WgradLayout L = wgrad_layout(a);
        float* part = a.partial + L.off[jb];
        float* tile = part + (int64_t)rc * PR * xc;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int crow = 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h;
                    tile[(int64_t)crow * xc + n0 + 32 * nt + m] = acc[ct][nt][i];
                }
        // column sums: the two half-waves hold disjoint row subsets of the same column
        float* psx = part + (int64_t)a.row_chunks * PR * xc + (int64_t)rc * xc;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float v = csx[nt] + __shfl_xor(csx[nt], 32);
            if (h == 0) psx[n0 + 32 * nt + m] = v;
        }
        if (first_slice) {
            float* psp = part + (int64_t)a.row_chunks * PR * xc + (int64_t)a.row_chunks * xc + (int64_t)rc * PR;
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) {
                const float v = csp[ct] + __shfl_xor(csp[ct], 32);
                if (h == 0) psp[32 * ct + m] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 fast path: the m-contraction operands come from gfx950's LDS transpose read.
//
// Each wave stages its own 32 rows (P: 32 x 64RT bytes, X: 32 x 128 bytes; coalesced 16-byte global loads, one block
// ahead in registers) into a PRIVATE double-buffered row-major LDS tile -- same-wave LDS accesses are ordered, so the
// row loop has no barrier.  ds_read_b64_tr_b16 then hands every lane four consecutive ROWS of its own column
// (measured mapping, tools/tr_probe.hip: within a 16-lane group, out(lane t, elem j) = in(lane 4j + (t >> 2), elem t & 3);
// with lane s of a group addressing row (s >> 2), columns 4(s & 3).. of a [4 rows x 16 cols] block, lane t receives
// column t of the four rows), i.e. exactly the 32x32x16 MFMA operand of a contraction over rows: two reads per
// operand, no identity-MFMA transposes, no fp32 -> bf16 re-conversion, no AGPR shuffling.  Bias gradients (column
// sums) are one extra MFMA against an all-ones A fragment.
typedef short v4s16 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_operand(const uint8_t* tile, int row_stride, int kb, int cb, int lane) {
    // operand of k-step rows kb .. kb+15 and columns cb .. cb+31: lane (i = lane & 31, h = lane >> 5) gets rows kb+8h .. +7 of column cb+i
    const int g = lane >> 4, sl = lane & 15;
    const uint8_t* p = tile + (size_t)(kb + 8 * (g >> 1) + (sl >> 2)) * row_stride + (cb + 16 * (g & 1) + 4 * (sl & 3)) * 2;
    typedef __attribute__((address_space(3))) v4s16 lds_v4;
    const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p));
    const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p + 4 * row_stride));
    typedef short v8s16 __attribute__((ext_vector_type(8)));
    const v8s16 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

template <int RT>
__global__ __launch_bounds__(VLPET_THREADS) void wgrad_tr_kernel(WgradArgs a) {
    constexpr int PR = 32 * RT;
    constexpr int PB = PR * 2;                       // bytes of a P row in the tile
    constexpr int XB = 128;                          // bytes of an X row (64-column slice)
    constexpr int NPP = PB * 32 / 16 / 64;           // 16-byte pieces per lane of a 32-row P block (= 2 RT)
    constexpr int NPR = PB / 16;                     // pieces per P row
    constexpr int BUF = 32 * (PB + XB);              // one buffer of a wave
    constexpr int NV = RT * 2 * 16 + RT + 2;         // per-lane values reduced across the 4 waves (as wgrad_kernel)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const WgId wg = wg_decode(a, a.nslice);
    if (!wg.ok) return;
    const int jb = wg.job;
    const uint8_t* P = reinterpret_cast<const uint8_t*>(a.job[jb].P);
    const uint8_t* X = reinterpret_cast<const uint8_t*>(a.job[jb].X);
    const int ldp = a.job[jb].ldp, ldx = a.job[jb].ldx, xc = a.job[jb].xcols;
    const int n0 = wg.slice * 64;
    if (n0 >= xc) return;
    const int rc = wg.rc;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar registers and scalar branches for everything derived from it
    const int m = lane & 31, h = lane >> 5;
    const int64_t r_begin = (int64_t)rc * a.rows_per_chunk;
    int64_t r_end = r_begin + a.rows_per_chunk;
    if (r_end > a.M) r_end = a.M;
    uint8_t* mine = smem + (size_t)wave * 2 * BUF;
    const bool want_csp = wg.slice == 0;

    f32x16 acc[RT][2];
    // column sums: A fragment "row t all ones" puts sum_k B[k][j] into row t of ONE accumulator, so the RT P tiles
    // (and the two X tiles) share a single f32x16 each: D[t][j] -- lane (j, h = 0), register t
    f32x16 sx = zero16(), sp = zero16();
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) { acc[ct][0] = zero16(); acc[ct][1] = zero16(); }
    bf16x8 erow[RT > 2 ? RT : 2];
#pragma unroll
    for (int t = 0; t < (RT > 2 ? RT : 2); ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) erow[t][j] = (m == t) ? (__bf16)1.0f : (__bf16)0.0f;

    u32x4 rp[NPP], rx[4];
    auto load_block = [&](int64_t rb) {              // unconditional (rows clamped): exact vmcnt bookkeeping by hipcc
#pragma unroll
        for (int i = 0; i < NPP; ++i) {
            const int q = lane + 64 * i;
            int64_t row = rb + q / NPR;
            if (row >= r_end) row = r_end - 1;
            rp[i] = *reinterpret_cast<const u32x4*>(P + row * ldp * 2 + (q % NPR) * 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int64_t row = rb + 8 * i + (lane >> 3);
            if (row >= r_end) row = r_end - 1;
            rx[i] = *reinterpret_cast<const u32x4*>(X + (row * ldx + n0) * 2 + (lane & 7) * 16);
        }
    };
    auto store_block = [&](int64_t rb, int buf) {    // registers -> the wave's tile; rows past the end become zeros
        uint8_t* tp = mine + (size_t)buf * BUF;
        uint8_t* tx = tp + 32 * PB;
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NPP; ++i) {
            const int q = lane + 64 * i;
            const bool ok = rb + q / NPR < r_end;
            *reinterpret_cast<u32x4*>(tp + (size_t)q * 16) = ok ? rp[i] : z;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = rb + 8 * i + (lane >> 3) < r_end;
            *reinterpret_cast<u32x4*>(tx + (size_t)(8 * i + (lane >> 3)) * XB + (lane & 7) * 16) = ok ? rx[i] : z;
        }
    };

    int64_t rb = r_begin + 32 * wave;
    int it = 0;
    if (rb < r_end) {
        load_block(rb);
        store_block(rb, 0);
        load_block(rb + 128);
    }
#pragma unroll 1
    for (; rb < r_end; rb += 128, ++it) {
        const uint8_t* tp = mine + (size_t)(it & 1) * BUF;
        const uint8_t* tx = tp + 32 * PB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bx[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                bx[nt] = tr_operand(tx, XB, 16 * ks, 32 * nt, lane);
                sx = mfma32(erow[nt], bx[nt], sx);
            }
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) {
                const bf16x8 ap = tr_operand(tp, PB, 16 * ks, 32 * ct, lane);
                acc[ct][0] = mfma32(ap, bx[0], acc[ct][0]);
                acc[ct][1] = mfma32(ap, bx[1], acc[ct][1]);
                if (want_csp) sp = mfma32(erow[ct], ap, sp);
            }
        }
        // next block: registers (loaded during the previous iteration) -> the other buffer; request the block after it
        store_block(rb + 128, (it + 1) & 1);
        load_block(rb + 256);
    }

    // ---- reduce the four waves (fixed order) and emit this chunk's partial: identical layout to wgrad_kernel
    float csp[RT], csx[2];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) csp[ct] = h == 0 ? sp[ct] : 0.f;          // row ct of the shared accumulator
    csx[0] = h == 0 ? sx[0] : 0.f;
    csx[1] = h == 0 ? sx[1] : 0.f;
    float* red = reinterpret_cast<float*>(smem);
    __syncthreads();                                 // the row tiles are dead: the reduction buffer aliases them
    if (wave > 0) {
        float* dst = red + (size_t)(wave - 1) * NV * 64;
        int k = 0;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) dst[(k++) * 64 + lane] = acc[ct][nt][i];
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) dst[(k++) * 64 + lane] = csp[ct];
        dst[(k++) * 64 + lane] = csx[0];
        dst[(k++) * 64 + lane] = csx[1];
    }
    __syncthreads();
    if (wave == 0) {
        for (int w = 0; w < 3; ++w) {
            const float* src = red + (size_t)w * NV * 64;
            int k = 0;
#pragma unroll
            for (int ct = 0; ct < RT; ++ct)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[ct][nt][i] += src[(k++) * 64 + lane];
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) csp[ct] += src[(k++) * 64 + lane];
            csx[0] += src[(k++) * 64 + lane];
            csx[1] += src[(k++) * 64 + lane];
        }
        const WgradLayout L = wgrad_layout(a);
        float* part = a.partial + L.off[jb];
        float* tile = part + (int64_t)rc * PR * xc;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int crow = 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h;
                    tile[(int64_t)crow * xc + n0 + 32 * nt + m] = acc[ct][nt][i];
                }
        float* psx = part + (int64_t)a.row_chunks * PR * xc + (int64_t)rc * xc;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            if (h == 0) psx[n0 + 32 * nt + m] = csx[nt];
        if (want_csp) {
            float* psp = part + (int64_t)a.row_chunks * PR * xc + (int64_t)a.row_chunks * xc + (int64_t)rc * PR;
#pragma unroll
            for (int ct = 0; ct < RT; ++ct)
                if (h == 0) psp[32 * ct + m] = csp[ct];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 streaming form (round 3): the op is a split-K GEMM with a 4-byte-per-flop appetite -- at M = 28 k the four jobs of
// a K1 call read 172 MB and do 16.5 GFLOP, i.e. HBM-bound by a factor of 4 -- so the kernel is built as a stream:
//   * a workgroup owns (job, 256-column slab of X, row chunk); wave w owns 64 columns of the slab and ALL 32*RT columns of
//     P: no cross-wave reduction at the end, the P tile is fetched once per workgroup (a quarter of the L2 -> LDS traffic
//     of the 64-column slices above), X rows arrive as 512 contiguous bytes per row;
//   * every tile travels global -> LDS by global_load_lds (no staging registers) into an NSTG-deep ring, NSTG - 1 steps
//     of 32 rows in flight per wave (counted vmcnt; one s_barrier per step for the shared P tile);
//   * operands come out of the row-major tiles with ds_read_b64_tr_b16 (tr_operand above).  The 128-byte rows of the X
//     sub-tiles are swizzled on the SOURCE side (16-byte slot ^= 4 for rows with bit 1 set) so that the four rows a
//     32-lane group reads fall into four different 64-byte bank windows; the 64 / 192-byte rows of P do so natively,
//     the 128 / 384-byte ones (even RT) get the same swizzle.
// Partials and finalize as above (same workspace layout); rows past the end of the chunk are zeroed in LDS.
// The LDS reads of the streaming kernel are inline asm: hipcc orders every LDS read it can see behind ALL outstanding
// global_load_lds of the wave (s_waitcnt vmcnt(0) before the first ds_read of a step -- which would wait for the stages just
// requested and serialise the ring).  The counted vmcnt of the stage is issued by hand (wgs_wait), and so is the lgkmcnt
// before the MFMAs (tr_fence ties the operand registers to the wait so that no MFMA is scheduled above it).
// Dropout on X (the LoRA down-projection gradient, K3): when the job carries the packed mask the forward left behind
// (DropSpec::bits, 1 bit per element), the DROPB instantiation brings each wave's 32 rows x 8 mask bytes along with the stage
// (one more 4-byte global_load_lds per lane: lane = (row, parity) owns the dword of the groups of its parity, drop_pos order)
// and clears the dropped elements of its own X sub-tile in LDS before the transpose reads; 1 / (1 - p) is applied to the sums
// by the finalize kernel.  Round 3: the mask had forced this job onto the per-wave register kernel (46 us per call in the
// LoRA step against 25 for the stream).
#include "cols_common.h"

template <int RT> struct WgsGeo {
    static constexpr int PB = 64 * RT;                 // bytes of a P row
    static constexpr int NPR = PB / 16;                // 16-byte slots per P row
    static constexpr int NPI = 2 * RT;                 // 1 KiB pieces of a 32-row P tile
    static constexpr int PPW = (NPI + 3) / 4;          // pieces per wave (the last ones may be padding)
    static constexpr int PT_B = PPW * 4 * 1024;        // P region of a stage
    static constexpr int XT_B = 32 * 128;              // one wave's X sub-tile
    static constexpr int STG_B = PT_B + 4 * XT_B;
    static constexpr int NI = PPW + 4;                 // global_load_lds instructions per wave and stage
    static constexpr bool PSWZ = (RT % 2) == 0;
};

template <int RT, int NSTG, bool DROPB>
__global__ __launch_bounds__(VLPET_THREADS, (RT <= 3 ? 2 : 1)) void wgrad_stream_kernel(WgradArgs a) {
    using GEO = WgsGeo<RT>;
    constexpr int PR = 32 * RT;
    constexpr int PB = GEO::PB, NPR = GEO::NPR, PPW = GEO::PPW, NI = GEO::NI + (DROPB ? 1 : 0);
    constexpr int STG = GEO::STG_B + (DROPB ? 4 * 256 : 0);      // + the four waves' mask words
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const WgId wg = wg_decode(a, a.nslice);            // nslice = 256-column slabs of the widest job
    if (!wg.ok) return;
    const int jb = wg.job;
    // one batch of scalar loads for everything the prologue needs (dependent kernel-argument loads cost ~1.5 us each)
    const uint8_t* P = reinterpret_cast<const uint8_t*>(a.job[jb].P);
    const uint8_t* X = reinterpret_cast<const uint8_t*>(a.job[jb].X);
    const int ldp_ = a.job[jb].ldp, ldx_ = a.job[jb].ldx;
    const int xc = a.job[jb].xcols;
    const int64_t rpc_ = a.rows_per_chunk, M_ = a.M;
    const bool dropj = DROPB && a.job[jb].has_drop != 0;
    const uint8_t* mbits = dropj ? a.job[jb].drop.bits : P;      // jobs without a mask fetch a dummy word: one vmcnt count for all
    asm volatile("" :: "s"(P), "s"(X), "s"(ldp_), "s"(ldx_), "s"(xc), "s"(rpc_), "s"(M_), "s"(mbits));
    const int64_t ldpb = (int64_t)ldp_ * 2, ldxb = (int64_t)ldx_ * 2;
    if (wg.slice * 256 >= xc) return;
    const int rc = wg.rc;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int n0 = wg.slice * 256 + wave * 64;
    const bool active = n0 < xc;                       // wave-uniform
    const bool want_csp = wg.slice == 0 && wave == 0;
    const int64_t r_begin = (int64_t)rc * rpc_;
    int64_t r_end = r_begin + rpc_;
    if (r_end > M_) r_end = M_;
    const int nsteps = r_end > r_begin ? (int)((r_end - r_begin + 31) >> 5) : 0;

    // per-lane source geometry of the stage pieces
    const int xrow = lane >> 3;                                                   // + 8 i
    const int xsrc = (((lane & 7) ^ (4 * ((lane >> 4) & 1))) * 16) + (active ? n0 : 0) * 2;
    int prow[PPW], psrc[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int q = 64 * (wave + 4 * j) + lane;
        prow[j] = q / NPR;
        int slot = q % NPR;
        if (GEO::PSWZ) slot ^= 4 * ((prow[j] >> 1) & 1);
        psrc[j] = slot * 16;
    }
    const int64_t ldm = dropj ? (int64_t)(ldx_ >> 3) : 0;
    const int msrc = dropj ? ((active ? n0 : 0) >> 3) + 4 * (lane & 1) : 0;
    auto issue = [&](int s) {
        const int64_t rb = r_begin + 32 * (int64_t)s;
        uint8_t* st = smem + (size_t)(s % NSTG) * STG;
        if constexpr (DROPB) {
            int64_t row = rb + (lane >> 1);
            if (row >= r_end) row = r_end - 1;
            __builtin_amdgcn_global_load_lds((gmem_cv*)(mbits + row * ldm + msrc), (lmem_v*)(st + GEO::STG_B + wave * 256), 4, 0, 0);
        }
#ifdef VLPET_WGRAD_EXP
        if (!(a.RT & 0x400))
#endif
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            int64_t row = rb + prow[j];
            if (row >= r_end) row = r_end - 1;
            glds16(P + row * ldpb + psrc[j], st + (size_t)(wave + 4 * j) * 1024);
        }
        uint8_t* sx_ = st + GEO::PT_B + (size_t)wave * GEO::XT_B;
#ifdef VLPET_WGRAD_EXP
        if (!(a.RT & 0x800))
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int64_t row = rb + 8 * i + xrow;
            if (row >= r_end) row = r_end - 1;
            glds16_row(X + row * ldxb + xsrc, sx_ + i * 1024);
        }
    };

    // per-lane LDS byte addresses of the transpose reads (k-step 0, rows 8h + (sl >> 2); the k-step and the +4 rows of the
    // second read are instruction offsets): lane (g = lane >> 4, sl = lane & 15) supplies 8 bytes of row .. at column
    // bytes 32 (g & 1) + 8 (sl & 3) of the 64-byte tile column ct / nt, XOR-swizzled by bit 1 of the row where the tile is
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    uint32_t xa[2], pa[RT];
    {
        const int g = lane >> 4, sl = lane & 15;
        const int row = 8 * (g >> 1) + (sl >> 2), bit = (row >> 1) & 1, lp = 32 * (g & 1) + 8 * (sl & 3);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            xa[nt] = (uint32_t)(GEO::PT_B + wave * GEO::XT_B + row * 128 + 64 * (nt ^ bit) + lp);
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
            pa[ct] = (uint32_t)(row * PB + 64 * (GEO::PSWZ ? (ct ^ bit) : ct) + lp);
    }

    f32x16 acc[RT][2];
    f32x16 sx = zero16(), sp = zero16();
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) { acc[ct][0] = zero16(); acc[ct][1] = zero16(); }
    bf16x8 erow[RT > 2 ? RT : 2];
#pragma unroll
    for (int t = 0; t < (RT > 2 ? RT : 2); ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) erow[t][j] = (m == t) ? (__bf16)1.0f : (__bf16)0.0f;

#pragma unroll
    for (int s = 0; s < NSTG - 1; ++s)
        if (s < nsteps) issue(s);

#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        int pending = nsteps - 1 - s;
        if (pending > NSTG - 2) pending = NSTG - 2;
        wgs_wait<NI, NSTG - 2>(pending);
        uint8_t* st = smem + (size_t)(s % NSTG) * STG;
        uint8_t* tx = st + GEO::PT_B + (size_t)wave * GEO::XT_B;
        if constexpr (DROPB) {
            if (dropj && active) {                   // own X pieces and mask words have landed: clear the dropped elements
                const uint32_t sb = lds0 + (uint32_t)((s % NSTG) * STG);
                const int row = lane >> 1, b = (row >> 1) & 1;
                uint32_t kw;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(kw) : "v"(sb + (uint32_t)(GEO::STG_B + wave * 256 + lane * 4)) : "memory");
                if (b) kw = (kw >> 16) | (kw << 16);             // rows with the slot ^ 4 swizzle: LDS slot t holds group t ^ 2
                const uint32_t xb = sb + (uint32_t)(GEO::PT_B + wave * GEO::XT_B + row * 128 + 16 * (lane & 1));
                u32x4 v[4];
                lds_read16<0>(v[0], xb); lds_read16<32>(v[1], xb); lds_read16<64>(v[2], xb); lds_read16<96>(v[3], xb);
                lgkm_fence(v[0]); lgkm_tie(v[1]); lgkm_tie(v[2]); lgkm_tie(v[3]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int lo = ((int)(kw << (31 - (8 * t + 2 * q)))) >> 31;
                        const int hi = ((int)(kw << (31 - (8 * t + 2 * q + 1)))) >> 31;
                        v[t][q] &= __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x07060100u);
                    }
                lds_write16<0>(xb, v[0]); lds_write16<32>(xb, v[1]); lds_write16<64>(xb, v[2]); lds_write16<96>(xb, v[3]);
            }
        }
        const int valid = (int)(r_end - (r_begin + 32 * (int64_t)s));            // rows of this step that exist
        const bool tail = valid < 32;
        if (tail) {                                                               // own X pieces have landed: zero the missing rows
            const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (8 * i + xrow >= valid) *reinterpret_cast<u32x4*>(tx + (size_t)i * 1024 + lane * 16) = z;
        }
        __builtin_amdgcn_s_barrier();
        if (s + NSTG - 1 < nsteps) issue(s + NSTG - 1);
        if (tail) {                                                               // everybody's P pieces have landed
            const u32x4 z = {0u, 0u, 0u, 0u};
            for (int q = tid; q < 32 * NPR; q += VLPET_THREADS)
                if (q / NPR >= valid) *reinterpret_cast<u32x4*>(st + (size_t)q * 16) = z;
            __syncthreads();
        }
#ifdef VLPET_WGRAD_EXP
        if (a.RT & 0x200) continue;                  // experiment: the stream without the products
#endif
        if (active) {
            const uint32_t sb = lds0 + (uint32_t)((s % NSTG) * STG);
            TrOp bx[2][2], ap[2][RT];
            auto reads = [&](auto ksc) {
                constexpr int ks = decltype(ksc)::value;
                tr_read<16 * ks * 128, (16 * ks + 4) * 128>(bx[ks][0], sb + xa[0]);
                tr_read<16 * ks * 128, (16 * ks + 4) * 128>(bx[ks][1], sb + xa[1]);
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) tr_read<16 * ks * PB, (16 * ks + 4) * PB>(ap[ks][ct], sb + pa[ct]);
            };
            auto products = [&](int ks, bool fenced_first) {
                if (!fenced_first) tr_tie(bx[ks][0]);
                tr_tie(bx[ks][1]);
                const bf16x8 b0 = tr_val(bx[ks][0]), b1 = tr_val(bx[ks][1]);
                sx = mfma32(erow[0], b0, sx);
                sx = mfma32(erow[1], b1, sx);
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) {
                    tr_tie(ap[ks][ct]);
                    const bf16x8 av = tr_val(ap[ks][ct]);
                    acc[ct][0] = mfma32(av, b0, acc[ct][0]);
                    acc[ct][1] = mfma32(av, b1, acc[ct][1]);
                    if (want_csp) sp = mfma32(erow[ct], av, sp);
                }
            };
            if constexpr (RT <= 3) {                 // both k-steps' operands requested before the first product
                reads(std::integral_constant<int, 0>{});
                reads(std::integral_constant<int, 1>{});
                tr_fence(bx[0][0]);
                products(0, true);
                products(1, false);
            } else {                                 // 12 accumulator tiles: one k-step's operands at a time
                reads(std::integral_constant<int, 0>{});
                tr_fence(bx[0][0]);
                products(0, true);
                reads(std::integral_constant<int, 1>{});
                tr_fence(bx[1][0]);
                products(1, true);
            }
        }
    }

    if (!active) return;
#ifdef VLPET_WGRAD_EXP
    if ((a.RT & 0x100) && acc[0][0][0] != 123.456f) return;      // experiment: no partial stores
    a.RT &= 0xff;
#endif
    const WgradLayout L = wgrad_layout(a);
    float* part = a.partial + L.off[jb];
    float* tile = part + (int64_t)rc * PR * xc;
#pragma unroll
    for (int ct = 0; ct < RT; ++ct)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int crow = 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h;
                tile[(int64_t)crow * xc + n0 + 32 * nt + m] = acc[ct][nt][i];
            }
    float* psx = part + (int64_t)a.row_chunks * PR * xc + (int64_t)rc * xc;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
        if (h == 0) psx[n0 + 32 * nt + m] = sx[nt];
    if (want_csp) {
        float* psp = part + (int64_t)a.row_chunks * PR * xc + (int64_t)a.row_chunks * xc + (int64_t)rc * PR;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
            if (h == 0) psp[32 * ct + m] = sp[ct];
    }
}

// sum the row-chunk partials, apply the scale, drop the rank padding, write in the parameter's layout.
// A block owns a [32 c x 64 n] tile: reads run along n (the contiguous axis of the partials, 8 independent sums per
// thread in flight), writes are contiguous in the OUTPUT layout -- along n for [r, d] gradients, and through an LDS
// transpose along c for the transposed [d, r] ones (a direct store there scatters 4-byte writes r*4 bytes apart and
// cost 32 us per launch).  The last blocks of a job sum the bias-gradient partials.
// Round 3: a block owns a [16 c x 64 n] tile, ONE float4 per thread, and keeps up to 16 row-chunk loads in flight
// (the first version walked 4 chunks per trip with 144 blocks: 23 us for a 19 MB reduction, all of it exposed latency).
__host__ __device__ inline int finalize_tiles(int PR, int xcols) { return (PR / 16) * (xcols / 64); }

// The kernel's own arguments are a compact per-job record filled on the host: the first version indexed WgradArgs and
// rebuilt the workspace layout on the device -- six DEPENDENT scalar loads from the kernel-argument segment before the first
// partial was requested, ~13 us of the kernel's 21 (an empty kernel with the same preamble took 19 us, one that returns at
// the top 5 us; profiles/r02_wgrad_stream_probes.md).  Here a block issues one batch of scalar loads, then every row-chunk
// load of its tile, then the sum.
struct FinJob {
    const float* part;      // this job's partial block
    float* out;
    float* colsum_x;
    float* colsum_p;
    int xcols, out_rows, ldo, transposed;
    float scale;
    int ntile;
};
struct FinArgs {
    FinJob job[4];
    int RC, PR;
    int njobs, blocks;      // (used by the batched launch: a call's own grid inside the batch's)
};

// the partials are read once, by one workgroup; non-temporal loads measured no faster here (K1 backward pass 2 + finalize 75.1 / 75.7 us
// with them against 73.8 / 74.4 us without at M = 28,000, same box: profiles/r04_finalize_ab.txt), unlike for the row streams (pet16.h;
// -DVLPET_FIN_NT=1 builds the non-temporal variant for A/B)
#ifndef VLPET_FIN_NT
#define VLPET_FIN_NT 0
#endif
#if VLPET_FIN_NT
#define VLPET_FIN_LOAD(p) __builtin_nontemporal_load(p)
#else
#define VLPET_FIN_LOAD(p) (*(p))
#endif
__device__ __forceinline__ void finalize_body(const FinArgs& a, float (*tile)[65]) {
    const FinJob J = a.job[blockIdx.y];
    const int PR = a.PR, RC = a.RC;
    const float* part = J.part;
    const int xc = J.xcols, R = J.out_rows;
    const int ntile = J.ntile;
    const int t = threadIdx.x;
    if ((int)blockIdx.x < ntile) {
        const int c0 = 16 * ((int)blockIdx.x / (xc / 64)), n0 = 64 * ((int)blockIdx.x % (xc / 64));
        if (c0 >= R) return;                                     // rank padding: nothing to write
        const int cc = t >> 4, nn = 4 * (t & 15);
        const float* p0 = part + (int64_t)(c0 + cc) * xc + n0 + nn;
        const int64_t cstride = (int64_t)PR * xc;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        // Round 4: every row-chunk load of the tile is requested before the first add (one trip of up to 48 loads, or 24 when
        // there are at most 24 chunks, instead of trips of 16: with 40-42 chunks the kernel waited out three memory round trips,
        // now one); chunk order kept, indices past the end re-read the last chunk and are not added.
        auto batch = [&](auto NB, int rc0) {
            constexpr int N = decltype(NB)::value;
            f32x4 v[N];
#pragma unroll
            for (int q = 0; q < N; ++q) {
                const int r2 = rc0 + q < RC ? rc0 + q : RC - 1;
                v[q] = VLPET_FIN_LOAD(reinterpret_cast<const f32x4*>(p0 + r2 * cstride));
            }
            __builtin_amdgcn_sched_barrier(0);                           // (every request before the first add)
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < N; ++q) s += rc0 + q < RC ? v[q] : zero;
        };
        if (RC <= 24) batch(std::integral_constant<int, 24>{}, 0);
        else
            for (int rc = 0; rc < RC; rc += 48) batch(std::integral_constant<int, 48>{}, rc);
        s = s * J.scale;
        if (!J.transposed) {
            if (c0 + cc < R) {
                float* o = J.out + (int64_t)(c0 + cc) * J.ldo + n0 + nn;
                if ((reinterpret_cast<uintptr_t>(o) & 15) == 0) *reinterpret_cast<f32x4*>(o) = s;     // (a gradient view in the
                else { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3]; }                           // flat buffer may sit at any 4-byte offset)
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[cc][nn + j] = s[j];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = t + 256 * k, n2 = e >> 4, c2 = e & 15;      // c fastest: contiguous in out[n * ldo + c]
                if (c0 + c2 < R) J.out[(int64_t)(n0 + n2) * J.ldo + c0 + c2] = tile[c2][n2];
            }
        }
        return;
    }
    const int64_t gid = (int64_t)((int)blockIdx.x - ntile) * 256 + t;
    if (gid < xc) {
        if (J.colsum_x != nullptr) {
            const float* p = part + (int64_t)RC * PR * xc;
            float sum = 0.f;
            int rc = 0;
            for (; rc + 8 <= RC; rc += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = p[(int64_t)(rc + q) * xc + gid];
#pragma unroll
                for (int q = 0; q < 8; ++q) sum += v[q];
            }
            for (; rc < RC; ++rc) sum += p[(int64_t)rc * xc + gid];
            J.colsum_x[gid] = sum * J.scale;
        }
    } else if (gid < xc + R) {
        if (J.colsum_p != nullptr) {
            const int c = (int)(gid - xc);
            const float* p = part + (int64_t)RC * PR * xc + (int64_t)RC * xc;
            float sum = 0.f;
            int rc = 0;
            for (; rc + 8 <= RC; rc += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = p[(int64_t)(rc + q) * PR + c];
#pragma unroll
                for (int q = 0; q < 8; ++q) sum += v[q];
            }
            for (; rc < RC; ++rc) sum += p[(int64_t)rc * PR + c];
            J.colsum_p[c] = sum;
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_finalize_kernel(FinArgs a) {
    __shared__ float tile[16][65];
    finalize_body(a, tile);
}

// Several calls' finalize passes as ONE launch (round 5).  A training step runs one per adapter call (19 in a BART-base step, 37 in
// T5-base), each a few dozen to a few hundred workgroups of pure latency; at the per-rank sizes of an 8-GPU run a launch boundary
// (~5 us in a replayed graph, profiles/r05_rank1of8_timeline.md) is as long as the sum itself.  Nobody reads a weight gradient before
// the optimizer, so a caller may queue them (vlpet_finalize_defer) and launch the queue once (vlpet_finalize_flush): blockIdx.z = the
// call, (x, y) its own grid; same per-call arithmetic and order of additions, so the results are bit-identical.
#define VLPET_FIN_BATCH 16
struct FinBatch { FinArgs a[VLPET_FIN_BATCH]; };
static_assert(sizeof(FinBatch) <= 4096, "kernel-argument segment");
__global__ __launch_bounds__(256) void wgrad_finalize_batch_kernel(FinBatch b) {
    __shared__ float tile[16][65];
    const FinArgs& a = b.a[blockIdx.z];
    if ((int)blockIdx.y >= a.njobs || (int)blockIdx.x >= a.blocks) return;
    finalize_body(a, tile);
}

// The switch is per host thread (a caller brackets single calls with it); the QUEUE is per process: a training framework runs its
// backward calls on its autograd engine's worker thread and flushes from the thread that called backward().
static thread_local bool g_fin_defer = false;
static std::mutex g_fin_mutex;
static std::vector<FinArgs> g_fin_queue;

int finalize_defer(int on) { const int was = g_fin_defer ? 1 : 0; g_fin_defer = on != 0; return was; }
int finalize_pending() { std::lock_guard<std::mutex> lk(g_fin_mutex); return (int)g_fin_queue.size(); }
void finalize_discard() { std::lock_guard<std::mutex> lk(g_fin_mutex); g_fin_queue.clear(); }
hipError_t finalize_flush(hipStream_t stream) {
    std::vector<FinArgs> q;
    { std::lock_guard<std::mutex> lk(g_fin_mutex); q.swap(g_fin_queue); }
    const size_t n = q.size();
    for (size_t k0 = 0; k0 < n; k0 += VLPET_FIN_BATCH) {
        const int nb = (int)(n - k0 < VLPET_FIN_BATCH ? n - k0 : VLPET_FIN_BATCH);
        if (nb == 1) {
            const FinArgs& f = q[k0];
            hipLaunchKernelGGL(wgrad_finalize_kernel, dim3((unsigned)f.blocks, f.njobs), dim3(256), 0, stream, f);
            continue;
        }
        FinBatch b{};
        int bx = 0, by = 0;
        for (int k = 0; k < nb; ++k) {
            b.a[k] = q[k0 + k];
            if (b.a[k].blocks > bx) bx = b.a[k].blocks;
            if (b.a[k].njobs > by) by = b.a[k].njobs;
        }
        hipLaunchKernelGGL(wgrad_finalize_batch_kernel, dim3((unsigned)bx, (unsigned)by, (unsigned)nb), dim3(256), 0, stream, b);
    }
    return hipGetLastError();
}

static hipError_t launch_finalize(const WgradArgs& a, int RT, int xmax, int rmax, hipStream_t stream, bool mask_unscaled = false) {
    const int PR = 32 * RT;
    const int blocks = finalize_tiles(PR, xmax) + (xmax + rmax + 255) / 256;   // tiles of every job fit: xcols <= xmax
    const WgradLayout L = wgrad_layout(a);
    FinArgs f{};
    f.RC = a.row_chunks; f.PR = PR; f.njobs = a.njobs; f.blocks = blocks;
    for (int j = 0; j < a.njobs; ++j) {
        const WgradJob& J = a.job[j];
        FinJob& F = f.job[j];
        F.part = a.partial + L.off[j];
        F.out = J.out; F.colsum_x = J.colsum_x; F.colsum_p = J.colsum_p;
        F.xcols = J.xcols; F.out_rows = J.out_rows; F.ldo = J.ldo; F.transposed = J.transposed;
        F.scale = J.scale; F.ntile = finalize_tiles(PR, J.xcols);
        if (mask_unscaled && J.has_drop) F.scale *= J.drop.keep_scale;      // the streaming kernel only clears the dropped elements
    }
    if (g_fin_defer) {      // (launched by finalize_flush)
        std::lock_guard<std::mutex> lk(g_fin_mutex);
        g_fin_queue.push_back(f);
        return hipSuccess;
    }
    hipLaunchKernelGGL(wgrad_finalize_kernel, dim3((unsigned)blocks, a.njobs), dim3(256), 0, stream, f);
    return hipGetLastError();
}

hipError_t launch_wgrad_finalize(const WgradArgs& a, hipStream_t stream) {
    int xmax = 0, rmax = 0;
    for (int j = 0; j < a.njobs; ++j) {
        if (a.job[j].xcols > xmax) xmax = a.job[j].xcols;
        if (a.job[j].out_rows > rmax) rmax = a.job[j].out_rows;
    }
    return launch_finalize(a, a.RT, xmax, rmax, stream);
}

template <typename IO, int RT>
static hipError_t launch_one(const WgradArgs& a, hipStream_t stream) {
    constexpr int NV = RT * 2 * 16 + RT + 2;
    int xmax = 0, rmax = 0;
    bool plain = true;                       // no dropout mask on X, 16-byte aligned rows: the transpose-read path applies
    bool streamable = true, masked = false;  // the streaming kernel also takes a mask if it is the forward's packed one
    for (int j = 0; j < a.njobs; ++j) {
        const WgradJob& J = a.job[j];
        if (J.xcols > xmax) xmax = J.xcols;
        if (J.out_rows > rmax) rmax = J.out_rows;
        if (J.ldp % 8 != 0 || J.ldx % 8 != 0) plain = streamable = false;
        if (J.has_drop) {
            plain = false;
            masked = true;
            if (J.drop.bits == nullptr || J.drop.keep != nullptr || J.colsum_x != nullptr || J.ldx % 64 != 0) streamable = false;
        }
    }
    // bf16, r <= 96, no dropout mask: the LDS transpose-read kernel (ds_read_b64_tr_b16 operands, no identity-MFMA
    // transposes, no fp32 -> bf16 re-conversion).  Round 2, after the XCD-aware numbering: 83.0 vs 86.7 us at M = 28 k,
    // and in the bench 96.6 vs 104.3 us per K1 call, 235 vs 279 us for the K4 weight gradient (16,415 vs 16,319 samples/s;
    // profiles/r02_kbench_wgrad_variants.txt).  VLPET_WGRAD_TR=0 selects the identity-transpose kernel.
    const bool use_tr = vlpet_tuning().wgrad_tr != 0;
    hipError_t e;
    // bf16, no dropout mask: the streaming kernel (256-column slabs, LDS-DMA ring).  VLPET_WGRAD_STREAM=0 falls back to the
    // per-wave transpose-read kernel below (A/B).
    const bool use_stream = vlpet_tuning().wgrad_stream != 0;
    if constexpr (std::is_same<IO, __bf16>::value) {
        if (streamable && use_stream) {
            // ring depth: 3 stages (72 KiB at RT = 3: two workgroups per CU); VLPET_WGRAD_NSTG=4 for A/B (one per CU, one more stage in flight)
            const int nstg = vlpet_tuning().wgrad_nstg;
            WgradArgs b = a; b.nslice = (xmax + 255) / 256;
#ifdef VLPET_WGRAD_EXP
            b.RT |= vlpet_tuning().wgs_mode << 8;
#endif
            auto go = [&](auto kern, int NSTG) -> hipError_t {
                const size_t lds = (size_t)NSTG * (WgsGeo<RT>::STG_B + (masked ? 1024 : 0));
                hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e2 != hipSuccess) return e2;
                hipLaunchKernelGGL(kern, dim3(wg_grid(b, b.nslice)), dim3(VLPET_THREADS), lds, stream, b);
                return hipGetLastError();
            };
            if (masked) e = go(wgrad_stream_kernel<RT, 3, true>, 3);
            else e = nstg == 4 ? go(wgrad_stream_kernel<RT, 4, false>, 4) : go(wgrad_stream_kernel<RT, 3, false>, 3);
            if (e != hipSuccess) return e;
            return launch_finalize(a, RT, xmax, rmax, stream, masked);
        }
    }
    if constexpr (std::is_same<IO, __bf16>::value && RT <= 3) {
        if (plain && use_tr) {
            const size_t tiles = (size_t)4 * 2 * 32 * (64 * RT + 128);
            const size_t redb = (size_t)3 * NV * 64 * sizeof(float);
            const size_t lds = tiles > redb ? tiles : redb;
            auto kern = wgrad_tr_kernel<RT>;
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            WgradArgs b = a; b.nslice = xmax / 64;
            hipLaunchKernelGGL(kern, dim3(wg_grid(b, b.nslice)), dim3(VLPET_THREADS), lds, stream, b);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            return launch_finalize(a, RT, xmax, rmax, stream);
        }
    }
    const size_t lds = (size_t)3 * NV * 64 * sizeof(float);
    auto kern = wgrad_kernel<IO, RT>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    WgradArgs b = a; b.nslice = xmax / 64;
    hipLaunchKernelGGL(kern, dim3(wg_grid(b, b.nslice)), dim3(VLPET_THREADS), lds, stream, b);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_finalize(a, RT, xmax, rmax, stream);
}

template <typename IO>
static hipError_t launch_io(const WgradArgs& a, hipStream_t stream) {
    switch (a.RT) {
        case 1: return launch_one<IO, 1>(a, stream);
        case 3: return launch_one<IO, 3>(a, stream);
        case 6: return launch_one<IO, 6>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_wgrad(const WgradArgs& a, int io_fp32, hipStream_t stream) {
    return io_fp32 ? launch_io<float>(a, stream) : launch_io<__bf16>(a, stream);
}
