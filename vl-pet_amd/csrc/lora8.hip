// K3 at rank <= 8: the LoRA delta as a streaming row kernel, no matrix cores.
//
//     out = base + scaling * ( (dropout(x) A^T) B^T )        A [r, d], B [d, r], r <= 8          (lora/controller.py:56-70,
//                                                                                               scripts/image-text/single_lora.sh:26)
//
// On the MFMA path (pet_fwd.hip / pet_fwd2p.hip) a rank of 8 runs a 32-wide tile, three quarters of it zero padding, inside a
// latency chain per workgroup whose dropout generator sits on the critical path (0.62 of the 3-unit stream without dropout, 0.41 with
// it at 28,000 rows).  The arithmetic at r <= 8 is 2 x 8 multiply-adds per element: a contraction of eight, not a matrix
// product.  Here a WAVE owns a row and walks rows with the next one prefetched (the K5 shape); a lane owns the 8-byte chunks
// lane, lane + 64, ... of the row (d / 256 chunks of four features: every load / store instruction moves 512 contiguous bytes and
// every lane has the same work -- 16-byte pieces would leave half the lanes idle on the second piece at d = 768).  The two weight
// slices of its features stay in registers as packed bf16 for the whole launch (A[0..7][its features], B[its features][0..7]: 96
// registers at d = 768) and both contractions are v_dot2c_f32_bf16 on the packed operands as loaded: z_c += <x pair, A_c pair>,
// one reduce-scatter over the wave for the eight sums (10 cross-lane moves, then eight v_readlane: z lives in scalar registers),
// z rounded to bf16 (the MFMA path rounds it there too), out_f = base_f + scaling * sum of four <z pair, B_f pair>.  ~100 dot
// products + ~150 other VALU operations per row and wave: bound by the row stream, with the generator behind it as in K5.
// Dropout flags come per 8-feature group from the shared generator (rng.h): the row's d / 8 groups are spread over the lanes (two
// rounds at d = 768, the second half full) and every lane fetches the groups of its chunks with ds_bpermute -- 1.5 generator calls
// per lane and row instead of one per chunk.
// The weights come out of the SAME packed pair the MFMA kernels use (pack.hip; tests/packing_spec.py pack_down4 / pack_up4, bf16
// plane), so the caller's pack cache serves both forms; the training form leaves z ([M, 32] bf16, columns >= 8 zero) and the packed
// mask in the layout pet_fwd.hip writes, so the backward (pet_cols_ng.hip) does not care which forward ran.
#include "common.h"
#include "kernels.h"
#include "rowops.h"
#include "rng.h"

constexpr int L8_WAVES = 4;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
#ifdef L8_NO_DOT2
    return __uint_as_float(a << 16) * __uint_as_float(b << 16) + (__uint_as_float(a & 0xffff0000u) * __uint_as_float(b & 0xffff0000u) + c);
#else
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
#endif
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// sums of z[0..7] over the 64 lanes, each returned wave-uniform: a reduce-scatter (lane halves keep four, then two, then one of
// the eight sums), three butterfly steps on the one value left, and a v_readlane per sum (lane 8 c ends up with the total of c)
__device__ __forceinline__ void wave_sum8(float (&z)[8], int lane) {
    const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0;
    float k4[4], k2[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) k4[j] = (b5 ? z[4 + j] : z[j]) + __shfl_xor(b5 ? z[j] : z[4 + j], 32, 64);
#pragma unroll
    for (int j = 0; j < 2; ++j) k2[j] = (b4 ? k4[2 + j] : k4[j]) + __shfl_xor(b4 ? k4[j] : k4[2 + j], 16, 64);
    float v = (b3 ? k2[1] : k2[0]) + __shfl_xor(b3 ? k2[0] : k2[1], 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
#pragma unroll
    for (int c = 0; c < 8; ++c) z[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 8 * c));
}

// NCH = chunks (4 features, 8 bytes) per lane: d = 256 * NCH
template <int NCH, int DROP>      // DROP: 0 no dropout, 1 the generator's mask (no mask LOADS in the row loop: hipcc keeps counted waits and the
                                  // prefetched row overlaps -- tail.hip's FULL note), 2 any source (explicit / packed mask)
__global__ __launch_bounds__(L8_WAVES * 64) void lora8_fwd_kernel(Lora8Args a) {
    constexpr int D = 256 * NCH, NG = D / 8, NR = (NG + 63) / 64;      // 8-feature groups of a row, generator rounds
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const DropSpec drop = DROP ? drop_resolved(a.drop) : a.drop;
    const float kscale = DROP ? drop.keep_scale : 1.0f, scaling = a.scaling;

    // ---- resident weights of this lane's chunks (features 256 m + 4 lane .. + 3), decoded from the MFMA packs (bf16 plane):
    //   down pack (pack.hip, tests/packing_spec.py pack_down4), fragment (stage t, k-step u) of c-tile 0: slot (i, hh, j) =
    //     A[crow(i)][64 t + 16 u + 8 hh + j], i.e. 8-feature group G = 8 t + 2 u + hh; c < 8 sits at MFMA row i = c (c < 4), 8 + c - 4
    //   up pack (pack_up4), fragment (stage T, v, ks = 0): slot (i, hh = 0, j) = B[64 T + 16 v + 32 hp + 4 b + aa][j], i = 8 b + 4 hp + aa
    uint32_t wa[NCH][8][2];
    u32x4 wb[NCH][4];
    const PackGeom pg = pack_geom(1, D, 1);
    const uint8_t* down = a.pk;
    const uint8_t* up = a.pk + pg.pack_bytes;
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const int fc = 256 * m + 4 * lane, G = fc >> 3;
        const int t = G >> 3, u = (G >> 1) & 3, hh = G & 1;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int i = c < 4 ? c : 8 + (c - 4);
            const u32x2 v = *reinterpret_cast<const u32x2*>(down + (size_t)(t * 4 + u) * 1024 + (size_t)(i + 32 * hh) * 16 + (fc & 4) * 2);
            wa[m][c][0] = v[0]; wa[m][c][1] = v[1];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = fc + e, T = f >> 6, rem = f & 63;
            const int hp = (rem >> 5) & 1, v2 = (rem >> 4) & 1, q = rem & 15, i = 8 * (q >> 2) + 4 * hp + (q & 3);
            wb[m][e] = *reinterpret_cast<const u32x4*>(up + (size_t)((2 * T + v2) * 2) * 1024 + (size_t)i * 16);
        }
    }

    const uint8_t* x = reinterpret_cast<const uint8_t*>(a.x);
    const uint8_t* base = reinterpret_cast<const uint8_t*>(a.base);
    uint8_t* out = reinterpret_cast<uint8_t*>(a.out);
    const int64_t rstride = (int64_t)gridDim.x * L8_WAVES;
    u32x2 cx[NCH], cb[NCH];
    auto load_row = [&](int64_t r, u32x2 (&rx)[NCH], u32x2 (&rb)[NCH]) {
        if (r >= a.M) r = a.M - 1;                      // the next row of this wave, requested one row ahead
        const int64_t o = r * (D * 2) + lane * 8;
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            rx[m] = *reinterpret_cast<const u32x2*>(x + o + m * 512);
            rb[m] = *reinterpret_cast<const u32x2*>(base + o + m * 512);
        }
    };
    {
        const int64_t r0 = (int64_t)blockIdx.x * L8_WAVES + wave;
        if (r0 < a.M) load_row(r0, cx, cb);
    }
    for (int64_t row = (int64_t)blockIdx.x * L8_WAVES + wave; row < a.M; row += rstride) {
        u32x2 nx[NCH], nb[NCH];
        load_row(row + rstride, nx, nb);
        // ---- keep flags: group 64 rho + lane from the generator / the caller's mask, then every lane fetches the groups of its chunks
        uint32_t kb[NCH];
        if constexpr (DROP) {
            uint32_t gbits[NR];
#pragma unroll
            for (int rho = 0; rho < NR; ++rho) {
                const int g = 64 * rho + lane;
                const bool live = NG % 64 == 0 || g < NG;
                gbits[rho] = DROP == 1 ? keep8((row * D + 8 * (live ? g : 0)) >> 3, drop.seed, drop.thr) : drop_bits8(drop, row, 8 * (live ? g : 0), D);
                // (byte stores: gathering the four groups of a dword from their lanes first was measured and is no faster)
                if (live && drop.bits_out != nullptr) drop.bits_out[row * (int64_t)(D >> 3) + drop_pos(g)] = (uint8_t)gbits[rho];
                if (live && drop.keep_out != nullptr) drop_export8(drop.keep_out, row * D + 8 * g, gbits[rho]);
            }
#pragma unroll
            for (int m = 0; m < NCH; ++m) {
                const int g = 32 * m + (lane >> 1);                          // the group of chunk m: held by lane g % 64 of round g / 64
                const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((g & 63) << 2, (int)gbits[(32 * m) >> 6 < NR ? (32 * m) >> 6 : 0]);
                kb[m] = (v >> ((lane & 1) * 4)) & 0xfu;                      // this chunk's four flags
            }
        }
        // ---- down: z_c = kscale * sum_f keep_f x_f A[c][f]
        float z[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) z[c] = 0.f;
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            u32x2 xm = cx[m];
            if constexpr (DROP) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {               // flags 2q, 2q + 1 -> the two halves of a packed pair
                    const int lo = ((int)(kb[m] << (31 - 2 * q))) >> 31, hi = ((int)(kb[m] << (30 - 2 * q))) >> 31;
                    xm[q] &= __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x07060100u);
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                z[c] = dot2_bf16(xm[0], wa[m][c][0], z[c]);
                z[c] = dot2_bf16(xm[1], wa[m][c][1], z[c]);
            }
        }
        wave_sum8(z, lane);
        uint32_t zp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) zp[q] = pack_bf16x2(z[2 * q] * kscale, z[2 * q + 1] * kscale);
        // ---- what the backward keeps: z (a 64-byte row: eight values, then zeros)
        if (a.save != nullptr && lane < 4) {
            const u32x4 zv = {zp[0], zp[1], zp[2], zp[3]}, z4 = {0u, 0u, 0u, 0u};
            reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(a.save) + row * 64)[lane] = lane == 0 ? zv : z4;
        }
        // ---- up: out_f = base_f + scaling * sum_c z_c B[f][c]
        const int64_t ro = row * (D * 2) + lane * 8;
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) s = dot2_bf16(zp[q], wb[m][e][q], s);
                const uint32_t bw = cb[m][e >> 1];
                const float bv = (e & 1) ? __uint_as_float(bw & 0xffff0000u) : __uint_as_float(bw << 16);
                o[e] = bv + scaling * s;
            }
            const u32x2 ov = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            *reinterpret_cast<u32x2*>(out + ro + m * 512) = ov;
        }
#pragma unroll
        for (int m = 0; m < NCH; ++m) { cx[m] = nx[m]; cb[m] = nb[m]; }
    }
}

// ---- the backward of the same op as a streaming row kernel (round 5; VERDICT r04 "missing" #5: it ran on the 32-wide MFMA tile).
//     dz_c = scaling * sum_f dy_f B[f][c]          dx_f = keep_f / (1 - p) * sum_c dz_c A[c][f]
//     dA[c][f] = sum_m dz_c (keep_f / (1 - p) x_f)  dB[f][c] = scaling * sum_m dy_f z_c            (z = the forward's saved, bf16-rounded sums)
// Eight contractions of eight per element again -- but the weight gradients are 2 x 96 fp32 accumulators per lane (a lane owns 12 features
// at d = 768), the two weight slices 2 x 48 registers, and VALU instructions cannot accumulate into the AGPR half of the file: no wave
// can hold all of it.  So a workgroup is EIGHT waves in two roles, two waves per SIMD (<= 256 registers):
//   RB (waves 0-3): dz (48 dot2 + the reduce-scatter of the forward), dx (48 dot2 on c-pairs) -> the dx row; dB += (scaling dy) (x) z
//   WA (waves 4-7): dz again (its own copy: no hand-over between waves, no barrier per row), dA += dz (x) dropout(x)
// Wave w of each role walks the same rows (blockIdx * 4 + w, stride 4 * grid), so dy is fetched from HBM once and once more from L2.
// At the end the four waves of a role add their slices through LDS and the workgroup writes ONE partial [dA | dB] in the final layouts;
// launch_tail_reduce sums the workgroups' partials (deterministic).  Grid = one workgroup per CU (fewer for short inputs): 256 x 48 KiB
// of partials at r = 8, whatever the rows.  d <= 768 (at d = 1024 role RB needs 64 + 64 + 128 registers: the MFMA path keeps it).
// MEASURED (profiles/r05_k3_streaming_bwd_ab.txt): parity-green and slower than the two-pass MFMA form it was meant to replace -- 54.3 vs 47.8 us
// at 28,000 rows, 26.5 vs 19.3 at 2,500, the LoRA r = 8 step 19.14 vs 18.89 ms.  ~20 us do not depend on the rows (decoding the weight slices
// from the packs, the LDS sums, 12.6 MB of partials through a reduce launch), and the row loop itself -- ~300 VALU instructions per row in
// role RB, a third of them the 96 accumulator FMAs -- streams at 3.3 TB/s, no better than pass 1 + pass 2 of the matrix-core form, whose
// padded 32-wide tile costs MFMA cycles that were idle anyway.  Unlike the forward, the backward is not a contraction of eight per
// element: its weight gradients are two rank-8 OUTER products per row, and outer products are what the matrix cores are for.  Off by
// default (csrc/tuning.h lora8_bwd); kept with its test for the measurement.
struct Lora8BwdArgs {
    const void* dy; const void* x; const void* z;   // [M, d] bf16, [M, d] bf16, [M, 32] bf16 (the forward's saved sums)
    const uint8_t* pk;
    DropSpec drop;          // generator (seed) or the forward's packed mask (bits); an explicit byte mask goes to the MFMA path
    void* dx;
    float* part;            // [gridDim.x][2][r * d]
    int64_t M; int d; int r;
    float scaling;
};

template <int NCH, int DROP>      // DROP: 0 none, 1 generator, 2 packed bits
__global__ __launch_bounds__(512, 2) void lora8_bwd_kernel(Lora8BwdArgs a) {
    constexpr int D = 256 * NCH, NG = D / 8, NR = (NG + 63) / 64;
    constexpr int NV = NCH * 32;                        // accumulators per lane and role
    extern __shared__ __attribute__((aligned(16))) float l8sm[];         // [4 waves][NV][64 lanes] for the final sums
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave >> 2, wr = wave & 3;          // 0 RB, 1 WA
    const DropSpec drop = DROP ? drop_resolved(a.drop) : a.drop;
    const float kscale = DROP ? drop.keep_scale : 1.0f, scaling = a.scaling;
    const int r = a.r;
    const PackGeom pg = pack_geom(1, D, 1);
    const uint8_t* down = a.pk;
    const uint8_t* up = a.pk + pg.pack_bytes;

    // ---- resident weight slices of this lane's chunks, re-paired for the backward's contractions (decoding: see lora8_fwd_kernel):
    //   bt[m][c][j] = (B[fc + 2j][c], B[fc + 2j + 1][c])   pairs over FEATURES: dz_c += <(dy_f, dy_f+1), bt>        (both roles)
    //   ac[m][e][q] = (A[2q][fc + e], A[2q + 1][fc + e])   pairs over C:        dx_f  = sum_q <(dz_2q, dz_2q+1), ac>   (role RB)
    uint32_t bt[NCH][8][2];
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const int fc = 256 * m + 4 * lane;
        u32x4 wb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = fc + e, T = f >> 6, rem = f & 63;
            const int hp = (rem >> 5) & 1, v2 = (rem >> 4) & 1, q = rem & 15, i = 8 * (q >> 2) + 4 * hp + (q & 3);
            wb[e] = *reinterpret_cast<const u32x4*>(up + (size_t)((2 * T + v2) * 2) * 1024 + (size_t)i * 16);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t lo = wb[2 * j][c >> 1], hi = wb[2 * j + 1][c >> 1];
                bt[m][c][j] = (c & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
            }
    }

    const uint8_t* dy = reinterpret_cast<const uint8_t*>(a.dy);
    const uint8_t* x = reinterpret_cast<const uint8_t*>(a.x);
    const uint8_t* zs = reinterpret_cast<const uint8_t*>(a.z);
    uint8_t* dx = reinterpret_cast<uint8_t*>(a.dx);
    const int64_t rstride = (int64_t)gridDim.x * 4;
    const int64_t row_first = (int64_t)blockIdx.x * 4 + wr;
    auto clampr = [&](int64_t rr) { return rr < a.M ? rr : a.M - 1; };
    auto keep_flags = [&](const uint32_t (&gb)[NR], uint32_t (&kb)[NCH]) {   // this lane's chunks' flags from the groups the lanes hold
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            const int g = 32 * m + (lane >> 1);
            const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((g & 63) << 2, (int)gb[(32 * m) >> 6 < NR ? (32 * m) >> 6 : 0]);
            kb[m] = (v >> ((lane & 1) * 4)) & 0xfu;
        }
    };
    auto group_bits = [&](int64_t row, uint32_t (&gb)[NR]) {            // group 64 rho + lane (DROP == 2: unconditional byte loads; 1: the generator)
#pragma unroll
        for (int rho = 0; rho < NR; ++rho) {
            const int g = 64 * rho + lane;
            const int gl = (NG % 64 == 0 || g < NG) ? g : 0;
            if constexpr (DROP == 1) gb[rho] = keep8((row * D + 8 * gl) >> 3, drop.seed, drop.thr);
            else if constexpr (DROP == 2) gb[rho] = drop.bits[row * (int64_t)(D >> 3) + drop_pos(gl)];
            else gb[rho] = 0xffu;
        }
    };
    auto mask_pairs = [&](u32x2& v, uint32_t kbm) {                      // clears the dropped halves of two packed pairs
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int lo = ((int)(kbm << (31 - 2 * q))) >> 31, hi = ((int)(kbm << (30 - 2 * q))) >> 31;
            v[q] &= __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x07060100u);
        }
    };
    auto dz_of = [&](const u32x2 (&dyr)[NCH], float (&dz)[8]) {
#pragma unroll
        for (int c = 0; c < 8; ++c) dz[c] = 0.f;
#pragma unroll
        for (int m = 0; m < NCH; ++m)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                dz[c] = dot2_bf16(dyr[m][0], bt[m][c][0], dz[c]);
                dz[c] = dot2_bf16(dyr[m][1], bt[m][c][1], dz[c]);
            }
        wave_sum8(dz, lane);
#pragma unroll
        for (int c = 0; c < 8; ++c) dz[c] *= scaling;
    };
    float* part = a.part + (size_t)blockIdx.x * 2 * (size_t)r * D;
    auto stash = [&](const float (&acc)[NCH][4][8]) {
#pragma unroll
        for (int m = 0; m < NCH; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 8; ++c) l8sm[((size_t)wr * NV + (m * 32 + e * 8 + c)) * 64 + lane] = acc[m][e][c];
    };
    auto sum_phase = [&](int ph) {                                        // ph 0: dA [r, D]; ph 1: dB [D, r]
        for (int i = threadIdx.x; i < NV * 64; i += 512) {
            const int k = i >> 6, ln = i & 63;
            const float sv = (l8sm[(size_t)k * 64 + ln] + l8sm[((size_t)NV + k) * 64 + ln]) + (l8sm[((size_t)2 * NV + k) * 64 + ln] + l8sm[((size_t)3 * NV + k) * 64 + ln]);
            const int m = k >> 5, e = (k >> 3) & 3, c = k & 7, f = 256 * m + 4 * ln + e;
            if (c < r) part[ph == 0 ? (size_t)c * D + f : (size_t)r * D + (size_t)f * r + c] = sv;
        }
    };

    if (role == 0) {
        // ---------------------------------------------------------------- RB: dz, dx, dB[f][c] += scaling dy_f z_c
        uint32_t ac[NCH][4][4];
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            const int fc = 256 * m + 4 * lane, G = fc >> 3;
            const int t = G >> 3, u = (G >> 1) & 3, hh = G & 1;
            uint32_t wa[8][2];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int i = c < 4 ? c : 8 + (c - 4);
                const u32x2 v = *reinterpret_cast<const u32x2*>(down + (size_t)(t * 4 + u) * 1024 + (size_t)(i + 32 * hh) * 16 + (fc & 4) * 2);
                wa[c][0] = v[0]; wa[c][1] = v[1];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t lo = wa[2 * q][e >> 1], hi = wa[2 * q + 1][e >> 1];
                    ac[m][e][q] = (e & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
                }
        }
        float acc[NCH][4][8];
#pragma unroll
        for (int m = 0; m < NCH; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[m][e][c] = 0.f;
        u32x2 cd[NCH], nd[NCH];
        u32x4 cz, nz;
        uint32_t cg[NR], ng[NR];
        auto load_row = [&](int64_t rr, u32x2 (&rd)[NCH], u32x4& rz, uint32_t (&gb)[NR]) {
            rr = clampr(rr);
            const int64_t o = rr * (D * 2) + lane * 8;
#pragma unroll
            for (int m = 0; m < NCH; ++m) rd[m] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(dy + o + m * 512));
            rz = *reinterpret_cast<const u32x4*>(zs + rr * 64);           // (the same 16 bytes for every lane: one request)
            group_bits(rr, gb);
        };
        load_row(row_first, cd, cz, cg);
        for (int64_t row = row_first; row < a.M; row += rstride) {
            load_row(row + rstride, nd, nz, ng);
            float dz[8];
            dz_of(cd, dz);
            uint32_t zp[4], kb[NCH];
#pragma unroll
            for (int q = 0; q < 4; ++q) zp[q] = pack_bf16x2(dz[2 * q], dz[2 * q + 1]);
            if constexpr (DROP != 0) keep_flags(cg, kb);
            const int64_t ro = row * (D * 2) + lane * 8;
#pragma unroll
            for (int m = 0; m < NCH; ++m) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float sv = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) sv = dot2_bf16(zp[q], ac[m][e][q], sv);
                    o[e] = sv * kscale;
                }
                u32x2 ov = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                if constexpr (DROP != 0) mask_pairs(ov, kb[m]);
                *reinterpret_cast<u32x2*>(dx + ro + m * 512) = ov;
            }
            float zc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) zc[c] = (c & 1) ? __uint_as_float(cz[c >> 1] & 0xffff0000u) : __uint_as_float(cz[c >> 1] << 16);
#pragma unroll
            for (int m = 0; m < NCH; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t w = cd[m][e >> 1];
                    const float dv = scaling * ((e & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16));
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[m][e][c] = __builtin_fmaf(dv, zc[c], acc[m][e][c]);
                }
#pragma unroll
            for (int m = 0; m < NCH; ++m) cd[m] = nd[m];
            cz = nz;
#pragma unroll
            for (int q = 0; q < NR; ++q) cg[q] = ng[q];
        }
        // (the four waves of WA, then those of RB, add their slices through LDS; every wave passes the same three barriers)
        __syncthreads();                                                  // S1: WA has stashed dA
        sum_phase(0);
        __syncthreads();                                                  // S2: the dA sums are read
        stash(acc);
        __syncthreads();                                                  // S3
        sum_phase(1);
    } else {
        // ---------------------------------------------------------------- WA: dz (own copy), dA[c][f] += dz_c (keep_f / (1 - p) x_f)
        float acc[NCH][4][8];
#pragma unroll
        for (int m = 0; m < NCH; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[m][e][c] = 0.f;
        u32x2 cd[NCH], nd[NCH], cx[NCH], nx[NCH];
        uint32_t cg[NR], ng[NR];
        auto load_row = [&](int64_t rr, u32x2 (&rd)[NCH], u32x2 (&rx)[NCH], uint32_t (&gb)[NR]) {
            rr = clampr(rr);
            const int64_t o = rr * (D * 2) + lane * 8;
#pragma unroll
            for (int m = 0; m < NCH; ++m) {
                rd[m] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(dy + o + m * 512));
                rx[m] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(x + o + m * 512));
            }
            group_bits(rr, gb);
        };
        load_row(row_first, cd, cx, cg);
        for (int64_t row = row_first; row < a.M; row += rstride) {
            load_row(row + rstride, nd, nx, ng);
            float dz[8];
            dz_of(cd, dz);
            uint32_t kb[NCH];
            if constexpr (DROP != 0) keep_flags(cg, kb);
#pragma unroll
            for (int m = 0; m < NCH; ++m) {
                u32x2 xm = cx[m];
                if constexpr (DROP != 0) mask_pairs(xm, kb[m]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t w = xm[e >> 1];
                    const float xv = kscale * ((e & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16));
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[m][e][c] = __builtin_fmaf(dz[c], xv, acc[m][e][c]);
                }
            }
#pragma unroll
            for (int m = 0; m < NCH; ++m) { cd[m] = nd[m]; cx[m] = nx[m]; }
#pragma unroll
            for (int q = 0; q < NR; ++q) cg[q] = ng[q];
        }
        stash(acc);
        __syncthreads();                                                  // S1
        sum_phase(0);
        __syncthreads();                                                  // S2
        __syncthreads();                                                  // S3: RB has stashed dB
        sum_phase(1);
    }
}

// workgroups of the streaming backward: one per CU, fewer when that would leave a wave without a row
int lora8_bwd_blocks(int64_t M) { const int64_t b = (M + 3) / 4; return (int)(b < 256 ? b : 256); }
size_t lora8_bwd_part_bytes(int64_t M, int d, int r) { return (size_t)lora8_bwd_blocks(M) * 2 * (size_t)r * d * sizeof(float); }
bool lora8_bwd_applies(int64_t M, int d, int r, int io_fp32, const DropSpec& drop) {
    (void)drop;                                         // (any mask source: the backward reads the packed mask the forward saved)
    return !io_fp32 && M > 0 && r >= 1 && r <= 8 && d % 256 == 0 && d <= 768;
}

template <int NCH>
static hipError_t launch_lora8_bwd_n(const Lora8BwdArgs& a, hipStream_t stream) {
    const int blocks = lora8_bwd_blocks(a.M);
    const size_t lds = (size_t)4 * NCH * 32 * 64 * sizeof(float);
    const int mode = !drop_active(a.drop) ? 0 : (a.drop.bits != nullptr ? 2 : 1);
    const void* kern = mode == 0 ? reinterpret_cast<const void*>(lora8_bwd_kernel<NCH, 0>)
                     : mode == 1 ? reinterpret_cast<const void*>(lora8_bwd_kernel<NCH, 1>) : reinterpret_cast<const void*>(lora8_bwd_kernel<NCH, 2>);
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (mode == 0) hipLaunchKernelGGL((lora8_bwd_kernel<NCH, 0>), dim3(blocks), dim3(512), lds, stream, a);
    else if (mode == 1) hipLaunchKernelGGL((lora8_bwd_kernel<NCH, 1>), dim3(blocks), dim3(512), lds, stream, a);
    else hipLaunchKernelGGL((lora8_bwd_kernel<NCH, 2>), dim3(blocks), dim3(512), lds, stream, a);
    return hipGetLastError();
}

// dx, dA [r, d], dB [d, r] (fp32, overwritten) of the K3 backward at r <= 8; part = lora8_bwd_part_bytes(M, d, r) of workspace
hipError_t launch_lora8_bwd(const void* dy, const void* x, const void* z, const uint8_t* pk, const DropSpec& drop, void* dx, float* da, float* db,
                            float* part, int64_t M, int d, int r, float scaling, hipStream_t stream) {
    Lora8BwdArgs a{};
    a.dy = dy; a.x = x; a.z = z; a.pk = pk; a.drop = drop; a.drop.keep_out = nullptr; a.drop.bits_out = nullptr;
    a.dx = dx; a.part = part; a.M = M; a.d = d; a.r = r; a.scaling = scaling;
    hipError_t e;
    switch (d / 256) {
        case 1: e = launch_lora8_bwd_n<1>(a, stream); break;
        case 2: e = launch_lora8_bwd_n<2>(a, stream); break;
        case 3: e = launch_lora8_bwd_n<3>(a, stream); break;
        default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    return launch_tail_reduce(part, lora8_bwd_blocks(M), r * d, da, db, stream);
}

bool lora8_applies(int64_t M, int d, int r, int io_fp32) {
    return !io_fp32 && M > 0 && r >= 1 && r <= 8 && d % 256 == 0 && d <= 1024;
}

template <int NCH>
static hipError_t launch_lora8_n(const Lora8Args& a, hipStream_t stream) {
    const int blocks = tail_blocks(a.M);
#ifndef VLPET_LORA8_GENERAL      // (-DVLPET_LORA8_GENERAL=1: every dropout launch on the any-source form, the round-4 behaviour, for A/B)
#define VLPET_LORA8_GENERAL 0
#endif
    if (!VLPET_LORA8_GENERAL && drop_active(a.drop) && a.drop.keep == nullptr && a.drop.bits == nullptr)
        hipLaunchKernelGGL((lora8_fwd_kernel<NCH, 1>), dim3(blocks), dim3(L8_WAVES * 64), 0, stream, a);
    else if (drop_active(a.drop)) hipLaunchKernelGGL((lora8_fwd_kernel<NCH, 2>), dim3(blocks), dim3(L8_WAVES * 64), 0, stream, a);
    else hipLaunchKernelGGL((lora8_fwd_kernel<NCH, 0>), dim3(blocks), dim3(L8_WAVES * 64), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_lora8_fwd(const Lora8Args& a, hipStream_t stream) {
    switch (a.d / 256) {
        case 1: return launch_lora8_n<1>(a, stream);
        case 2: return launch_lora8_n<2>(a, stream);
        case 3: return launch_lora8_n<3>(a, stream);
        case 4: return launch_lora8_n<4>(a, stream);
        default: return hipErrorInvalidValue;
    }
}
