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

// (Round 5 also built the BACKWARD at r <= 8 as a streaming row kernel: parity-green and slower than the two-pass MFMA form -- 54.3 vs 47.8 us at
//  28,000 rows, 26.5 vs 19.3 at 2,500, profiles/r05_k3_streaming_bwd_ab.txt: its weight gradients are rank-8 OUTER products per row, which is what
//  the matrix cores are for.  Removed in round 6.)

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
