// K1 backward, column-parallel pass 2 for SIX bottleneck tiles (r = r_g = 192), second form (round 5): the up-side weight gradients run ONE
// STEP LATE and the up-side elementwise block starts from the forward's output.  Autograd of my_transformers/modeling_t5.py:366-390, 782-806
// (the T5 script, scripts/image-text/T5-VL-PET-large.sh:41-59).  Read pet_cols6.hip first: same decomposition (four roles UE / UW / DE / DW
// per column quarter, 64-column workgroups, 32-row steps), same partial-sum layout, same XCD placement.
//
// What bounded pet_cols6.hip: a step was a CHAIN.  UW contracts this step's dh / dq (written by UE) with this step's z tiles, so a step had two
// barriers -- stage hand-over, then the dh / dq hand-over -- and every other wave stood at the next stage barrier while UW ran its 24
// MFMAs behind UE's projections and elementwise block: 3.8 us per 32 rows, 107-113 us at 18,250 rows for a pass whose MFMA pipes are
// 17 % busy.  Here UW works on the PREVIOUS step's dh / dq / z while UE produces this step's: one barrier per step, the two halves of the
// chain overlap.  The price is a third slot for the z tiles (they are read by UE in step s and by UW in step s + 1) -- 24 KiB that the
// 160 KiB of LDS do not have next to two full stages -- and it is paid by taking out of LDS what only ONE wave reads:
//   * dy and -- new -- y: a UE lane needs exactly the 16 contiguous columns of its row (32 bytes) of each; it loads them itself, one step
//     ahead, into registers.  With y = gs * h * g the elementwise block is dh = gs * dy * g, dq = dy * y * (1 - g) (pet_dz2.hip's note):
//     no adapter-chain projection in UE at all (12 MFMAs, 12 fragment reads, 48 weight registers less), no x2 read by UE;
//   * dx1_in: the same 32 bytes per DE lane.
// LDS: z ring 3 x [z_a | z_g] 72 KiB, dpre ring 2 x [dp_a | dp_g] 48 KiB, row ring 2 x [x2 | x1] 16 KiB (only the W waves' transpose reads
// are left on it), dh 2 x 4 KiB, dq 2 x 4 KiB, biases: 152.5 KiB.
// Applies when the caller passes y (vlpet_adapter_gate_bwd_saved_y) or the gate is additive (dq = gs * dy * g (1 - g): neither h nor y).
#include "cols_common.h"

// timing ablations (results wrong on purpose; tools/gpu/r5_u.sh): 1 = no wait for the stage pieces, 2 = no role work between the barriers,
// 4 = no dx1 / dx2 stores
#ifndef VLPET_C6Y_ABL
#define VLPET_C6Y_ABL 0
#endif
template <int RT> struct Colz6yGeo {
    static constexpr int KT = 2 * RT;
    static constexpr int PB = 64 * RT, NPR = PB / 16;   // bytes / 16-byte slots of a bottleneck row
    static constexpr int PT_B = 32 * PB;
    static constexpr int NZ = 3, ND = 2, NXS = 2;
    static constexpr int Z_OFF = 0, DP_OFF = Z_OFF + NZ * 2 * PT_B, XR_OFF = DP_OFF + ND * 2 * PT_B;
    static constexpr int XR_B = 2 * 4096;               // [x2 | x1] pair tiles of a stage
    static constexpr int DH_OFF = XR_OFF + NXS * XR_B, DQ_OFF = DH_OFF + 2 * 4096, BIAS_OFF = DQ_OFF + 2 * 4096;
    static constexpr size_t lds() { return (size_t)BIAS_OFF + 2 * 64 * 4; }
};

template <int RT, bool ADD, bool HAS_IN>
__global__ __launch_bounds__(512, 2) void k1_cols6y_kernel(ColzArgs a) {
    using GEO = Colz6yGeo<RT>;
    static_assert(RT % 2 == 0, "even tile counts (the bottleneck rows take the row tiles' swizzle)");
    constexpr int KT = GEO::KT, PB = GEO::PB, NPR = GEO::NPR, PT_B = GEO::PT_B;
    constexpr int Z_OFF = GEO::Z_OFF, DP_OFF = GEO::DP_OFF, XR_OFF = GEO::XR_OFF, XR_B = GEO::XR_B;
    constexpr int DH_OFF = GEO::DH_OFF, DQ_OFF = GEO::DQ_OFF, BIAS_OFF = GEO::BIAS_OFF;
    constexpr int PR = 32 * RT;
    constexpr int NPW = 8 * RT / 4;                     // bottleneck pieces (1 KiB) per E wave and stage: three per tensor
    static_assert(NPW == 12, "piece j of an E wave belongs to tensor j / 3");
    constexpr int GRP = 6;                              // B fragments per batch of a projection
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    // ---- which (column block, row chunk): as in pet_cols6.hip
    const int d = a.d, NCB = d >> 6, CBH = NCB >> 1;
    int grp, mem;
    cols_decode((int)blockIdx.x, CBH, grp, mem);
    if (grp >= 2 * a.row_chunks) return;
    const int rc = grp >> 1, cb = (grp & 1) * CBH + mem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave & 1, side = (wave >> 1) & 1, kind = wave >> 2;   // column quarter, U / D, E / W  (waves w, w + 4 share a SIMD)
    const int m = lane & 31, h = lane >> 5;
    const int64_t ld2 = (int64_t)d * 2;
    const int c0 = 64 * cb + 32 * nt;
    const int64_t r_begin = (int64_t)rc * a.rows_per_chunk;
    int64_t r_end = r_begin + a.rows_per_chunk;
    if (r_end > a.M) r_end = a.M;
    const int nsteps = r_end > r_begin ? (int)((r_end - r_begin + 31) >> 5) : 0;

    const PackGeom pg = pack_geom(RT, d, 1);
    if (tid < 128) {                                    // up-side biases of the workgroup's 64 columns -> LDS (fp32); UE uses the gate's
        float* sbias = reinterpret_cast<float*>(smem + BIAS_OFF);
        const uint8_t* pk = tid < 64 ? a.pk_a : a.pk_g;
        sbias[tid] = reinterpret_cast<const float*>(pk + pg.bias_off)[PR + 64 * cb + (tid & 63)];
    }

    // ---- stage pieces (1 KiB each).  E waves: the 48 pieces of the four bottleneck tiles (E wave i takes pieces i + 4 j; piece q belongs
    // to tensor q / 12 = j / 3); W waves: rows 8 i .. of the x2 and x1 pair tiles.  All swizzles on the source side.
    const uint8_t* pbase[NPW]; uint32_t pdst[NPW], poff[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int q = (wave & 3) + 4 * j, t = q / KT, piece = q % KT;
        const int sig = piece * 64 + lane, pr = sig / NPR;
        pbase[j] = reinterpret_cast<const uint8_t*>(t == 0 ? a.z_a : t == 1 ? a.z_g : t == 2 ? a.dp_a : a.dp_g);
        poff[j] = (uint32_t)(pr * PB + ((sig % NPR) ^ fsw(pr)) * 16);
        pdst[j] = (uint32_t)((t & 1) * PT_B + piece * 1024);              // inside its ring slot: [z_a | z_g] or [dp_a | dp_g]
    }
    const int xrow = 8 * (wave & 3) + (lane >> 3);
    const uint32_t xoff = (uint32_t)xrow * (uint32_t)ld2 + (uint32_t)(64 * cb * 2 + (((lane & 7) ^ fsw(xrow)) * 16));
    const uint8_t* xbase[2] = {reinterpret_cast<const uint8_t*>(a.x2), reinterpret_cast<const uint8_t*>(a.x1)};
    auto sbase = [](const uint8_t* p) {                 // a wave-uniform pointer as a fresh scalar (see pet_cols.hip)
        const uint64_t u = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const uint8_t*>(((uint64_t)hi << 32) | lo);
    };
    auto issue_p = [&](int s) {                         // (E waves)
        const int64_t rb = r_begin + 32 * (int64_t)s;
        uint8_t* zs = smem + Z_OFF + (size_t)(s % GEO::NZ) * (2 * PT_B);
        uint8_t* ds = smem + DP_OFF + (size_t)(s % GEO::ND) * (2 * PT_B);
        const int last = (int)(r_end - rb) - 1;         // (>= 31 except in the last step: rows past the end re-read the last row)
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int pr = (int)(poff[j] / PB);
            glds16(sbase(pbase[j] + rb * PB) + poff[j] - (uint32_t)(pr > last ? pr - last : 0) * PB, (j < NPW / 2 ? zs : ds) + pdst[j]);
        }
    };
    auto issue_x = [&](int s) {                         // (W waves)
        const int64_t rb = r_begin + 32 * (int64_t)s;
        uint8_t* st = smem + XR_OFF + (size_t)(s % GEO::NXS) * XR_B;
        const int last = (int)(r_end - rb) - 1;
        const uint32_t xo = xoff - (uint32_t)(xrow > last ? xrow - last : 0) * (uint32_t)ld2;
#pragma unroll
        for (int t = 0; t < 2; ++t) glds16_row(sbase(xbase[t] + rb * ld2) + xo, st + t * 4096 + (wave & 3) * 1024);
    };
    auto issue = [&](int s) { if (kind) issue_x(s); else issue_p(s); };
    // the 32 bytes (columns c0 + 16 h .. + 15) of row m of step s of a row tensor that only this lane needs: straight into registers
    auto lane_row = [&](const void* base, int s, u32x4& lo, u32x4& hi) {
        const int64_t rb = r_begin + 32 * (int64_t)s;
        const int last = (int)(r_end - rb) - 1;
        const uint8_t* p = sbase(reinterpret_cast<const uint8_t*>(base) + rb * ld2) + (uint32_t)(m > last ? last : m) * (uint32_t)ld2 + (uint32_t)((c0 + 16 * h) * 2);
        lo = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        hi = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p) + 1);
    };

    // ---- per-lane LDS byte addresses (bottleneck tiles: relative to their ring slot; row tiles: relative to the tile)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    uint32_t a_xtr[2], a_ptr[2][2], a_xcl[2], a_pbf[4];
    {
        const int g4 = lane >> 4, sl = lane & 15;
        const int trow = 8 * (g4 >> 1) + (sl >> 2);                       // first row of this lane's transpose reads (second: + 4)
        const int tslot = 2 * (g4 & 1) + ((sl & 3) >> 1), thalf = 8 * (sl & 1);
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int r = trow + 4 * hi;
            a_xtr[hi] = (uint32_t)(r * 128 + (((4 * nt + tslot) ^ fsw(r)) * 16) + thalf);
#pragma unroll
            for (int par = 0; par < 2; ++par)                             // slot 4 ct + tslot of a bottleneck row: ct odd / even (+ 128 (ct >> 1))
                a_ptr[hi][par] = (uint32_t)(r * PB + (((4 * par + tslot) ^ fsw(r)) * 16) + thalf);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) a_xcl[k] = (uint32_t)(m * 128 + (((4 * nt + 2 * h + k) ^ fsw(m)) * 16));   // columns 8k .. of the lane's 16
#pragma unroll
        for (int k = 0; k < 4; ++k) a_pbf[k] = (uint32_t)(m * PB + (((2 * k + h) ^ fsw(m)) * 16));             // B fragment, k-step 4j + k (+ 128 j)
    }
    auto ones_row = [&](int k) {
        int mm = m;
        asm volatile("" : "+v"(mm));
        const uint32_t w = (mm == (k & 3) + 8 * (k >> 2)) ? 0x3f803f80u : 0u;
        const u32x4 v = {w, w, w, w};
        return __builtin_bit_cast(bf16x8, v);
    };
    const int RC = a.row_chunks;
    const int col = c0 + m;
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    auto zslot = [&](int s) { return lds0 + (uint32_t)(Z_OFF + (s % GEO::NZ) * (2 * PT_B)); };
    auto dslot = [&](int s) { return lds0 + (uint32_t)(DP_OFF + (s % GEO::ND) * (2 * PT_B)); };
    auto xslot = [&](int s) { return lds0 + (uint32_t)(XR_OFF + (s % GEO::NXS) * XR_B); };

    // the ONE barrier of a step: wait for the own pieces of stage s, barrier (stage s is complete; whatever step s - 1 wrote or read is
    // done), request stage s + 1.  Ring reuse: z slot (s + 1) % 3 was last read by UW in step s - 1, the dpre / row slots (s + 1) % 2 by
    // DE / DW in step s - 1, dh / dq slot s & 1 (written by UE in step s) by DE / UW in step s - 1.
    auto step_top = [&](int s, int extra) {
        if (!(VLPET_C6Y_ABL & 1)) vm_wait(extra);
        __builtin_amdgcn_s_barrier();
        if (s + 1 < nsteps) issue(s + 1);
        const int valid = (int)(r_end - (r_begin + 32 * (int64_t)s));
        if (valid < 32) {                               // zero the bottleneck rows past the end (their products must vanish)
            const u32x4 z = {0u, 0u, 0u, 0u};
            uint8_t* zt = smem + Z_OFF + (size_t)(s % GEO::NZ) * (2 * PT_B);
            uint8_t* dt = smem + DP_OFF + (size_t)(s % GEO::ND) * (2 * PT_B);
            for (int q = tid; q < 2 * 32 * NPR; q += 512) {
                const int rr = (q / NPR) & 31;
                if (rr >= valid) { *reinterpret_cast<u32x4*>(zt + (size_t)q * 16) = z; *reinterpret_cast<u32x4*>(dt + (size_t)q * 16) = z; }
            }
            __syncthreads();
        }
    };
    auto store_partials = [&](int ja, int jg, const f32x16* accA, const f32x16* accG) {
        float* tA = a.part[ja] + (int64_t)rc * PR * d;
        float* tG = a.part[jg] + (int64_t)rc * PR * d;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int crow = 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h;
                tA[(int64_t)crow * d + col] = accA[ct][i];
                tG[(int64_t)crow * d + col] = accG[ct][i];
            }
    };

    if (kind == 0) {
        // ================================================================ the E roles: resident weights, no accumulators
        bf16x8 wA[KT], wG[KT];                          // UE: Wgu only (wA unused); DE: Wd^T, Wgd^T
        {
            const int i = m, v = (i >> 2) & 1, ip = (i & 3) | (nt << 2) | ((i >> 3) << 3);
            const int64_t off = (int64_t)(side == 0 ? 1 : 3) * pg.pack_bytes + (int64_t)cb * (4 * RT * 1024)
                              + (int64_t)(v * KT) * 1024 + (ip + 32 * h) * 16;
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
                if (side != 0) wA[ks] = *reinterpret_cast<const bf16x8*>(a.pk_a + off + ks * 1024);
                wG[ks] = *reinterpret_cast<const bf16x8*>(a.pk_g + off + ks * 1024);
            }
        }
        // acc (+)= W . B fragments of tile T of the ring slot at `sb` (lane = row); BIAS >= 0: the accumulator starts at the up-side bias there
        auto project = [&](uint32_t sb, auto TC, auto BIASC, const bf16x8* w, f32x16& acc) {
            constexpr int T = decltype(TC)::value, BIAS = decltype(BIASC)::value;
            sfor<KT / GRP>([&](auto G) {
                u32x4 bf[GRP], bb[4];
                if constexpr (BIAS >= 0 && G.value == 0) {
                    const uint32_t a_bias = lds0 + (uint32_t)(BIAS_OFF + BIAS + (32 * nt + 16 * h) * 4);
                    sfor<4>([&](auto Q) { lds_read16<16 * Q.value>(bb[Q.value], a_bias); });
                }
                sfor<GRP>([&](auto K) {
                    constexpr int ks = G.value * GRP + K.value;
                    lds_read16<T * PT_B + 128 * (ks >> 2)>(bf[K.value], sb + a_pbf[ks & 3]);
                });
                lgkm_fence(bf[0]);
                if constexpr (BIAS >= 0 && G.value == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        lgkm_tie(bb[q]);
#pragma unroll
                        for (int w2 = 0; w2 < 4; ++w2) acc[4 * q + w2] = __uint_as_float(bb[q][w2]);
                    }
                }
#pragma unroll
                for (int k = 0; k < GRP; ++k) { if (k) lgkm_tie(bf[k]); acc = mfma32(w[G.value * GRP + k], as_bf(bf[k]), acc); }
            });
        };
        if (nsteps > 0) issue(0);                       // (first requests before the resident operands are waited for: pet_dz2.hip's note)
        if (side == 0) {
            // ---------------------------------------------------------------- UE: the gate's up projection, dh / dq from dy and y
            u32x4 cdy[2], cy[2], ndy[2], ny[2];
            ndy[0] = ndy[1] = ny[0] = ny[1] = u32x4{0u, 0u, 0u, 0u};
            if (nsteps > 0) {
                lane_row(a.dy, 0, cdy[0], cdy[1]);
                if constexpr (!ADD) lane_row(a.y, 0, cy[0], cy[1]);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // weights in registers, biases in LDS
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) asm volatile("" : "+v"(wG[ks]));
#pragma unroll 1
            for (int s = 0; s < nsteps; ++s) {
                const int valid = (int)(r_end - (r_begin + 32 * (int64_t)s));
                const uint32_t dh0 = lds0 + (uint32_t)(DH_OFF + (s & 1) * 4096), dq0 = lds0 + (uint32_t)(DQ_OFF + (s & 1) * 4096);
                step_top(s, 0);
                if (s + 1 < nsteps) {                   // this lane's dy / y of the next step
                    lane_row(a.dy, s + 1, ndy[0], ndy[1]);
                    if constexpr (!ADD) lane_row(a.y, s + 1, ny[0], ny[1]);
                }
                if (VLPET_C6Y_ABL & 2) continue;
                f32x16 aG;
                project(zslot(s), I1{}, std::integral_constant<int, 256>{}, wG, aG);
                const float live = m < valid ? 1.f : 0.f, gsr = live * a.gs;
                sfor<4>([&](auto C) {
                    constexpr int c = C.value;
                    asm volatile("" : "+v"(aG[4 * c]), "+v"(aG[4 * c + 1]), "+v"(aG[4 * c + 2]), "+v"(aG[4 * c + 3]) :: "memory");
                    float dh[4], dq[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * c + j;
                        const float gt = sigm(aG[e]);
                        const float dyr = bf_at(cdy[e >> 3], e & 7);
                        if constexpr (ADD) {
                            dh[j] = gsr * dyr;
                            dq[j] = dh[j] * gt * (1.0f - gt);
                        } else {
                            dh[j] = gsr * dyr * gt;
                            dq[j] = live * dyr * bf_at(cy[e >> 3], e & 7) * (1.0f - gt);
                        }
                    }
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                    const bf16x4 th = {(__bf16)dh[0], (__bf16)dh[1], (__bf16)dh[2], (__bf16)dh[3]};
                    const bf16x4 tq = {(__bf16)dq[0], (__bf16)dq[1], (__bf16)dq[2], (__bf16)dq[3]};
                    lds_write8<8 * (c & 1)>(dh0 + a_xcl[c >> 1], __builtin_bit_cast(u32x2, th));
                    lds_write8<8 * (c & 1)>(dq0 + a_xcl[c >> 1], __builtin_bit_cast(u32x2, tq));
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (every LDS access of this step is complete at the next barrier)
                cdy[0] = ndy[0]; cdy[1] = ndy[1];
                if constexpr (!ADD) { cy[0] = ny[0]; cy[1] = ny[1]; }
            }
            __builtin_amdgcn_s_barrier();                                 // the last dh / dq tiles are visible to DE / UW
        } else {
            // ---------------------------------------------------------------- DE: input gradients, one step late
            const float s2 = a.s2;
            f32x16 p2 = zero16(), p1 = zero16();
            u32x4 dinA = {0u, 0u, 0u, 0u}, dinB = {0u, 0u, 0u, 0u};
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) { asm volatile("" : "+v"(wA[ks])); asm volatile("" : "+v"(wG[ks])); }
            auto finish = [&](int sp) {
                const int64_t rb = r_begin + 32 * (int64_t)sp;
                const int valid = (int)(r_end - rb) < 32 ? (int)(r_end - rb) : 32;
                const bool row_ok = m < valid;
                const uint32_t rowoff = (uint32_t)(row_ok ? m : valid - 1) * (uint32_t)ld2 + (uint32_t)((c0 + 16 * h) * 2);
                const uint32_t dh0 = lds0 + (uint32_t)(DH_OFF + (sp & 1) * 4096);
                u32x4 dhv0, dhv1;
                lds_read16<0>(dhv0, dh0 + a_xcl[0]); lds_read16<0>(dhv1, dh0 + a_xcl[1]);
                lgkm_fence(dhv0); lgkm_tie(dhv1);
                float o[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] = s2 * bf_at(e < 8 ? dhv0 : dhv1, e & 7) + p2[e];
                const u32x4 v0 = pack8(o), v1 = pack8(o + 8);
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] = p1[e] + (HAS_IN ? bf_at(e < 8 ? dinA : dinB, e & 7) : 0.f);
                const u32x4 u0 = pack8(o), u1 = pack8(o + 8);
                if (row_ok) {
                    uint8_t* q2 = const_cast<uint8_t*>(sbase(reinterpret_cast<const uint8_t*>(a.dx2) + rb * ld2)) + rowoff;
                    uint8_t* q1 = const_cast<uint8_t*>(sbase(reinterpret_cast<const uint8_t*>(a.dx1) + rb * ld2)) + rowoff;
                    reinterpret_cast<u32x4*>(q2)[0] = v0; reinterpret_cast<u32x4*>(q2)[1] = v1;
                    reinterpret_cast<u32x4*>(q1)[0] = u0; reinterpret_cast<u32x4*>(q1)[1] = u1;
                }
            };
#pragma unroll 1
            for (int s = 0; s < nsteps; ++s) {
                step_top(s, (s >= 2 ? 4 : 0) + (HAS_IN && s >= 1 ? 2 : 0));  // (younger than stage s: the four output stores of step s - 2, the two dx1_in loads of step s - 1)
                if (VLPET_C6Y_ABL & 2) continue;
                if (s > 0 && !(VLPET_C6Y_ABL & 4)) finish(s - 1);         // (uses p2 / p1 / din of step s - 1)
                if constexpr (HAS_IN) lane_row(a.dxin, s, dinA, dinB);  // this step's incoming rows: used by finish(s) in step s + 1 (one buffer: finish(s - 1) is done)
                p2 = zero16(); p1 = zero16();
                project(dslot(s), I0{}, std::integral_constant<int, -1>{}, wA, p2);
                project(dslot(s), I1{}, std::integral_constant<int, -1>{}, wG, p1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                                 // UE has written the last dh tile
            if (nsteps > 0) finish(nsteps - 1);
        }
    } else {
        // ================================================================ the W roles: accumulators, no weights
        f32x16 accA[RT], accG[RT];
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) { accA[ct] = zero16(); accG[ct] = zero16(); }
        // acc[ct] += P^T (tile TP of the ring slot at `sb`) . X (row tile at xlo / xhi); one 16-row k-step and three column tiles at a time
        // (208 of the 256 registers are accumulators here); SLOT >= 0: the column sums of X go to that slot of sx
        auto wg_products = [&](uint32_t sb, auto TPC, auto SLOTC, uint32_t xlo, uint32_t xhi, f32x16* acc, f32x16& sx) {
            constexpr int TP = decltype(TPC)::value, SLOT = decltype(SLOTC)::value;
            sfor<2>([&](auto KS) {
                constexpr int ks = KS.value;
                TrOp bx;
                tr_read2<ks * 16 * 128>(bx, xlo, xhi);
                sfor<RT / 3>([&](auto CG) {
                    TrOp ap[3];
                    sfor<3>([&](auto C3) {
                        constexpr int ct = CG.value * 3 + C3.value;
                        tr_read2<TP * PT_B + 128 * (ct >> 1) + ks * 16 * PB>(ap[C3.value], sb + a_ptr[0][ct & 1], sb + a_ptr[1][ct & 1]);
                    });
                    tr_fence(ap[0]);
                    if constexpr (CG.value == 0) tr_tie(bx);
                    const bf16x8 vx = tr_val(bx);
                    if constexpr (SLOT >= 0 && CG.value == 0) sx = mfma32(ones_row(SLOT), vx, sx);
#pragma unroll
                    for (int c3 = 0; c3 < 3; ++c3) {
                        if (c3) tr_tie(ap[c3]);
                        acc[CG.value * 3 + c3] = mfma32(tr_val(ap[c3]), vx, acc[CG.value * 3 + c3]);
                    }
                });
            });
        };
        using NOSLOT = std::integral_constant<int, -1>;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (nsteps > 0) issue(0);
        if (side == 0) {
            // ---------------------------------------------------------------- UW: dWu, dWgu (stage s - 1 in step s), every bias sum
            const bool want_csp = cb == 0 && nt == 0;
            f32x16 sx = zero16();
            auto late = [&](int sp) {                                     // dh / dq of step sp (UE wrote them in step sp) x z of stage sp
                const uint32_t dh0 = lds0 + (uint32_t)(DH_OFF + (sp & 1) * 4096), dq0 = lds0 + (uint32_t)(DQ_OFF + (sp & 1) * 4096);
                wg_products(zslot(sp), I0{}, I0{}, dh0 + a_xtr[0], dh0 + a_xtr[1], accA, sx);
                wg_products(zslot(sp), I1{}, I1{}, dq0 + a_xtr[0], dq0 + a_xtr[1], accG, sx);
            };
#pragma unroll 1
            for (int s = 0; s < nsteps; ++s) {
                step_top(s, 0);
                if (VLPET_C6Y_ABL & 2) continue;
                if (s > 0) late(s - 1);
                if (want_csp) {                                           // column sums of dpre_a, dpre_g of THIS stage: one wave per row chunk
                    const uint32_t sb = dslot(s);
                    sfor<2>([&](auto KS) {
                        constexpr int ks = KS.value;
                        sfor<2>([&](auto TT) {
                            sfor<RT / 3>([&](auto CG) {
                                TrOp ap[3];
                                sfor<3>([&](auto C3) {
                                    constexpr int ct = CG.value * 3 + C3.value;
                                    tr_read2<TT.value * PT_B + 128 * (ct >> 1) + ks * 16 * PB>(ap[C3.value], sb + a_ptr[0][ct & 1], sb + a_ptr[1][ct & 1]);
                                });
                                tr_fence(ap[0]);
#pragma unroll
                                for (int c3 = 0; c3 < 3; ++c3) {
                                    if (c3) tr_tie(ap[c3]);
                                    sx = mfma32(ones_row(2 + TT.value * RT + CG.value * 3 + c3), tr_val(ap[c3]), sx);
                                }
                            });
                        });
                    });
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                                 // UE has written the last dh / dq tiles
            if (nsteps > 0) late(nsteps - 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            store_partials(1, 3, accA, accG);
            if (h == 0) {
                a.part[1][(int64_t)RC * PR * d + (int64_t)rc * d + col] = sx[0];
                a.part[3][(int64_t)RC * PR * d + (int64_t)rc * d + col] = sx[1];
                if (want_csp) {
                    float* psa = a.part[0] + (int64_t)RC * PR * d + (int64_t)RC * d + (int64_t)rc * PR;
                    float* psg = a.part[2] + (int64_t)RC * PR * d + (int64_t)RC * d + (int64_t)rc * PR;
#pragma unroll
                    for (int ct = 0; ct < RT; ++ct) { psa[32 * ct + m] = sx[2 + ct]; psg[32 * ct + m] = sx[2 + RT + ct]; }
                }
            }
        } else {
            // ---------------------------------------------------------------- DW: dWd, dWgd
#pragma unroll 1
            for (int s = 0; s < nsteps; ++s) {
                const uint32_t xs = xslot(s);
                step_top(s, 0);
                if (VLPET_C6Y_ABL & 2) continue;
                wg_products(dslot(s), I0{}, NOSLOT{}, xs + a_xtr[0], xs + a_xtr[1], accA, accA[0]);
                wg_products(dslot(s), I1{}, NOSLOT{}, xs + 4096 + a_xtr[0], xs + 4096 + a_xtr[1], accG, accG[0]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            store_partials(0, 2, accA, accG);
        }
    }
}

// Row chunks: (d / 64) column blocks in two halves; a group = (row chunk, half) keeps its d / 128 workgroups on one XCD, at
// most 32 workgroups (one per CU) there: 8 * floor(32 / (d / 128)) groups = half as many row chunks.
void k1_cols6_plan(int64_t M, int d, int* row_chunks, int64_t* rows_per_chunk) {
    const int cbh = d >= 128 ? d / 128 : 1;
    int64_t rc = cols_groups_max(cbh < 32 ? cbh : 32) / 2;
    const int64_t blocks32 = (M + 31) / 32;
    if (rc > blocks32) rc = blocks32;
    if (rc < 1) rc = 1;
    const int64_t per = (blocks32 + rc - 1) / rc;
    rc = (blocks32 + per - 1) / per;
    *row_chunks = (int)rc;
    *rows_per_chunk = per * 32;
}

// six tiles, bf16, saved activations -- and what the elementwise block of this kernel starts from: the forward's output y, or the additive
// gate (which needs neither h nor y).  (pet_cols6.hip, the round-3 form that recomputed h from x2, was removed in round 6: without y the
// six-tile backward takes the older two-pass form of pet_gate_bwd3.hip.)
bool k1_cols6_applies(const PetBwdArgs& a, int io_fp32) {
    return !io_fp32 && (a.flags & PET_GATE) && a.saved != nullptr && !drop_active(a.drop) && a.RT == 6 &&
           a.d % 128 == 0 && a.d / 128 <= 32 && ((a.flags & PET_GATE_ADD) != 0 || a.y != nullptr) && Colz6yGeo<6>::lds() <= (size_t)160 * 1024;
}

template <bool ADD, bool HAS_IN>
static hipError_t launch_cols6y_cfg(const ColzArgs& c, hipStream_t stream) {
    const size_t lds = Colz6yGeo<6>::lds();
    const int cbh = c.d / 128;
    const unsigned grid = cols_grid(cbh, 2 * c.row_chunks);
    auto kern = k1_cols6y_kernel<6, ADD, HAS_IN>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, c);
    return hipGetLastError();
}

hipError_t launch_k1_cols6y(const ColzArgs& c, hipStream_t stream) {
    const bool add = (c.flags & PET_GATE_ADD) != 0, in = c.dxin != nullptr;
    if (add) return in ? launch_cols6y_cfg<true, true>(c, stream) : launch_cols6y_cfg<true, false>(c, stream);
    return in ? launch_cols6y_cfg<false, true>(c, stream) : launch_cols6y_cfg<false, false>(c, stream);
}
