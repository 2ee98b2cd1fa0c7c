// K1 backward, column-parallel pass 2 at r <= 96, second form (round 5): elementwise block from the forward's output, single-reader
// row tensors in registers, the up side's weight-gradient products one step late.  Autograd of my_transformers/modeling_bart.py:
// 1147-1155, 1195-1209 (T5: my_transformers/modeling_t5.py:366-390, 782-806).  Read pet_cols.hip first: same decomposition (a 128-column
// workgroup, a column quarter = the U wave + the D wave of one SIMD, 32-row steps, one barrier per step), same partial-sum layout.
//
// What changes, and why (pet_cols6y.hip's header has the r = 192 version of the argument):
//   * y = gs * h * g is at hand (vlpet_adapter_gate_bwd_saved_y), so dh = gs dy g, dq = dy y (1 - g): the U wave needs the GATE chain's
//     up projection only -- the adapter chain's 6 MFMAs, fragment reads, bias and 24 weight registers are gone, and so is its x2 read;
//   * a D lane needs exactly its own 32 bytes of dx1_in per row: it loads them itself, one step ahead, straight into registers.  (The U
//     lanes' dy / y stay tiles of the LDS-DMA ring here: single-buffered register loads of them -- all the U wave's 248 registers allow
//     next to its accumulators -- exposed their latency in every step: 99 vs 74 us, profiles/r05_k1_colsy_ab.txt.)  The ring carries
//     dy, y, x2, x1 and the four bottleneck tiles; without the dx1_in tile there is room for a third z slot;
//   * with that slot the U wave's products run ONE STEP LATE (dWu += z_a(s-1)^T dh(s-1), dWgu likewise) and sit BETWEEN the gate projection
//     and the elementwise block of step s: the step's matrix-core work (6 + 12 MFMAs) is issued first, the ~0.75 us of VALU work of the
//     elementwise block then runs while the products execute -- in pet_cols.hip the products' operands were this step's dh / dq, i.e. the
//     U wave was projection -> elementwise -> LDS round trip -> products, a 2.4-us chain per step that IS the step.
// MEASURED (profiles/r05_k1_colsy_ab.txt): parity-green (187 cases) and slower than pet_cols.hip -- pass 2 + finalize 79.4 vs 75.7 us warm at
// 28,000 rows, 87.8 vs 82.7 cold, 93 vs 78 us per launch inside the configs[1] step.  At r = 192 the same restructuring removed a second
// barrier per step (pet_cols6y.hip: -20 %); here there was none to remove, the reordering buys less than a step's extra LDS traffic
// costs, and y is a FOURTH row stream of a pass that runs at what its row streams cost when they are cold (profiles/r04_store_probe.txt).
// Off by default (csrc/tuning.h colsy); kept with its test.
// LDS (r = 96): z ring 3 x 12 KiB, dpre ring 2 x 12, row ring 2 x [dy | y | x2 | x1] 32, dh 2 x 8, dq 2 x 8, biases: 157 KiB.
#include "cols_common.h"

template <int RT> struct ColzyGeo {
    static constexpr int KT = 2 * RT;
    static constexpr int PB = 64 * RT;                  // bytes of a bottleneck row
    static constexpr int PT_B = 32 * PB;                // one bottleneck tile
    static constexpr int NZ = 3, ND = 2, NXS = 2;
    static constexpr int XR_B = 4 * 2 * 4096;           // [dy | y | x2 | x1], two pair tiles [32 rows x 128 B] each
    static constexpr int Z_OFF = 0, DP_OFF = Z_OFF + NZ * 2 * PT_B, XR_OFF = DP_OFF + ND * 2 * PT_B;
    static constexpr int DH_OFF = XR_OFF + NXS * XR_B, DQ_OFF = DH_OFF + 2 * 8192, BIAS_OFF = DQ_OFF + 2 * 8192;
    static constexpr size_t lds() { return (size_t)BIAS_OFF + 2 * 128 * 4; }
};

template <int RT, bool ADD, bool HAS_IN>
__global__ __launch_bounds__(512, 2) void k1_colsy_kernel(ColzArgs a) {
    using GEO = ColzyGeo<RT>;
    constexpr int KT = GEO::KT, PB = GEO::PB, PT_B = GEO::PT_B, XR_B = GEO::XR_B;
    constexpr int Z_OFF = GEO::Z_OFF, DP_OFF = GEO::DP_OFF, XR_OFF = GEO::XR_OFF, DH_OFF = GEO::DH_OFF, DQ_OFF = GEO::DQ_OFF, BIAS_OFF = GEO::BIAS_OFF;
    constexpr int PR = 32 * RT;
    constexpr int GRP = KT <= 6 ? KT : 6;               // B fragments per batch of a projection
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int d = a.d, NCB = d >> 7;
    int rc, cb;
    cols_decode((int)blockIdx.x, NCB, rc, cb);
    if (rc >= a.row_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, wc = wave & 3;          // waves w and w + 4 share a SIMD: the up side and the down side of a column quarter
    const int pp = wc >> 1, nt = wc & 1;
    const int m = lane & 31, h = lane >> 5;
    const int64_t ld2 = (int64_t)d * 2;
    const int c0 = 128 * cb + 32 * wc;
    const int64_t r_begin = (int64_t)rc * a.rows_per_chunk;
    int64_t r_end = r_begin + a.rows_per_chunk;
    if (r_end > a.M) r_end = a.M;
    const int nsteps = r_end > r_begin ? (int)((r_end - r_begin + 31) >> 5) : 0;
    const PackGeom pg = pack_geom(RT, d, 1);

    // ---- the stage pieces (1 KiB each) of this wave: two row pieces (tensor t = x2 / x1, pair tile wave >> 2, rows 8 (wave & 3) ..) and
    // RT bottleneck pieces q' = wave + 8 j (tensor q' / KT, piece q' % KT of the 32 contiguous rows)
    const int xrow = 8 * (wave & 3) + (lane >> 3);
    const uint32_t xoff = (uint32_t)xrow * (uint32_t)ld2 + (uint32_t)((128 * cb + 64 * (wave >> 2)) * 2 + (((lane & 7) ^ fsw(xrow)) * 16));
    const uint8_t* xbase[4] = {reinterpret_cast<const uint8_t*>(a.dy), reinterpret_cast<const uint8_t*>(ADD ? a.dy : a.y),      // (additive gate: the y tile is not used)
                               reinterpret_cast<const uint8_t*>(a.x2), reinterpret_cast<const uint8_t*>(a.x1)};
    const uint32_t xdst = (uint32_t)((wave >> 2) * 4096 + (wave & 3) * 1024);
    const uint8_t* pbase[RT]; uint32_t pdst[RT], poff[RT]; int prow[RT], pring[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
        const int q = wave + 8 * j, t = q / KT, piece = q % KT;
        const int sig = piece * 64 + lane;
        prow[j] = sig / (PB / 16);
        pbase[j] = reinterpret_cast<const uint8_t*>(t == 0 ? a.z_a : t == 1 ? a.z_g : t == 2 ? a.dp_a : a.dp_g);
        poff[j] = (uint32_t)(prow[j] * PB + ((sig % (PB / 16)) ^ gsw(prow[j])) * 16);
        pring[j] = t >> 1;                                                 // 0: the z ring, 1: the dpre ring
        pdst[j] = (uint32_t)((t & 1) * PT_B + piece * 1024);
    }
    auto sbase = [](const uint8_t* p) {
        const uint64_t u = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const uint8_t*>(((uint64_t)hi << 32) | lo);
    };
    auto issue = [&](int s) {
        const int64_t rb = r_begin + 32 * (int64_t)s;
        uint8_t* xs = smem + XR_OFF + (size_t)(s % GEO::NXS) * XR_B;
        uint8_t* zs = smem + Z_OFF + (size_t)(s % GEO::NZ) * (2 * PT_B);
        uint8_t* ds = smem + DP_OFF + (size_t)(s % GEO::ND) * (2 * PT_B);
        const int last = (int)(r_end - rb) - 1;         // (>= 31 except in the last step: rows past the end re-read the last row)
        const uint32_t xo = xoff - (uint32_t)(xrow > last ? xrow - last : 0) * (uint32_t)ld2;
#pragma unroll
        for (int t = 0; t < 4; ++t) glds16_row(sbase(xbase[t] + rb * ld2) + xo, xs + t * 8192 + xdst);
#pragma unroll
        for (int j = 0; j < RT; ++j)
            glds16(sbase(pbase[j] + rb * PB) + poff[j] - (uint32_t)(prow[j] > last ? prow[j] - last : 0) * PB, (pring[j] ? ds : zs) + pdst[j]);
    };
    // the 32 bytes (columns c0 + 16 h .. + 15) of row m of step s of a row tensor that only this lane needs: straight into registers
    auto lane_row = [&](const void* base, int s, u32x4& lo, u32x4& hi) {
        const int64_t rb = r_begin + 32 * (int64_t)s;
        const int last = (int)(r_end - rb) - 1;
        const uint8_t* p = sbase(reinterpret_cast<const uint8_t*>(base) + rb * ld2) + (uint32_t)(m > last ? last : m) * (uint32_t)ld2 + (uint32_t)((c0 + 16 * h) * 2);
        lo = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        hi = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p) + 1);
    };

    // ---- per-lane LDS byte addresses (bottleneck tiles: relative to their ring slot; row tiles: relative to the tensor's pair tiles)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    uint32_t a_xtr[2], a_ptr[2], a_xcl[2], a_pbf[2];
    {
        const int g4 = lane >> 4, sl = lane & 15;
        const int trow = 8 * (g4 >> 1) + (sl >> 2);
        const int tslot = 2 * (g4 & 1) + ((sl & 3) >> 1), thalf = 8 * (sl & 1);
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int r = trow + 4 * hi;
            a_xtr[hi] = (uint32_t)(pp * 4096 + r * 128 + (((4 * nt + tslot) ^ fsw(r)) * 16) + thalf);
            a_ptr[hi] = (uint32_t)(r * PB + ((tslot ^ gsw(r)) * 16) + thalf);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            a_xcl[k] = (uint32_t)(pp * 4096 + m * 128 + (((4 * nt + 2 * h + k) ^ fsw(m)) * 16));
            a_pbf[k] = (uint32_t)(m * PB + (((2 * k + h) ^ gsw(m)) * 16));
        }
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

    f32x16 accA[RT], accG[RT], sx = zero16();
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) { accA[ct] = zero16(); accG[ct] = zero16(); }

    // the one barrier of a step.  Ring reuse: z slot (s + 1) % 3 was last read by the U waves' late products in step s - 1, the dpre /
    // row slots (s + 1) % 2 by the D waves in step s - 1, dh / dq slot s & 1 (written by U in step s) by D / U in step s - 1.
    auto step_top = [&](int s, int extra) {
        vm_wait(extra);
        __builtin_amdgcn_s_barrier();
        if (s + 1 < nsteps) issue(s + 1);
        const int valid = (int)(r_end - (r_begin + 32 * (int64_t)s));
        if (valid < 32) {                               // zero the bottleneck rows past the end (their products must vanish)
            const u32x4 z = {0u, 0u, 0u, 0u};
            uint8_t* zt = smem + Z_OFF + (size_t)(s % GEO::NZ) * (2 * PT_B);
            uint8_t* dt = smem + DP_OFF + (size_t)(s % GEO::ND) * (2 * PT_B);
            for (int q = tid; q < 2 * 32 * (PB / 16); q += 512) {
                const int rr = (q / (PB / 16)) & 31;
                if (rr >= valid) { *reinterpret_cast<u32x4*>(zt + (size_t)q * 16) = z; *reinterpret_cast<u32x4*>(dt + (size_t)q * 16) = z; }
            }
            __syncthreads();
        }
    };
    // weight-gradient products of one job: acc[ct] += P^T (tile TP of the ring slot at `sb`) . X (row tile at xlo / xhi)
    // BOTH: the operands of both 16-row k-steps in one batch (one LDS round trip, 32 registers in flight: the U wave, whose chain is the step)
    // or one k-step at a time (16 registers: the D wave, which also holds two projections and the dx1_in rows)
    auto wg_products = [&](uint32_t sb, auto TPC, uint32_t xlo, uint32_t xhi, f32x16* acc, auto SLOTC, auto BOTHC) {     // SLOT >= 0: the column sums of X go to that slot of sx
        constexpr int TP = decltype(TPC)::value, SLOT = decltype(SLOTC)::value;
        if constexpr (decltype(BOTHC)::value) {
            TrOp bx[2], ap[2][RT];
            sfor<2>([&](auto KS) {
                constexpr int ks = KS.value;
                tr_read2<ks * 16 * 128>(bx[ks], xlo, xhi);
                sfor<RT>([&](auto CT) { tr_read2<TP * PT_B + 64 * CT.value + ks * 16 * PB>(ap[ks][CT.value], sb + a_ptr[0], sb + a_ptr[1]); });
            });
            tr_fence(bx[0]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks) tr_tie(bx[ks]);
                const bf16x8 vx = tr_val(bx[ks]);
                if constexpr (SLOT >= 0) sx = mfma32(ones_row(SLOT), vx, sx);
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) { tr_tie(ap[ks][ct]); acc[ct] = mfma32(tr_val(ap[ks][ct]), vx, acc[ct]); }
            }
        } else {
            sfor<2>([&](auto KS) {
                constexpr int ks = KS.value;
                TrOp bx, ap[RT];
                tr_read2<ks * 16 * 128>(bx, xlo, xhi);
                sfor<RT>([&](auto CT) { tr_read2<TP * PT_B + 64 * CT.value + ks * 16 * PB>(ap[CT.value], sb + a_ptr[0], sb + a_ptr[1]); });
                tr_fence(bx);
                const bf16x8 vx = tr_val(bx);
                if constexpr (SLOT >= 0) sx = mfma32(ones_row(SLOT), vx, sx);
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) { tr_tie(ap[ct]); acc[ct] = mfma32(tr_val(ap[ct]), vx, acc[ct]); }
            });
        }
    };

    if (nsteps > 0) issue(0);                           // (first requests before the resident operands: pet_dz2.hip's note)
    if (tid < 256) {                                    // up-side biases of the workgroup's 128 columns -> LDS (fp32); U uses the gate's
        float* sbias = reinterpret_cast<float*>(smem + BIAS_OFF);
        const uint8_t* pk = tid < 128 ? a.pk_a : a.pk_g;
        sbias[tid] = reinterpret_cast<const float*>(pk + pg.bias_off)[PR + 128 * cb + (tid & 127)];
    }
    // resident weights (A fragments of this wave's 32 columns; pet_cols.hip's note).  Role U: Wgu only; role D: Wd^T, Wgd^T.
    bf16x8 wA[KT], wG[KT];
    {
        const int i = m, v = (i >> 2) & 1, ip = (i & 3) | (nt << 2) | ((i >> 3) << 3);
        const int64_t off = (int64_t)(role == 0 ? 1 : 3) * pg.pack_bytes + (int64_t)(2 * cb + pp) * (4 * RT * 1024)
                          + (int64_t)(v * KT) * 1024 + (ip + 32 * h) * 16;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            if (role != 0) wA[ks] = *reinterpret_cast<const bf16x8*>(a.pk_a + off + ks * 1024);
            wG[ks] = *reinterpret_cast<const bf16x8*>(a.pk_g + off + ks * 1024);
        }
    }

    if (role == 0) {
        // ================================================================ role U: gate projection, late products, dh / dq, bias sums
        int csp_tile[2] = {-1, -1};
        {
            const int w = wc * NCB + cb;
            if (w < 2 * RT) csp_tile[0] = w;
            if (w + 4 * NCB < 2 * RT) csp_tile[1] = w + 4 * NCB;
        }
        const bool want_csp = csp_tile[0] >= 0;
        const uint32_t a_bias = lds0 + (uint32_t)(BIAS_OFF + 512 + (32 * wc + 16 * h) * 4);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // weights in registers, biases in LDS (and the first stage landed)
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) asm volatile("" : "+v"(wG[ks]));
        auto late = [&](int sp) {                       // dWu += z_a(sp)^T dh(sp), dWgu += z_g(sp)^T dq(sp): this wave's own columns of step sp
            const uint32_t dh0 = lds0 + (uint32_t)(DH_OFF + (sp & 1) * 8192), dq0 = lds0 + (uint32_t)(DQ_OFF + (sp & 1) * 8192);
            wg_products(zslot(sp), I0{}, dh0 + a_xtr[0], dh0 + a_xtr[1], accA, I0{}, std::true_type{});
            wg_products(zslot(sp), I1{}, dq0 + a_xtr[0], dq0 + a_xtr[1], accG, I1{}, std::true_type{});
        };
#pragma unroll 1
        for (int s = 0; s < nsteps; ++s) {
            const int valid = (int)(r_end - (r_begin + 32 * (int64_t)s));
            const uint32_t dh0 = lds0 + (uint32_t)(DH_OFF + (s & 1) * 8192), dq0 = lds0 + (uint32_t)(DQ_OFF + (s & 1) * 8192);
            step_top(s, 0);
            // (1) the gate's up projection of stage s, starting at its bias
            f32x16 aG;
            {
                const uint32_t sb = zslot(s);
                u32x4 bb[4], bf0[GRP];
                sfor<4>([&](auto Q) { lds_read16<16 * Q.value>(bb[Q.value], a_bias); });
                sfor<GRP>([&](auto K) { lds_read16<PT_B + 64 * (K.value >> 1)>(bf0[K.value], sb + a_pbf[K.value & 1]); });
                lgkm_fence(bb[0]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q) lgkm_tie(bb[q]);
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) aG[4 * q + w2] = __uint_as_float(bb[q][w2]);
                }
#pragma unroll
                for (int k = 0; k < GRP; ++k) { lgkm_tie(bf0[k]); aG = mfma32(wG[k], as_bf(bf0[k]), aG); }
                sfor<KT / GRP - 1>([&](auto G) {
                    u32x4 bf[GRP];
                    sfor<GRP>([&](auto K) {
                        constexpr int ks = (G.value + 1) * GRP + K.value;
                        lds_read16<PT_B + 64 * (ks >> 1)>(bf[K.value], sb + a_pbf[ks & 1]);
                    });
                    lgkm_fence(bf[0]);
#pragma unroll
                    for (int k = 0; k < GRP; ++k) { if (k) lgkm_tie(bf[k]); aG = mfma32(wG[(G.value + 1) * GRP + k], as_bf(bf[k]), aG); }
                });
            }
            // (2) the previous step's products: queued behind the projection in the matrix pipe, executing while (3) runs on the VALU
            if (s > 0) late(s - 1);
            // (3) elementwise: dh, dq of step s from dy, y (registers) and the gate -> this wave's columns of the dh / dq tiles
            {
                const float live = m < valid ? 1.f : 0.f, gsr = live * a.gs;
                const uint32_t xs = xslot(s);
                u32x2 dyv[4], yv[4];
                sfor<4>([&](auto C) {
                    lds_read8<8 * (C.value & 1)>(dyv[C.value], xs + a_xcl[C.value >> 1]);
                    lds_read8<8192 + 8 * (C.value & 1)>(yv[C.value], xs + a_xcl[C.value >> 1]);
                });
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dyv[0]), "+v"(yv[0]), "+v"(dyv[1]), "+v"(yv[1]), "+v"(dyv[2]), "+v"(yv[2]), "+v"(dyv[3]), "+v"(yv[3]) :: "memory");
                sfor<4>([&](auto C) {
                    constexpr int c = C.value;
                    asm volatile("" : "+v"(dyv[c]), "+v"(yv[c]), "+v"(aG[4 * c]), "+v"(aG[4 * c + 1]), "+v"(aG[4 * c + 2]), "+v"(aG[4 * c + 3]) :: "memory");
                    float dh[4], dq[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * c + j;
                        const float gt = sigm(aG[e]);
                        const float dyr = (j & 1) ? bf_hi(dyv[c][j >> 1]) : bf_lo(dyv[c][j >> 1]);
                        if constexpr (ADD) {
                            dh[j] = gsr * dyr;
                            dq[j] = dh[j] * gt * (1.0f - gt);
                        } else {
                            dh[j] = gsr * dyr * gt;
                            dq[j] = live * dyr * ((j & 1) ? bf_hi(yv[c][j >> 1]) : bf_lo(yv[c][j >> 1])) * (1.0f - gt);
                        }
                    }
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                    const bf16x4 th = {(__bf16)dh[0], (__bf16)dh[1], (__bf16)dh[2], (__bf16)dh[3]};
                    const bf16x4 tq = {(__bf16)dq[0], (__bf16)dq[1], (__bf16)dq[2], (__bf16)dq[3]};
                    lds_write8<8 * (c & 1)>(dh0 + a_xcl[c >> 1], __builtin_bit_cast(u32x2, th));
                    lds_write8<8 * (c & 1)>(dq0 + a_xcl[c >> 1], __builtin_bit_cast(u32x2, tq));
                });
            }
            if (want_csp) {                             // column sums of the dpre tiles of THIS stage (pet_cols.hip's distribution over the U waves)
                const uint32_t sb = dslot(s);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = csp_tile[j];
                    if (k < 0) break;
                    const uint32_t off = (uint32_t)((k / RT) * PT_B + 64 * (k % RT));
                    TrOp ap[2];
                    tr_read2<0>(ap[0], sb + a_ptr[0] + off, sb + a_ptr[1] + off);
                    tr_read2<16 * PB>(ap[1], sb + a_ptr[0] + off, sb + a_ptr[1] + off);
                    tr_fence(ap[0]);
                    sx = mfma32(ones_row(2 + j), tr_val(ap[0]), sx);
                    tr_tie(ap[1]);
                    sx = mfma32(ones_row(2 + j), tr_val(ap[1]), sx);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (every LDS access of this step is complete at the next barrier)
        }
        __builtin_amdgcn_s_barrier();                                     // the last dh tile is visible to role D
        if (nsteps > 0) late(nsteps - 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (h == 0) {
            a.part[1][(int64_t)RC * PR * d + (int64_t)rc * d + col] = sx[0];
            a.part[3][(int64_t)RC * PR * d + (int64_t)rc * d + col] = sx[1];
            if (want_csp) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = csp_tile[j];
                    if (k < 0) break;
                    float* ps = a.part[k / RT == 0 ? 0 : 2] + (int64_t)RC * PR * d + (int64_t)RC * d + (int64_t)rc * PR;
                    ps[32 * (k % RT) + m] = sx[2 + j];
                }
            }
        }
    } else {
        // ================================================================ role D: dWd, dWgd, input gradients (one step late)
        const float s2 = a.s2;
        f32x16 p2 = zero16(), p1 = zero16();
        u32x4 dinA = {0u, 0u, 0u, 0u}, dinB = {0u, 0u, 0u, 0u};
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) { asm volatile("" : "+v"(wA[ks])); asm volatile("" : "+v"(wG[ks])); }
        constexpr int GD = (HAS_IN && KT % 3 == 0) ? 3 : GRP;    // B fragments per batch here (with the dx1_in rows in registers: three, not six)
        auto project = [&](uint32_t sb, auto TC, const bf16x8* w, f32x16& acc) {
            constexpr int T = decltype(TC)::value;
            sfor<KT / GD>([&](auto G) {
                u32x4 bf[GD];
                sfor<GD>([&](auto K) {
                    constexpr int ks = G.value * GD + K.value;
                    lds_read16<T * PT_B + 64 * (ks >> 1)>(bf[K.value], sb + a_pbf[ks & 1]);
                });
                lgkm_fence(bf[0]);
#pragma unroll
                for (int k = 0; k < GD; ++k) { if (k) lgkm_tie(bf[k]); acc = mfma32(w[G.value * GD + k], as_bf(bf[k]), acc); }
            });
        };
        auto finish = [&](int sp) {                     // input gradients of step sp (its dh tile is complete; dinA / dinB hold its dx1_in rows)
            const int64_t rb = r_begin + 32 * (int64_t)sp;
            const int valid = (int)(r_end - rb) < 32 ? (int)(r_end - rb) : 32;
            const bool row_ok = m < valid;
            const uint32_t rowoff = (uint32_t)(row_ok ? m : valid - 1) * (uint32_t)ld2 + (uint32_t)((c0 + 16 * h) * 2);
            const uint32_t dh0 = lds0 + (uint32_t)(DH_OFF + (sp & 1) * 8192);
            u32x4 dhv0, dhv1;
            lds_read16<0>(dhv0, dh0 + a_xcl[0]); lds_read16<0>(dhv1, dh0 + a_xcl[1]);
            lgkm_fence(dhv0); lgkm_tie(dhv1);
            {
                float o[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] = s2 * bf_at(e < 8 ? dhv0 : dhv1, e & 7) + p2[e];
                const u32x4 v0 = pack8(o), v1 = pack8(o + 8);
                if (row_ok) {
                    uint8_t* q2 = const_cast<uint8_t*>(sbase(reinterpret_cast<const uint8_t*>(a.dx2) + rb * ld2)) + rowoff;
                    reinterpret_cast<u32x4*>(q2)[0] = v0;
                    reinterpret_cast<u32x4*>(q2)[1] = v1;
                }
            }
            {
                float o[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] = p1[e] + (HAS_IN ? bf_at(e < 8 ? dinA : dinB, e & 7) : 0.f);
                const u32x4 v0 = pack8(o), v1 = pack8(o + 8);
                if (row_ok) {
                    uint8_t* q1 = const_cast<uint8_t*>(sbase(reinterpret_cast<const uint8_t*>(a.dx1) + rb * ld2)) + rowoff;
                    reinterpret_cast<u32x4*>(q1)[0] = v0;
                    reinterpret_cast<u32x4*>(q1)[1] = v1;
                }
            }
        };
#pragma unroll 1
        for (int s = 0; s < nsteps; ++s) {
            const uint32_t xs = xslot(s);
            step_top(s, (s >= 2 ? 4 : 0) + (HAS_IN && s >= 1 ? 2 : 0));  // (younger than stage s: the four output stores of step s - 2, the two dx1_in loads of step s - 1)
            if (s > 0) finish(s - 1);
            if constexpr (HAS_IN) lane_row(a.dxin, s, dinA, dinB);      // this step's incoming rows: used by finish(s) in step s + 1
            wg_products(dslot(s), I0{}, xs + 16384 + a_xtr[0], xs + 16384 + a_xtr[1], accA, std::integral_constant<int, -1>{}, std::true_type{});
            wg_products(dslot(s), I1{}, xs + 24576 + a_xtr[0], xs + 24576 + a_xtr[1], accG, std::integral_constant<int, -1>{}, std::true_type{});
            p2 = zero16(); p1 = zero16();
            project(dslot(s), I0{}, wA, p2);
            project(dslot(s), I1{}, wG, p1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                                     // role U has written the last dh tile
        if (nsteps > 0) finish(nsteps - 1);
    }
    // ---- this row chunk's partial sums, in wgrad.hip's workspace layout (wgrad_finalize_kernel sums the chunks)
    {
        float* tA = a.part[role == 0 ? 1 : 0] + (int64_t)rc * PR * d;
        float* tG = a.part[role == 0 ? 3 : 2] + (int64_t)rc * PR * d;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int crow = 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h;
                tA[(int64_t)crow * d + col] = accA[ct][i];
                tG[(int64_t)crow * d + col] = accG[ct][i];
            }
    }
}

bool k1_colsy_applies(const ColzArgs& c, int RT) {
    return vlpet_tuning().colsy != 0 && (RT == 1 || RT == 3) && ((c.flags & PET_GATE_ADD) != 0 || c.y != nullptr);
}

template <int RT, bool ADD, bool HAS_IN>
static hipError_t launch_colsy_cfg(const ColzArgs& c, hipStream_t stream) {
    const size_t lds = ColzyGeo<RT>::lds();
    auto kern = k1_colsy_kernel<RT, ADD, HAS_IN>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int ncb = c.d / 128;
    const unsigned grid = cols_grid(ncb, c.row_chunks);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, c);
    return hipGetLastError();
}
template <int RT>
static hipError_t launch_colsy_rt(const ColzArgs& c, hipStream_t stream) {
    const bool add = (c.flags & PET_GATE_ADD) != 0, in = c.dxin != nullptr;
    if (add) return in ? launch_colsy_cfg<RT, true, true>(c, stream) : launch_colsy_cfg<RT, true, false>(c, stream);
    return in ? launch_colsy_cfg<RT, false, true>(c, stream) : launch_colsy_cfg<RT, false, false>(c, stream);
}
hipError_t launch_k1_colsy(const ColzArgs& c, int RT, hipStream_t stream) {
    if (RT == 1) return launch_colsy_rt<1>(c, stream);
    if (RT == 3) return launch_colsy_rt<3>(c, stream);
    return hipErrorInvalidValue;
}
