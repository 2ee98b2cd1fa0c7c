// K1 backward, column-parallel pass 2 for SIX bottleneck tiles (r = r_g = 192: the T5 script, scripts/image-text/
// T5-VL-PET-large.sh:41-59; autograd of my_transformers/modeling_t5.py:366-390,782-806).  Same decomposition, memory system and
// partial-sum layout as pet_cols.hip (read its header first); what changes is the register arithmetic.  At r = 192 a wave's
// slice of ONE weight gradient ([192 x 32] fp32) is 96 registers and its slice of one weight 48, so the two roles of
// pet_cols.hip (48 + 48 weights, 96 + 96 accumulators) become FOUR, two per SIMD:
//     UE: Wu, Wgu resident (96);   a_A, a_G, elementwise dh / dq -> LDS tiles          UW: dWu, dWgu (192) + bias sums
//     DE: Wd^T, Wgd^T resident (96); dx2 = s2*dh + p2, dx1 = p1 (+ dx1_in), stores    DW: dWd, dWgd (192)
// and a workgroup (8 waves) covers 64 columns (2 column quarters x 4 roles).  UW needs the dh / dq tiles of the SAME step (the
// z tiles it contracts them with live in the two-stage ring), so a step has two barriers: [stage hand-over] UE, DE, DW work,
// [dh / dq hand-over] UW works; UE and UW share a SIMD, so does the pair DE / DW.  dh goes to DE one step late, as in pet_cols.
// Cost of the narrower workgroup: the four [M, 192] bottleneck tensors are re-read by 12 column blocks instead of 6 (they are
// 1.5 KiB per row, as much as a row tensor) -- this kernel moves ~3x its row-tensor bytes through the LDS-DMA path and is bound
// by that; the six column blocks of a half share an XCD (L2), the two halves of a row chunk may sit on two.
#include "cols_common.h"

template <int RT, bool HAS_IN> struct Colz6Geo {
    static constexpr int KT = 2 * RT;
    static constexpr int PB = 64 * RT, NPR = PB / 16;   // bytes / 16-byte slots of a bottleneck row
    static constexpr int PT_B = 32 * PB;
    static constexpr int NX = HAS_IN ? 4 : 3;           // dy, x2, x1 (+ the incoming dx1): one pair tile [32 rows x 128 B] each
    static constexpr int X_B = NX * 4096;
    static constexpr int STG_B = X_B + 4 * PT_B;
    static constexpr int NSTG = 2;
    static constexpr int DH_OFF = NSTG * STG_B, DQ_OFF = DH_OFF + 2 * 4096, BIAS_OFF = DQ_OFF + 4096;
    static constexpr size_t lds() { return (size_t)BIAS_OFF + 2 * 64 * 4; }
};

template <int RT, bool ADD, bool HAS_IN>
__global__ __launch_bounds__(512, 2) void k1_cols6_kernel(ColzArgs a) {
    using GEO = Colz6Geo<RT, HAS_IN>;
    static_assert(RT % 2 == 0, "even tile counts (the bottleneck rows take the row tiles' swizzle)");
    constexpr int KT = GEO::KT, PB = GEO::PB, NPR = GEO::NPR, PT_B = GEO::PT_B, X_B = GEO::X_B, STG_B = GEO::STG_B, NX = GEO::NX;
    constexpr int NSTG = GEO::NSTG, DH_OFF = GEO::DH_OFF, DQ_OFF = GEO::DQ_OFF, BIAS_OFF = GEO::BIAS_OFF;
    constexpr int PR = 32 * RT;
    constexpr int NPW = 8 * RT / 4;                     // bottleneck pieces (1 KiB) per E wave and stage
    constexpr int GRP = 6;                              // B fragments per batch of a projection
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    // ---- which (column block, row chunk).  Groups = (row chunk, half of the column blocks): the column blocks of a group share
    // an XCD (block b runs on XCD b % 8) and with it the L2 copies of the bottleneck rows they all re-read
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
    if (tid < 128) {                                    // up-side biases of the workgroup's 64 columns -> LDS (fp32)
        float* sbias = reinterpret_cast<float*>(smem + BIAS_OFF);
        const uint8_t* pk = tid < 64 ? a.pk_a : a.pk_g;
        sbias[tid] = reinterpret_cast<const float*>(pk + pg.bias_off)[PR + 64 * cb + (tid & 63)];
    }

    // ---- stage pieces (1 KiB each): the E waves take the 8 RT pieces of the four bottleneck tiles (piece p of tensor t: 64 slots
    // of the 32-row tile), the W waves -- whose registers are accumulators -- only the row tensors (wave 4 + i: rows 8 i .. of
    // each).  All swizzles on the source side.
    const uint8_t* pbase[NPW]; uint32_t pdst[NPW], poff[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int q = (wave & 3) + 4 * j, t = q / KT, piece = q % KT;
        const int sig = piece * 64 + lane, pr = sig / NPR;
        pbase[j] = reinterpret_cast<const uint8_t*>(t == 0 ? a.z_a : t == 1 ? a.z_g : t == 2 ? a.dp_a : a.dp_g);
        poff[j] = (uint32_t)(pr * PB + ((sig % NPR) ^ fsw(pr)) * 16);
        pdst[j] = (uint32_t)(X_B + t * PT_B + piece * 1024);
    }
    const int xrow = 8 * (wave & 3) + (lane >> 3);
    const uint32_t xoff = (uint32_t)xrow * (uint32_t)ld2 + (uint32_t)(64 * cb * 2 + (((lane & 7) ^ fsw(xrow)) * 16));
    const uint8_t* xbase[NX];
#pragma unroll
    for (int t = 0; t < NX; ++t) xbase[t] = reinterpret_cast<const uint8_t*>(t == 0 ? a.dy : t == 1 ? a.x2 : t == 2 ? a.x1 : a.dxin);
    auto sbase = [](const uint8_t* p) {                 // a wave-uniform pointer as a fresh scalar (see pet_cols.hip)
        const uint64_t u = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const uint8_t*>(((uint64_t)hi << 32) | lo);
    };
    auto issue_p = [&](int s) {                         // (E waves)
        const int64_t rb = r_begin + 32 * (int64_t)s;
        uint8_t* st = smem + (size_t)(s % NSTG) * STG_B;
        const int last = (int)(r_end - rb) - 1;         // (>= 31 except in the last step: rows past the end re-read the last row)
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int pr = (int)(poff[j] / PB);
            glds16(sbase(pbase[j] + rb * PB) + poff[j] - (uint32_t)(pr > last ? pr - last : 0) * PB, st + pdst[j]);
        }
    };
    auto issue_x = [&](int s) {                         // (W waves)
        const int64_t rb = r_begin + 32 * (int64_t)s;
        uint8_t* st = smem + (size_t)(s % NSTG) * STG_B;
        const int last = (int)(r_end - rb) - 1;
        const uint32_t xo = xoff - (uint32_t)(xrow > last ? xrow - last : 0) * (uint32_t)ld2;
#pragma unroll
        for (int t = 0; t < NX; ++t) glds16_row(sbase(xbase[t] + rb * ld2) + xo, st + t * 4096 + (wave & 3) * 1024);
    };
    auto issue = [&](int s) { if (kind) issue_x(s); else issue_p(s); };

    // ---- per-lane LDS byte addresses (relative to the stage base)
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
                a_ptr[hi][par] = (uint32_t)(X_B + r * PB + (((4 * par + tslot) ^ fsw(r)) * 16) + thalf);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) a_xcl[k] = (uint32_t)(m * 128 + (((4 * nt + 2 * h + k) ^ fsw(m)) * 16));   // columns 8k .. of the lane's 16
#pragma unroll
        for (int k = 0; k < 4; ++k) a_pbf[k] = (uint32_t)(X_B + m * PB + (((2 * k + h) ^ fsw(m)) * 16));       // B fragment, k-step 4j + k (+ 128 j)
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
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    auto step_top = [&](int s, int extra) {             // wait for the own pieces of stage s, barrier, request stage s + 1
        vm_wait(extra);                                 // (two stages: nothing else of this wave is in flight beyond `extra`)
        __builtin_amdgcn_s_barrier();
        if (s + 1 < nsteps) issue(s + 1);
        const int valid = (int)(r_end - (r_begin + 32 * (int64_t)s));
        if (valid < 32) {                               // zero the bottleneck rows past the end (their products must vanish)
            const u32x4 z = {0u, 0u, 0u, 0u};
            uint8_t* pt = smem + (size_t)(s % NSTG) * STG_B + X_B;
            for (int q = tid; q < 4 * 32 * NPR; q += 512) {
                const int rr = (q / NPR) & 31;
                if (rr >= valid) *reinterpret_cast<u32x4*>(pt + (size_t)q * 16) = z;
            }
            __syncthreads();
        }
    };
    auto hand_over = [&]() {                            // the dh / dq tiles of this step are complete (UE -> UW)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
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
        bf16x8 wA[KT], wG[KT];
        {
            const int i = m, v = (i >> 2) & 1, ip = (i & 3) | (nt << 2) | ((i >> 3) << 3);
            const int64_t off = (int64_t)(side == 0 ? 1 : 3) * pg.pack_bytes + (int64_t)cb * (4 * RT * 1024)
                              + (int64_t)(v * KT) * 1024 + (ip + 32 * h) * 16;
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
                wA[ks] = *reinterpret_cast<const bf16x8*>(a.pk_a + off + ks * 1024);
                wG[ks] = *reinterpret_cast<const bf16x8*>(a.pk_g + off + ks * 1024);
            }
        }
        // acc (+)= W . B fragments of bottleneck tile T (lane = row); BIAS >= 0: the accumulator starts at the up-side bias there
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
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // weights in registers, biases in LDS
        if (nsteps > 0) issue(0);
        if (side == 0) {
            // ---------------------------------------------------------------- UE: up projections, dh / dq
            const float s2 = a.s2, sd = a.sd;
#pragma unroll 1
            for (int s = 0; s < nsteps; ++s) {
                const int valid = (int)(r_end - (r_begin + 32 * (int64_t)s));
                const uint32_t sb = lds0 + (uint32_t)((s % NSTG) * STG_B);
                const uint32_t dh0 = lds0 + (uint32_t)(DH_OFF + (s & 1) * 4096), dq0 = lds0 + (uint32_t)DQ_OFF;
                step_top(s, 0);
                f32x16 aA, aG;
                project(sb, I0{}, I0{}, wA, aA);
                project(sb, I1{}, std::integral_constant<int, 256>{}, wG, aG);
                const float gsr = m < valid ? a.gs : 0.f;
                u32x2 dyv[4], x2v[4];
                sfor<4>([&](auto C) {
                    lds_read8<8 * (C.value & 1)>(dyv[C.value], sb + a_xcl[C.value >> 1]);
                    lds_read8<4096 + 8 * (C.value & 1)>(x2v[C.value], sb + a_xcl[C.value >> 1]);
                });
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dyv[0]), "+v"(x2v[0]), "+v"(dyv[1]), "+v"(x2v[1]), "+v"(dyv[2]), "+v"(x2v[2]), "+v"(dyv[3]), "+v"(x2v[3]) :: "memory");
                sfor<4>([&](auto C) {
                    constexpr int c = C.value;
                    asm volatile("" : "+v"(dyv[c]), "+v"(x2v[c]), "+v"(aG[4 * c]), "+v"(aG[4 * c + 1]), "+v"(aG[4 * c + 2]), "+v"(aG[4 * c + 3]) :: "memory");
                    float dh[4], dq[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * c + j;
                        const float gt = sigm(aG[e]);
                        const float dyp = gsr * ((j & 1) ? bf_hi(dyv[c][j >> 1]) : bf_lo(dyv[c][j >> 1]));
                        if constexpr (ADD) {
                            dh[j] = dyp;
                            dq[j] = dyp * gt * (1.0f - gt);
                        } else {
                            const float hv = s2 * ((j & 1) ? bf_hi(x2v[c][j >> 1]) : bf_lo(x2v[c][j >> 1])) + sd * aA[e];
                            dh[j] = dyp * gt;
                            dq[j] = dh[j] * hv * (1.0f - gt);
                        }
                    }
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                    const bf16x4 th = {(__bf16)dh[0], (__bf16)dh[1], (__bf16)dh[2], (__bf16)dh[3]};
                    const bf16x4 tq = {(__bf16)dq[0], (__bf16)dq[1], (__bf16)dq[2], (__bf16)dq[3]};
                    lds_write8<8 * (c & 1)>(dh0 + a_xcl[c >> 1], __builtin_bit_cast(u32x2, th));
                    lds_write8<8 * (c & 1)>(dq0 + a_xcl[c >> 1], __builtin_bit_cast(u32x2, tq));
                });
                hand_over();
            }
            __builtin_amdgcn_s_barrier();                                 // the last dh tile is visible to DE
        } else {
            // ---------------------------------------------------------------- DE: input gradients, one step late
            const float s2 = a.s2;
            f32x16 p2 = zero16(), p1 = zero16();
            u32x4 dinA = {0u, 0u, 0u, 0u}, dinB = {0u, 0u, 0u, 0u};
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
                const uint32_t sb = lds0 + (uint32_t)((s % NSTG) * STG_B);
                step_top(s, s >= 2 ? 4 : 0);                              // (the four output stores of step s - 2 are younger than stage s)
                if (s > 0) finish(s - 1);
                if constexpr (HAS_IN) {
                    lds_read16<3 * 4096>(dinA, sb + a_xcl[0]); lds_read16<3 * 4096>(dinB, sb + a_xcl[1]);
                    lgkm_fence(dinA); lgkm_tie(dinB);
                }
                p2 = zero16(); p1 = zero16();
                project(sb, I2{}, std::integral_constant<int, -1>{}, wA, p2);
                project(sb, I3{}, std::integral_constant<int, -1>{}, wG, p1);
                hand_over();
            }
            __builtin_amdgcn_s_barrier();                                 // UE has written the last dh tile
            if (nsteps > 0) finish(nsteps - 1);
        }
    } else {
        // ================================================================ the W roles: accumulators, no weights
        f32x16 accA[RT], accG[RT];
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) { accA[ct] = zero16(); accG[ct] = zero16(); }
        // acc[ct] += P^T (tile TP) . X (row tile at xlo / xhi); one 16-row k-step and three column tiles at a time (208 of the 256
        // registers are accumulators here); SLOT >= 0: the column sums of X go to that slot of sx
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
            // ---------------------------------------------------------------- UW: dWu, dWgu, every bias sum
            const bool want_csp = cb == 0 && nt == 0;
            f32x16 sx = zero16();
#pragma unroll 1
            for (int s = 0; s < nsteps; ++s) {
                const uint32_t sb = lds0 + (uint32_t)((s % NSTG) * STG_B);
                const uint32_t dh0 = lds0 + (uint32_t)(DH_OFF + (s & 1) * 4096), dq0 = lds0 + (uint32_t)DQ_OFF;
                step_top(s, 0);
                hand_over();                                              // UE has written this step's dh, dq
                wg_products(sb, I0{}, I0{}, dh0 + a_xtr[0], dh0 + a_xtr[1], accA, sx);
                wg_products(sb, I1{}, I1{}, dq0 + a_xtr[0], dq0 + a_xtr[1], accG, sx);
                if (want_csp) {                                           // column sums of dpre_a, dpre_g: one wave per row chunk
                    sfor<2>([&](auto KS) {
                        constexpr int ks = KS.value;
                        sfor<2>([&](auto TT) {
                            sfor<RT / 3>([&](auto CG) {
                                TrOp ap[3];
                                sfor<3>([&](auto C3) {
                                    constexpr int ct = CG.value * 3 + C3.value;
                                    tr_read2<(2 + TT.value) * PT_B + 128 * (ct >> 1) + ks * 16 * PB>(ap[C3.value], sb + a_ptr[0][ct & 1], sb + a_ptr[1][ct & 1]);
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
            __builtin_amdgcn_s_barrier();
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
                const uint32_t sb = lds0 + (uint32_t)((s % NSTG) * STG_B);
                step_top(s, 0);
                wg_products(sb, I2{}, NOSLOT{}, sb + 4096 + a_xtr[0], sb + 4096 + a_xtr[1], accA, accA[0]);
                wg_products(sb, I3{}, NOSLOT{}, sb + 8192 + a_xtr[0], sb + 8192 + a_xtr[1], accG, accG[0]);
                hand_over();
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

bool k1_cols6_applies(const PetBwdArgs& a, int io_fp32) {
    return !io_fp32 && (a.flags & PET_GATE) && a.saved != nullptr && !drop_active(a.drop) && a.RT == 6 &&
           a.d % 128 == 0 && a.d / 128 <= 32;
}

template <bool ADD, bool HAS_IN>
static hipError_t launch_cols6_cfg(const ColzArgs& c, hipStream_t stream) {
    const size_t lds = Colz6Geo<6, HAS_IN>::lds();
    auto kern = k1_cols6_kernel<6, ADD, HAS_IN>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int cbh = c.d / 128;
    const unsigned grid = cols_grid(cbh, 2 * c.row_chunks);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, c);
    return hipGetLastError();
}

hipError_t launch_k1_cols6(const ColzArgs& c, hipStream_t stream) {
    const bool add = (c.flags & PET_GATE_ADD) != 0, in = c.dxin != nullptr;
    if (add) return in ? launch_cols6_cfg<true, true>(c, stream) : launch_cols6_cfg<true, false>(c, stream);
    return in ? launch_cols6_cfg<false, true>(c, stream) : launch_cols6_cfg<false, false>(c, stream);
}
