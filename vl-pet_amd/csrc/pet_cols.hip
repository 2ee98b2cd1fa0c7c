// K1 backward, pass 2 of the two-pass form (round 3 rebuild): column-parallel, weights resident in registers, every row
// tensor streamed once.  Autograd of my_transformers/modeling_bart.py:1147-1155,1195-1209 (T5: my_transformers/
// modeling_t5.py:366-390,782-806) given the dpre of pass 1 (pet_gate_dz_kernel).
//
// Why this shape.  The row kernels of this op are latency chains: a workgroup owns 128 rows, streams 0.6 MB of packed
// weights through its LDS for them and synchronises at every 64-feature stage, so its time hardly depends on its rows and
// the chip runs them in rounds of 256.  Everything that is left after pass 1 is separable by COLUMN:
//     dx2[m,f] = s2*dh[m,f] + sum_c Wd[c,f] dpre_a[m,c]        dx1[m,f] = sum_c Wgd[c,f] dpre_g[m,c] (+ dx1_in[m,f])
//     dh = gs*dy*g,  dq = dh*h*(1-g),  h = s2*x2 + sd*(bu + Wu z_a),  g = sigmoid(bgu + Wgu z_g)     (f by f)
//     dWu[f,c] = sd * sum_m dh[m,f] z_a[m,c]     dWgu[f,c] = sum_m dq[m,f] z_g[m,c]
//     dWd[c,f] = sum_m dpre_a[m,c] x2[m,f]       dWgd[c,f] = sum_m dpre_g[m,c] x1[m,f]
// so a workgroup owns 128 columns and a chunk of rows; a column quarter (32 columns = one MFMA tile) belongs to TWO waves of
// one SIMD: the up side keeps Wu / Wgu of those columns (48 registers at r = 96) and the dWu / dWgu accumulators (96), the down
// side Wd^T / Wgd^T and the dWd / dWgd accumulators, for the whole launch.  Rows stream by 32 at a time: dy, x1, x2 are read
// once, dx1, dx2 written once (5 units = the op's algorithmic traffic), plus the four [M, 32*RT] bottleneck tensors (z_a, z_g,
// dpre_a, dpre_g), which the d/128 column blocks of a row chunk re-read through the L2 of one XCD.  Row chunks are any multiple
// of 32 rows: no round quantisation.  (The first build put both sides on one wave per SIMD with the whole register file: hipcc
// spilled 360 registers in its loop; with the roles split, every accumulator and weight is live in exactly one loop and the
// kernel fits 254 registers without a spill.)
//
// Per 32-row step and wave (r = 96): a_A = bu + Wu z_a, a_G = bgu + Wgu z_g (12 MFMAs), elementwise dh, dq -> LDS tiles,
// p2 = Wd^T dpre_a, p1 = Wgd^T dpre_g (12), dWd += dpre_a^T x2, dWgd += dpre_g^T x1 (12), dx2 = s2*dh + p2 and
// dx1 = p1 (+ dx1_in) stored as 64 bytes per row, dWu += z_a^T dh, dWgu += z_g^T dq (12) + the bias sums (ones-row MFMAs).
//
// Memory system: every global read is a global_load_lds into a ring of NSTG stages (stage = 32 rows: three [32 x 256 B]
// row tiles as two 128-byte-wide pair tiles each + four [32 x 64*RT B] bottleneck tiles), two steps ahead, counted vmcnt,
// one barrier per step; the contraction over rows takes its operands with ds_read_b64_tr_b16 from the row-major tiles
// (trread.h; the 128-byte rows are XOR-swizzled on the source side as in wgrad_stream_kernel), the projections read their B
// fragments (lane = row) with ds_read_b128.  All LDS accesses of the loop are inline asm (see trread.h for why).
// Partials go to the workspace in wgrad.hip's layout; wgrad_finalize_kernel sums the row chunks (deterministic).
#include "cols_common.h"
#include "cols_reduce.h"

template <int RT, bool HAS_IN = false> struct ColzGeo {
    static constexpr int KT = 2 * RT;
    static constexpr int PB = 64 * RT;                  // bytes of a bottleneck row
    static constexpr int PT_B = 32 * PB;                // one bottleneck tile
    static constexpr int NX = HAS_IN ? 4 : 3;           // row tensors of a stage: dy, x2, x1 (+ the incoming dx1)
    static constexpr int X_B = NX * 2 * 4096;           // two pair tiles [32 rows x 128 B] each
    static constexpr int STG_B = X_B + 4 * PT_B;        // + z_a, z_g, dpre_a, dpre_g
    static constexpr int DQ_B = 2 * 4096;
    static constexpr int BIAS_B = 2 * 128 * 4;
    static constexpr size_t lds(int nstg) { return (size_t)nstg * STG_B + 2 * 8192 + DQ_B + BIAS_B; }    // ring, dh (x2), dq, biases
};

#ifndef VLPET_COLS_GRP
#define VLPET_COLS_GRP 6
#endif
#ifndef VLPET_COLS_WGBOTH
#define VLPET_COLS_WGBOTH 1
#endif
#ifndef VLPET_COLS_EWAHEAD
#define VLPET_COLS_EWAHEAD 1
#endif

template <int RT, int NSTG, bool ADD, bool HAS_IN>
__global__ __launch_bounds__(512, 2) void k1_cols_kernel(ColzArgs a) {
    using GEO = ColzGeo<RT, HAS_IN>;
    constexpr int KT = GEO::KT, PB = GEO::PB, PT_B = GEO::PT_B, X_B = GEO::X_B, STG_B = GEO::STG_B, NX = GEO::NX;
    constexpr int DH_OFF = NSTG * STG_B, DQ_OFF = DH_OFF + 2 * 8192, BIAS_OFF = DQ_OFF + GEO::DQ_B;
    constexpr int PR = 32 * RT;
    constexpr int NW = NX + RT;                         // global_load_lds instructions per wave and stage
    // LDS round trips are what a step's instruction stream waits for (a wave alone covers none of them), so operands are
    // requested in batches as large as the registers allow and each batch is waited for once:
    constexpr int GRP = VLPET_COLS_GRP < KT ? (KT % VLPET_COLS_GRP == 0 ? VLPET_COLS_GRP : KT) : KT;   // B fragments per batch of a projection
    constexpr bool WG_BOTH = VLPET_COLS_WGBOTH != 0;    // both 16-row k-steps of a weight-gradient job in one batch
    constexpr bool EW_AHEAD = VLPET_COLS_EWAHEAD != 0;  // all dy / x2 pieces of the elementwise stage requested at once
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    // ---- which (column block, row chunk): the column blocks of a row chunk share an XCD (they re-read the same bottleneck rows)
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
    bf16x8 wA[KT], wG[KT];                              // resident weights (loaded below, after the first stage requests)
    // ---- the stage pieces (1 KiB each) of this wave: pieces q = wave, wave + 8, wave + 16 are 8 rows of a pair tile of a row
    // tensor (tensor q / 8, pair (q / 4) % 2, rows 8 (q % 4) ..), pieces q' = wave + 8 j < 8 RT belong to the bottleneck tiles
    // (tensor q' / KT, piece q' % KT of the 32 contiguous rows)
    const uint8_t* xbase[NX]; uint32_t xdst[NX];
    const int xrow = 8 * (wave & 3) + (lane >> 3);
    const uint32_t xoff = (uint32_t)xrow * (uint32_t)ld2
                        + (uint32_t)((128 * cb + 64 * (wave >> 2)) * 2 + (((lane & 7) ^ fsw(xrow)) * 16));
#pragma unroll
    for (int t = 0; t < NX; ++t) {
        xbase[t] = reinterpret_cast<const uint8_t*>(t == 0 ? a.dy : t == 1 ? a.x2 : t == 2 ? a.x1 : a.dxin);
        xdst[t] = (uint32_t)((t * 2 + (wave >> 2)) * 4096 + (wave & 3) * 1024);
    }
    const uint8_t* pbase[RT]; uint32_t pdst[RT], poff[RT]; int prow[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
        const int q = wave + 8 * j, t = q / KT, piece = q % KT;
        const int sig = piece * 64 + lane;                                // LDS slot of this lane inside the tile
        prow[j] = sig / (PB / 16);
        pbase[j] = reinterpret_cast<const uint8_t*>(t == 0 ? a.z_a : t == 1 ? a.z_g : t == 2 ? a.dp_a : a.dp_g);
        poff[j] = (uint32_t)(prow[j] * PB + ((sig % (PB / 16)) ^ gsw(prow[j])) * 16);
        pdst[j] = (uint32_t)(X_B + t * PT_B + piece * 1024);
    }
    // a wave-uniform pointer the compiler must treat as a fresh scalar: keeps the per-lane part of an address a 32-bit loop
    // invariant (otherwise hipcc hoists one 64-bit per-lane address per stream out of the loop and spills them)
    auto sbase = [](const uint8_t* p) {
        const uint64_t u = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const uint8_t*>(((uint64_t)hi << 32) | lo);
    };
    auto issue = [&](int s) {
        const int64_t rb = r_begin + 32 * (int64_t)s;
        uint8_t* st = smem + (size_t)(s % NSTG) * STG_B;
        if (rb + 32 <= r_end) {
#pragma unroll
            for (int t = 0; t < NX; ++t) glds16_row(sbase(xbase[t] + rb * ld2) + xoff, st + xdst[t]);
#pragma unroll
            for (int j = 0; j < RT; ++j) glds16(sbase(pbase[j] + rb * PB) + poff[j], st + pdst[j]);
        } else {                                        // last step of the chunk: rows past the end re-read the last row
            const int last = (int)(r_end - rb) - 1;
            const uint32_t xo = xoff - (uint32_t)(xrow > last ? xrow - last : 0) * (uint32_t)ld2;
#pragma unroll
            for (int t = 0; t < NX; ++t) glds16_row(sbase(xbase[t] + rb * ld2) + xo, st + xdst[t]);
#pragma unroll
            for (int j = 0; j < RT; ++j)
                glds16(sbase(pbase[j] + rb * PB) + poff[j] - (uint32_t)(prow[j] > last ? prow[j] - last : 0) * PB, st + pdst[j]);
        }
    };

    // ---- per-lane LDS byte addresses (relative to the stage base)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    uint32_t a_xtr[2], a_ptr[2], a_xcl[2], a_pbf[2];
    {
        const int g4 = lane >> 4, sl = lane & 15;
        const int trow = 8 * (g4 >> 1) + (sl >> 2);                       // first row of this lane's transpose reads (second: + 4)
        const int tslot = 2 * (g4 & 1) + ((sl & 3) >> 1), thalf = 8 * (sl & 1);      // 8 bytes at slot 4 nt / 4 ct + tslot
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int r = trow + 4 * hi;
            a_xtr[hi] = (uint32_t)(pp * 4096 + r * 128 + (((4 * nt + tslot) ^ fsw(r)) * 16) + thalf);   // row tiles
            a_ptr[hi] = (uint32_t)(X_B + r * PB + ((tslot ^ gsw(r)) * 16) + thalf);                      // bottleneck tiles (+ 64 ct)
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            a_xcl[k] = (uint32_t)(pp * 4096 + m * 128 + (((4 * nt + 2 * h + k) ^ fsw(m)) * 16));        // columns 8k .. 8k+7 of the lane's 16 (row m)
            a_pbf[k] = (uint32_t)(X_B + m * PB + (((2 * k + h) ^ gsw(m)) * 16));                         // B fragment of row m, k-step 2j + k (+ 64 j)
        }
    }
    auto ones_row = [&](int k) {          // A fragment whose row (slot k) is all ones: D[row][n] = column sums of the B operand
        int mm = m;
        asm volatile("" : "+v"(mm));      // (rebuilt at every use: as loop invariants the fragments cost four registers per slot)
        const uint32_t w = (mm == (k & 3) + 8 * (k >> 2)) ? 0x3f803f80u : 0u;
        const u32x4 v = {w, w, w, w};
        return __builtin_bit_cast(bf16x8, v);
    };
    const int RC = a.row_chunks;
    const int col = c0 + m;
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using X2O = std::integral_constant<int, 8192>; using X1O = std::integral_constant<int, 16384>;

    // this row chunk's accumulators: role U: dWu, dWgu;  role D: dWd, dWgd  ([32*RT x 32 columns] each), and the column sums by
    // ones-row MFMAs (slot k = MFMA row (k & 3) + 8 (k >> 2) = register k of the lanes h = 0): U: slots 0 / 1 = dh / dq;
    // slots 2 + ct / 2 + RT + ct = dpre_a / dpre_g (one U wave per row chunk)
    f32x16 accA[RT], accG[RT], sx = zero16();
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) { accA[ct] = zero16(); accG[ct] = zero16(); }

    // ---- the step frame shared by the roles: wait for the own pieces of stage s, barrier, request stage s + NSTG - 1, zero the
    // bottleneck rows past the end in the last step.  `extra` = this wave's OTHER vector-memory operations younger than its
    // requests of stage s (role D: output stores and dx1_in loads).
    auto step_top = [&](int s, int extra) {
        int ahead = nsteps - 1 - s;                                       // stages already requested beyond s
        if (ahead > NSTG - 2) ahead = NSTG - 2;
        vm_wait(ahead * NW + extra);
        __builtin_amdgcn_s_barrier();                                     // stage s has landed for every wave; stage s - 1 is free
        if (s + NSTG - 1 < nsteps) issue(s + NSTG - 1);
        const int valid = (int)(r_end - (r_begin + 32 * (int64_t)s));
        if (valid < 32) {                                                 // zero the bottleneck rows past the end (their products must vanish)
            const u32x4 z = {0u, 0u, 0u, 0u};
            uint8_t* pt = smem + (size_t)(s % NSTG) * STG_B + X_B;
            for (int q = tid; q < 4 * 32 * (PB / 16); q += 512) {
                const int rr = (q / (PB / 16)) & 31;
                if (rr >= valid) *reinterpret_cast<u32x4*>(pt + (size_t)q * 16) = z;
            }
            __syncthreads();
        }
    };
    // projection: acc += W (resident A fragments) . B fragments of bottleneck tile T (lane = row), GRP k-steps per batch
    auto project = [&](uint32_t sb, auto TC, const bf16x8* w, f32x16& acc) {
        constexpr int T = decltype(TC)::value;
        sfor<KT / GRP>([&](auto G) {
            u32x4 bf[GRP];
            sfor<GRP>([&](auto K) {
                constexpr int ks = G.value * GRP + K.value;
                lds_read16<T * PT_B + 64 * (ks >> 1)>(bf[K.value], sb + a_pbf[ks & 1]);
            });
            lgkm_fence(bf[0]);
#pragma unroll
            for (int k = 0; k < GRP; ++k) { if (k) lgkm_tie(bf[k]); acc = mfma32(w[G.value * GRP + k], as_bf(bf[k]), acc); }
        });
    };
    // weight-gradient products of one job: acc[ct] += P^T (tile TP) . X (row tile at xlo / xhi + XO); one 16-row k-step at a
    // time (4 operands = 16 registers in flight)
    auto wg_products = [&](uint32_t sb, auto TPC, auto XOC, uint32_t xlo, uint32_t xhi, f32x16* acc, int slot) {
        constexpr int TP = decltype(TPC)::value, XO = decltype(XOC)::value;
        if constexpr (WG_BOTH) {
            TrOp bx[2], ap[2][RT];
            sfor<2>([&](auto KS) {
                constexpr int ks = KS.value;
                tr_read2<XO + ks * 16 * 128>(bx[ks], xlo, xhi);
                sfor<RT>([&](auto CT) { tr_read2<TP * PT_B + 64 * CT.value + ks * 16 * PB>(ap[ks][CT.value], sb + a_ptr[0], sb + a_ptr[1]); });
            });
            tr_fence(bx[0]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks) tr_tie(bx[ks]);
                const bf16x8 vx = tr_val(bx[ks]);
                if (slot >= 0) sx = mfma32(ones_row(slot), vx, sx);
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) { tr_tie(ap[ks][ct]); acc[ct] = mfma32(tr_val(ap[ks][ct]), vx, acc[ct]); }
            }
        } else {
            sfor<2>([&](auto KS) {
                constexpr int ks = KS.value;
                TrOp bx, ap[RT];
                tr_read2<XO + ks * 16 * 128>(bx, xlo, xhi);
                sfor<RT>([&](auto CT) { tr_read2<TP * PT_B + 64 * CT.value + ks * 16 * PB>(ap[CT.value], sb + a_ptr[0], sb + a_ptr[1]); });
                tr_fence(bx);
                const bf16x8 vx = tr_val(bx);
                if (slot >= 0) sx = mfma32(ones_row(slot), vx, sx);
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) { tr_tie(ap[ct]); acc[ct] = mfma32(tr_val(ap[ct]), vx, acc[ct]); }
            });
        }
    };

    // The first stages' requests go out BEFORE the wave fetches its resident weights and the biases (round 5; they used to follow them:
    // two memory round trips in a row at the head of every launch).  One wait covers both.
#pragma unroll
    for (int s0 = 0; s0 < NSTG - 1; ++s0)
        if (s0 < nsteps) issue(s0);
    // up-side biases of the workgroup's 128 columns -> LDS (fp32)
    if (tid < 256) {
        float* sbias = reinterpret_cast<float*>(smem + BIAS_OFF);
        const uint8_t* pk = tid < 128 ? a.pk_a : a.pk_g;
        sbias[tid] = reinterpret_cast<const float*>(pk + pg.bias_off)[PR + 128 * cb + (tid & 127)];
    }
    // resident weights: A fragments of this wave's 32 columns.  Role U: Wu / Wgu from the "up" packs (slot = W[f][16ks + 8hh + j]);
    // role D: Wd / Wgd transposed from the "down_t" packs (slot = W[16ks + 8hh + j][f]).  MFMA row i stands for column c0 +
    // 16*((i>>2)&1) + 4*(i>>3) + (i&3), so that a lane (m, h) ends with the 16 CONTIGUOUS columns c0 + 16h .. +15 of row m; in the
    // packs' own numbering (tests/packing_spec.py f_of4) that row is lane (i&3) | (nt << 2) | (i>>3 << 3) of n-tile v = (i>>2)&1.
    {
        const int i = m, v = (i >> 2) & 1, ip = (i & 3) | (nt << 2) | ((i >> 3) << 3);
        const int64_t off = (int64_t)(role == 0 ? 1 : 3) * pg.pack_bytes + (int64_t)(2 * cb + pp) * (4 * RT * 1024)
                          + (int64_t)(v * KT) * 1024 + (ip + 32 * h) * 16;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            wA[ks] = *reinterpret_cast<const bf16x8*>(a.pk_a + off + ks * 1024);
            wG[ks] = *reinterpret_cast<const bf16x8*>(a.pk_g + off + ks * 1024);
        }
    }

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // weights in registers, biases in LDS (and the first stages landed)
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) { asm volatile("" : "+v"(wA[ks])); asm volatile("" : "+v"(wG[ks])); }   // (hipcc's own guard of these loads lands here, not in the first step)
#ifdef VLPET_COLS_STAMPS      // diagnosis build: wall-clock time per phase, summed over the steps, printed by one wave of each side
    uint64_t tacc[5] = {0, 0, 0, 0, 0}, tlast = wall_clock64();
#define COLS_STAMP(k) { const uint64_t tn = wall_clock64(); tacc[k] += tn - tlast; tlast = tn; }
#else
#define COLS_STAMP(k)
#endif

    // The only hand-off between the roles is the dh tile (U -> D, for dx2 = s2*dh + ..).  It is double-buffered OUTSIDE the
    // ring and consumed one step late: D finishes the input gradients of step s - 1 at the start of step s, so the one barrier
    // of a step (the stage hand-over) also orders that hand-off and the two roles never wait for each other inside a step.
    if (role == 0) {
        // ================================================================ role U: up projections, dh / dq, dWu, dWgu, all bias sums
        const float s2 = a.s2, sd = a.sd;
        // the down-side bias sums (column sums of the 2 RT bottleneck tiles dpre_a / dpre_g, needed once per row chunk): tile k goes
        // to the up-side wave number w = wc * NCB + cb of the row chunk's workgroups with k = w (mod 4 NCB) -- at d = 768 one tile
        // for wave 0 of each of the six column blocks (all twelve products on ONE wave made its workgroup the last to finish)
        int csp_tile[2] = {-1, -1};
        {
            const int w = wc * NCB + cb;
            if (w < 2 * RT) csp_tile[0] = w;
            if (w + 4 * NCB < 2 * RT) csp_tile[1] = w + 4 * NCB;
        }
        const bool want_csp = csp_tile[0] >= 0;
        const uint32_t a_bias = lds0 + (uint32_t)(BIAS_OFF + (32 * wc + 16 * h) * 4);
        const uint32_t dq0 = lds0 + (uint32_t)DQ_OFF;
#pragma unroll 1
        for (int s = 0; s < nsteps; ++s) {
            const int valid = (int)(r_end - (r_begin + 32 * (int64_t)s));
            const uint32_t sb = lds0 + (uint32_t)((s % NSTG) * STG_B);
            const uint32_t dh0 = lds0 + (uint32_t)(DH_OFF + (s & 1) * 8192);
            COLS_STAMP(4)
            step_top(s, 0);
            COLS_STAMP(0)
#if defined(VLPET_COLS_ABL) && (VLPET_COLS_ABL & 8)
            continue;                                                     // (timing ablation: the LDS-DMA stream + barriers alone)
#endif
            // up projection of one chain, starting at its bias: the bias values and the first batch of B fragments are one LDS batch
            auto project_up = [&](auto TC, auto OC, const bf16x8* w, f32x16& acc) {
                constexpr int T = decltype(TC)::value;
                u32x4 bb[4], bf0[GRP];
                sfor<4>([&](auto Q) { lds_read16<decltype(OC)::value + 16 * Q.value>(bb[Q.value], a_bias); });
                sfor<GRP>([&](auto K) { lds_read16<T * PT_B + 64 * (K.value >> 1)>(bf0[K.value], sb + a_pbf[K.value & 1]); });
                lgkm_fence(bb[0]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q) lgkm_tie(bb[q]);
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) acc[4 * q + w2] = __uint_as_float(bb[q][w2]);
                }
#pragma unroll
                for (int k = 0; k < GRP; ++k) { lgkm_tie(bf0[k]); acc = mfma32(w[k], as_bf(bf0[k]), acc); }
                sfor<KT / GRP - 1>([&](auto G) {
                    u32x4 bf[GRP];
                    sfor<GRP>([&](auto K) {
                        constexpr int ks = (G.value + 1) * GRP + K.value;
                        lds_read16<T * PT_B + 64 * (ks >> 1)>(bf[K.value], sb + a_pbf[ks & 1]);
                    });
                    lgkm_fence(bf[0]);
#pragma unroll
                    for (int k = 0; k < GRP; ++k) { if (k) lgkm_tie(bf[k]); acc = mfma32(w[(G.value + 1) * GRP + k], as_bf(bf[k]), acc); }
                });
            };
            {
                f32x16 aA, aG;                                            // both up projections
                project_up(I0{}, I0{}, wA, aA);
                project_up(I1{}, std::integral_constant<int, 512>{}, wG, aG);
#ifdef VLPET_COLS_STAMPS
                asm volatile("s_nop 0" : "+v"(aA[15]), "+v"(aG[15]));
#endif
                COLS_STAMP(1)
                const float gsr = m < valid ? a.gs : 0.f;                 // rows past the end: dh = dq = 0
                u32x2 dyv[4], x2v[4];
                if constexpr (EW_AHEAD) {                                 // (requested while the projections' MFMAs run)
                    sfor<4>([&](auto C) {
                        lds_read8<8 * (C.value & 1)>(dyv[C.value], sb + a_xcl[C.value >> 1]);
                        lds_read8<8192 + 8 * (C.value & 1)>(x2v[C.value], sb + a_xcl[C.value >> 1]);
                    });
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dyv[0]), "+v"(x2v[0]), "+v"(dyv[1]), "+v"(x2v[1]), "+v"(dyv[2]), "+v"(x2v[2]), "+v"(dyv[3]), "+v"(x2v[3]) :: "memory");
                }
#if defined(VLPET_COLS_ABL) && (VLPET_COLS_ABL & 4)
                asm volatile("" :: "v"(aA[0]), "v"(aG[0]), "v"(dyv[0]), "v"(x2v[0]));
                if (false)
#endif
                sfor<4>([&](auto C) {                                     // 4 columns at a time (a small live set): dy, x2 in, dh, dq out
                    constexpr int c = C.value;
                    if constexpr (!EW_AHEAD) {
                        lds_read8<8 * (c & 1)>(dyv[c], sb + a_xcl[c >> 1]);
                        lds_read8<8192 + 8 * (c & 1)>(x2v[c], sb + a_xcl[c >> 1]);
                    }
                    // (the statement also pins this chunk's share of the projections behind the previous chunk's stores: hipcc would
                    //  otherwise start all 16 sigmoids at once and keep their temporaries live)
                    if constexpr (EW_AHEAD) asm volatile("" : "+v"(dyv[c]), "+v"(x2v[c]), "+v"(aG[4 * c]), "+v"(aG[4 * c + 1]), "+v"(aG[4 * c + 2]), "+v"(aG[4 * c + 3]) :: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dyv[c]), "+v"(x2v[c]), "+v"(aG[4 * c]), "+v"(aG[4 * c + 1]), "+v"(aG[4 * c + 2]), "+v"(aG[4 * c + 3]) :: "memory");
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
            }
            // this wave's own columns of dh, dq: its writes above are ordered before these reads by the waits in between
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            COLS_STAMP(2)
#if !(defined(VLPET_COLS_ABL) && (VLPET_COLS_ABL & 1))
            wg_products(sb, I0{}, I0{}, dh0 + a_xtr[0], dh0 + a_xtr[1], accA, 0);
            wg_products(sb, I1{}, I0{}, dq0 + a_xtr[0], dq0 + a_xtr[1], accG, 1);
#endif
            if (want_csp) {                                               // its own block (inside the products it would make every accumulator a phi)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = csp_tile[j];
                    if (k < 0) break;
                    const uint32_t off = (uint32_t)((2 + k / RT) * PT_B + 64 * (k % RT));    // tile dpre_a / dpre_g, c-tile k % RT
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
#ifdef VLPET_COLS_STAMPS
            asm volatile("s_nop 0" : "+v"(accA[RT - 1][15]), "+v"(accG[RT - 1][15]));
#endif
            COLS_STAMP(3)
        }
#ifdef VLPET_COLS_STAMPS
        if (lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 101))
            printf("cols stamps U blk %d (10 ns units, %d steps): top %llu up %llu ew %llu wgrad %llu other %llu\n", (int)blockIdx.x, nsteps,
                   (unsigned long long)tacc[0], (unsigned long long)tacc[1], (unsigned long long)tacc[2], (unsigned long long)tacc[3], (unsigned long long)tacc[4]);
#endif
        __builtin_amdgcn_s_barrier();                                     // the last dh tile is visible to role D
        if (a.red.slab != nullptr) {                                      // (in-launch reduce: write-through partials, cols_reduce.h)
            if (h == 0) {
                cols_red_store_f32(a.red.bias_x + (int64_t)rc * d + col, sx[0]);
                cols_red_store_f32(a.red.bias_x + ((int64_t)RC + rc) * d + col, sx[1]);
                if (want_csp) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int k = csp_tile[j];
                        if (k < 0) break;
                        cols_red_store_f32(a.red.bias_p + (int64_t)rc * (2 * PR) + 32 * k + m, sx[2 + j]);
                    }
                }
            }
        } else
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
        f32x16 p2 = zero16(), p1 = zero16();            // Wd^T dpre_a, Wgd^T dpre_g of the PREVIOUS step's rows
        // Order of this wave's vector-memory operations around step s: ... S(s-2) | [wait for G(s)] G(s+NSTG-1) S(s-1) ...
        // (G = its stage requests, S = the four output stores of a step's rows).  The incoming dx1 rows travel with the stage (a
        // fourth row tensor) and wait in registers for a step: register loads next to the LDS-DMA queue are not an option -- hidden
        // in asm, hipcc copies their destination registers before the data lands; visible, it drains the queue for them.
        u32x4 dinA = {0u, 0u, 0u, 0u}, dinB = {0u, 0u, 0u, 0u};           // dx1_in of the previous step's rows (this lane's 16 columns)
        auto finish = [&](int sp, const u32x4& din0, const u32x4& din1) {   // input gradients of step sp (its dh tile is complete)
            const int64_t rb = r_begin + 32 * (int64_t)sp;
            const int valid = (int)(r_end - rb) < 32 ? (int)(r_end - rb) : 32;
            const bool row_ok = m < valid;
            // byte offset of this lane's 16 columns of its row, relative to row rb (rows past the end: the last row)
            const uint32_t rowoff = (uint32_t)(row_ok ? m : valid - 1) * (uint32_t)ld2 + (uint32_t)((c0 + 16 * h) * 2);
            const uint32_t dh0 = lds0 + (uint32_t)(DH_OFF + (sp & 1) * 8192);
            u32x4 dhv0, dhv1;
            lds_read16<0>(dhv0, dh0 + a_xcl[0]); lds_read16<0>(dhv1, dh0 + a_xcl[1]);
            lgkm_fence(dhv0); lgkm_tie(dhv1);
            // (round 6: both tensors through a wave-private LDS tile, stored as 64-byte runs of 16 rows per instruction like the forward's pass B --
            //  built, bit-identical, no gain alone and ~2 us per launch slower in the step: profiles/r06_k1_cols_staged_stores_ab.txt; removed)
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
                if constexpr (HAS_IN) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[e] = p1[e] + bf_at(e < 8 ? din0 : din1, e & 7);
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[e] = p1[e];
                }
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
            const uint32_t sb = lds0 + (uint32_t)((s % NSTG) * STG_B);
            COLS_STAMP(4)
            step_top(s, s >= 2 ? 4 : 0);
            COLS_STAMP(0)
#if defined(VLPET_COLS_ABL) && (VLPET_COLS_ABL & 8)
            continue;
#endif
#if !(defined(VLPET_COLS_ABL) && (VLPET_COLS_ABL & 2))
            if (s > 0) finish(s - 1, dinA, dinB);
#endif
            COLS_STAMP(1)
            if constexpr (HAS_IN) {
                lds_read16<3 * 8192>(dinA, sb + a_xcl[0]); lds_read16<3 * 8192>(dinB, sb + a_xcl[1]);
                lgkm_fence(dinA); lgkm_tie(dinB);
            }
            wg_products(sb, I2{}, X2O{}, sb + a_xtr[0], sb + a_xtr[1], accA, -1);
            wg_products(sb, I3{}, X1O{}, sb + a_xtr[0], sb + a_xtr[1], accG, -1);
#ifdef VLPET_COLS_STAMPS
            asm volatile("s_nop 0" : "+v"(accA[RT - 1][15]), "+v"(accG[RT - 1][15]));
#endif
            COLS_STAMP(2)
            p2 = zero16(); p1 = zero16();
            project(sb, I2{}, wA, p2);
            project(sb, I3{}, wG, p1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef VLPET_COLS_STAMPS
            asm volatile("s_nop 0" : "+v"(p1[15]), "+v"(p2[15]));
#endif
            COLS_STAMP(3)
        }
#ifdef VLPET_COLS_STAMPS
        if (lane == 0 && wave == 4 && (blockIdx.x == 0 || blockIdx.x == 101))
            printf("cols stamps D blk %d (10 ns units, %d steps): top %llu finish %llu wgrad %llu project %llu other %llu\n", (int)blockIdx.x, nsteps,
                   (unsigned long long)tacc[0], (unsigned long long)tacc[1], (unsigned long long)tacc[2], (unsigned long long)tacc[3], (unsigned long long)tacc[4]);
#endif
        __builtin_amdgcn_s_barrier();                                     // role U has written the last dh tile
        if (nsteps > 0) finish(nsteps - 1, dinA, dinB);
    }
    // ---- this row chunk's partial sums: summed over the row chunks INSIDE this launch by the workgroups of the column block (round 6,
    // cols_reduce.h) ...
    if (a.red.slab != nullptr) {
        using RG = ColsRedGeo<RT, 2>;
        const __amdgpu_buffer_rsrc_t mine = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(sbase(reinterpret_cast<const uint8_t*>(a.red.slab) + ((int64_t)rc * NCB + cb) * RG::SLAB_B)), 0, RG::SLAB_B, 0x00020000);
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) {
            cols_red_put<RT, 2>(mine, wave, lane, 0, ct, accA[ct]);
            cols_red_put<RT, 2>(mine, wave, lane, 1, ct, accG[ct]);
        }
        cols_red_finish<RT, 2, 2, 2 * RT>(a.red, NCB, RC, cb, rc, d, reinterpret_cast<volatile unsigned*>(smem),
            [](int rl, int jb) { return rl == 0 ? (jb == 0 ? 1 : 3) : (jb == 0 ? 0 : 2); },         // U: dWu, dWgu;  D: dWd, dWgd
            [](int x) { return x == 0 ? 1 : 3; },                                                    // column sums of dh -> dbu, of dq -> dbgu
            [](int k, int& job, int& first) { job = k / RT == 0 ? 0 : 2; first = 32 * (k % RT); });  // dpre_a tiles -> dbd, dpre_g tiles -> dbgd
        return;
    }
    // ... or, without the reduce-scatter arguments, in wgrad.hip's workspace layout for wgrad_finalize_kernel (the round-3 form, kept
    // for same-box A/Bs: api.hip run_bwd, `phases` bit 5)
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

// Row chunks: (d / 128) column blocks x chunks workgroups, at most 32 per XCD (one per CU: the ring takes the LDS), i.e. at most
// 8 * floor(32 / NCB) chunks; a chunk is a multiple of 32 rows.
void k1_cols_plan(int64_t M, int d, int* row_chunks, int64_t* rows_per_chunk) {
    const int ncb = d >= 128 ? d / 128 : 1;             // (d < 128: the form does not apply; the plan only sizes workspaces)
    int64_t rc = cols_groups_max(ncb < 32 ? ncb : 32);
    const int64_t blocks32 = (M + 31) / 32;
    if (rc > blocks32) rc = blocks32;
    if (rc < 1) rc = 1;
    const int64_t per = (blocks32 + rc - 1) / rc;
    rc = (blocks32 + per - 1) / per;
    *row_chunks = (int)rc;
    *rows_per_chunk = per * 32;
}

bool k1_cols_applies(const PetBwdArgs& a, int io_fp32) {
    return !io_fp32 && (a.flags & PET_GATE) && a.saved != nullptr && !drop_active(a.drop) && (a.RT == 1 || a.RT == 3) &&
           a.d % 128 == 0 && a.d / 128 <= 32;
}

template <int RT, bool ADD, bool HAS_IN>
static hipError_t launch_cols_cfg(const ColzArgs& c, hipStream_t stream) {
    constexpr int NSTG = RT <= 1 ? 3 : 2;               // (r = 96: two 48-KiB stages + the dh / dq tiles = 121 KiB)
    const size_t lds = ColzGeo<RT, HAS_IN>::lds(NSTG);
    auto kern = k1_cols_kernel<RT, NSTG, ADD, HAS_IN>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int ncb = c.d / 128;
    const unsigned grid = cols_grid(ncb, c.row_chunks);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, c);
    return hipGetLastError();
}
template <int RT>
static hipError_t launch_cols_rt(const ColzArgs& c, hipStream_t stream) {
    const bool add = (c.flags & PET_GATE_ADD) != 0, in = c.dxin != nullptr;
    if (add) return in ? launch_cols_cfg<RT, true, true>(c, stream) : launch_cols_cfg<RT, true, false>(c, stream);
    return in ? launch_cols_cfg<RT, false, true>(c, stream) : launch_cols_cfg<RT, false, false>(c, stream);
}

hipError_t launch_k1_cols(const ColzArgs& c, int RT, hipStream_t stream) {
    if (RT == 1) return launch_cols_rt<1>(c, stream);
    if (RT == 3) return launch_cols_rt<3>(c, stream);
    return hipErrorInvalidValue;
}
