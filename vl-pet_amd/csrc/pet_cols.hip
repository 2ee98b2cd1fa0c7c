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
// so a workgroup owns 128 columns and a chunk of rows; each of its four waves owns 32 columns (one MFMA tile), keeps ITS
// slice of the four weights (4 x 32 x r: 96 registers at r = 96) and its [r x 32] slices of the four weight gradients
// (192 accumulator registers) in registers for the whole launch -- one wave per SIMD, the whole 512-entry register file --
// and streams rows 32 at a time: reads dy, x1, x2 once, writes dx1, dx2 once (5 units = the op's algorithmic traffic), plus
// the four [M, 32*RT] bottleneck tensors (z_a, z_g, dpre_a, dpre_g), which the d/128 column blocks of a row chunk re-read
// through the L2 of one XCD.  Row chunks are any multiple of 32 rows: no round quantisation.  (A two-roles-per-SIMD form --
// up side and down side on two waves of 256 registers -- was built first: 48 weights + 96 accumulators + 16 column sums per
// wave leave hipcc 60-80 registers short in the loop, and a spill there is a scratch access on the vmcnt queue that the
// counted waits below do not know about.)
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
#include <type_traits>
#include <utility>
#include "pet16.h"
#include "kernels.h"
#include "trread.h"

template <int RT> struct ColzGeo {
    static constexpr int KT = 2 * RT;
    static constexpr int PB = 64 * RT;                  // bytes of a bottleneck row
    static constexpr int PT_B = 32 * PB;                // one bottleneck tile
    static constexpr int X_B = 6 * 4096;                // dy, x2, x1: two pair tiles [32 rows x 128 B] each
    static constexpr int STG_B = X_B + 4 * PT_B;        // + z_a, z_g, dpre_a, dpre_g
    static constexpr int NI = 6 + 2 * RT;               // global_load_lds instructions per wave and stage
    static constexpr int DQ_B = 2 * 4096;
    static constexpr int BIAS_B = 2 * 128 * 4;
    static constexpr size_t lds(int nstg) { return (size_t)nstg * STG_B + DQ_B + BIAS_B; }
};

template <typename F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

template <int OFF> __device__ __forceinline__ void lds_read16(u32x4& o, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(o) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_write16(uint32_t addr, const u32x4& v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lgkm_fence(u32x4& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) :: "memory"); }
__device__ __forceinline__ void lgkm_tie(u32x4& a) { asm volatile("" : "+v"(a) :: "memory"); }
__device__ __forceinline__ bf16x8 as_bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// at most n vector-memory operations of this wave still in flight (n is wave-uniform)
__device__ __forceinline__ void vm_wait(int n) {
#define VLPET_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        VLPET_VMW(0) VLPET_VMW(1) VLPET_VMW(2) VLPET_VMW(3) VLPET_VMW(4) VLPET_VMW(5) VLPET_VMW(6) VLPET_VMW(7)
        VLPET_VMW(8) VLPET_VMW(9) VLPET_VMW(10) VLPET_VMW(11) VLPET_VMW(12) VLPET_VMW(13) VLPET_VMW(14) VLPET_VMW(15)
        VLPET_VMW(16) VLPET_VMW(17) VLPET_VMW(18) VLPET_VMW(19) VLPET_VMW(20) VLPET_VMW(21) VLPET_VMW(22) VLPET_VMW(23)
        VLPET_VMW(24) VLPET_VMW(25) VLPET_VMW(26) VLPET_VMW(27) VLPET_VMW(28) VLPET_VMW(29) VLPET_VMW(30)
        default: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;       // more than 30: stricter is safe
    }
#undef VLPET_VMW
}

__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bf_at(const u32x4& v, int j) { return (j & 1) ? bf_hi(v[j >> 1]) : bf_lo(v[j >> 1]); }
__device__ __forceinline__ u32x4 pack8(const float* v) {
    bf16x8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (__bf16)v[j];
    return __builtin_bit_cast(u32x4, t);
}
// sigmoid with one v_exp_f32 and one v_rcp_f32
__device__ __forceinline__ float sigm(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

template <int RT, int NSTG, bool ADD, bool HAS_IN>
__global__ __launch_bounds__(256, 1) void k1_cols_kernel(ColzArgs a) {
    using GEO = ColzGeo<RT>;
    constexpr int KT = GEO::KT, PB = GEO::PB, PT_B = GEO::PT_B, X_B = GEO::X_B, STG_B = GEO::STG_B, NI = GEO::NI;
    constexpr int DQ_OFF = NSTG * STG_B, BIAS_OFF = DQ_OFF + GEO::DQ_B;
    constexpr int PR = 32 * RT;
    constexpr int NL = HAS_IN ? 2 : 0;                  // register loads (dx1_in) per step
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    // ---- which (column block, row chunk): the column blocks of a row chunk share an XCD (they re-read the same bottleneck rows)
    const int d = a.d, NCB = d >> 7;
    const int bq = blockIdx.x >> 3;
    const int cb = bq % NCB;
    const int rc = (bq / NCB) * 8 + (blockIdx.x & 7);
    if (rc >= a.row_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wc = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pp = wc >> 1, nt = wc & 1;
    const int m = lane & 31, h = lane >> 5;
    const int64_t ld2 = (int64_t)d * 2;
    const int c0 = 128 * cb + 32 * wc;
    const int64_t r_begin = (int64_t)rc * a.rows_per_chunk;
    int64_t r_end = r_begin + a.rows_per_chunk;
    if (r_end > a.M) r_end = a.M;
    const int nsteps = r_end > r_begin ? (int)((r_end - r_begin + 31) >> 5) : 0;
    const uint8_t* DY = reinterpret_cast<const uint8_t*>(a.dy);
    const uint8_t* X1 = reinterpret_cast<const uint8_t*>(a.x1);
    const uint8_t* X2 = reinterpret_cast<const uint8_t*>(a.x2);
    const uint8_t* DXIN = reinterpret_cast<const uint8_t*>(a.dxin);

    // ---- resident weights: A fragments of this wave's 32 columns.  Wu / Wgu from the "up" packs (slot = W[f][16ks + 8hh + j]),
    // Wd / Wgd transposed from the "down_t" packs (slot = W[16ks + 8hh + j][f]).  MFMA row i stands for column c0 + 16*((i>>2)&1)
    // + 4*(i>>3) + (i&3), so that a lane (m, h) ends with the 16 CONTIGUOUS columns c0 + 16h .. +15 of row m; in the packs' own
    // numbering (tests/packing_spec.py f_of4) that row is lane (i&3) | (nt << 2) | (i>>3 << 3) of n-tile v = (i>>2)&1.
    bf16x8 wU[KT], wGU[KT], wD[KT], wGD[KT];
    {
        const PackGeom pg = pack_geom(RT, d, 1);
        const int i = m, v = (i >> 2) & 1, ip = (i & 3) | (nt << 2) | ((i >> 3) << 3);
        const int64_t off = (int64_t)(2 * cb + pp) * (4 * RT * 1024) + (int64_t)(v * KT) * 1024 + (ip + 32 * h) * 16;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            wU[ks] = *reinterpret_cast<const bf16x8*>(a.pk_a + pg.pack_bytes + off + ks * 1024);
            wGU[ks] = *reinterpret_cast<const bf16x8*>(a.pk_g + pg.pack_bytes + off + ks * 1024);
            wD[ks] = *reinterpret_cast<const bf16x8*>(a.pk_a + 3 * pg.pack_bytes + off + ks * 1024);
            wGD[ks] = *reinterpret_cast<const bf16x8*>(a.pk_g + 3 * pg.pack_bytes + off + ks * 1024);
        }
        // up-side biases of the workgroup's 128 columns -> LDS (fp32)
        float* sbias = reinterpret_cast<float*>(smem + BIAS_OFF);
        const uint8_t* pk = tid < 128 ? a.pk_a : a.pk_g;
        sbias[tid] = reinterpret_cast<const float*>(pk + pg.bias_off)[PR + 128 * cb + (tid & 127)];
    }

    // ---- per-lane source geometry of the stage pieces: wave w loads rows 8w .. 8w+7 of both pair tiles of the three row
    // tensors (6 instructions) and 2*RT of the 8*RT one-KiB pieces of the bottleneck tiles
    const int xrow = 8 * wc + (lane >> 3);
    const uint32_t xcol = (uint32_t)((128 * cb) * 2 + (((lane & 7) ^ (4 * ((xrow >> 1) & 1))) * 16));
    const uint32_t xoff = (uint32_t)xrow * (uint32_t)ld2 + xcol;          // (32 rows x d*2 bytes: far below 4 GiB)
    const uint32_t xdst = (uint32_t)(wc * 1024);
    const uint8_t* pbase[2 * RT]; uint32_t poff[2 * RT]; uint32_t pdst[2 * RT];
#pragma unroll
    for (int j = 0; j < 2 * RT; ++j) {
        const int q = wc + 4 * j, t = q / KT, piece = q % KT;            // wave-uniform
        pbase[j] = reinterpret_cast<const uint8_t*>(t == 0 ? a.z_a : t == 1 ? a.z_g : t == 2 ? a.dp_a : a.dp_g);
        poff[j] = (uint32_t)(piece * 1024 + lane * 16);
        pdst[j] = (uint32_t)(X_B + t * PT_B + piece * 1024);
    }
    // a wave-uniform pointer the compiler must treat as a fresh scalar: keeps "scalar base + 32-bit lane offset" addressing (with
    // the lane offsets visible as loop invariants hipcc hoists one 64-bit per-lane address per stream out of the loop -- 18
    // registers it then spills and reloads in front of every request)
    auto sbase = [](const uint8_t* p) {
        const uint64_t u = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const uint8_t*>(((uint64_t)hi << 32) | lo);
    };
    auto issue = [&](int s) {
        const int64_t rb = r_begin + 32 * (int64_t)s;
        uint8_t* st = smem + (size_t)(s % NSTG) * STG_B;
        if (rb + 32 <= r_end) {                                           // scalar base + 32-bit lane offset
            const int64_t ro = rb * ld2;
            const uint8_t* b0 = sbase(DY + ro); const uint8_t* b1 = sbase(X2 + ro); const uint8_t* b2 = sbase(X1 + ro);
            glds16_row(b0 + xoff, st + xdst);         glds16_row(b0 + 128 + xoff, st + xdst + 4096);
            glds16_row(b1 + xoff, st + xdst + 8192);  glds16_row(b1 + 128 + xoff, st + xdst + 12288);
            glds16_row(b2 + xoff, st + xdst + 16384); glds16_row(b2 + 128 + xoff, st + xdst + 20480);
#pragma unroll
            for (int j = 0; j < 2 * RT; ++j) glds16(sbase(pbase[j] + rb * PB) + poff[j], st + pdst[j]);
        } else {                                                          // last step of the chunk: rows past the end re-read the last row
            int64_t row = rb + xrow;
            if (row >= r_end) row = r_end - 1;
            const int64_t ro = row * ld2 + (int64_t)xcol;
            glds16_row(DY + ro, st + xdst);         glds16_row(DY + ro + 128, st + xdst + 4096);
            glds16_row(X2 + ro, st + xdst + 8192);  glds16_row(X2 + ro + 128, st + xdst + 12288);
            glds16_row(X1 + ro, st + xdst + 16384); glds16_row(X1 + ro + 128, st + xdst + 20480);
#pragma unroll
            for (int j = 0; j < 2 * RT; ++j) {
                int64_t prow = rb + (int)(poff[j] / PB);
                if (prow >= r_end) prow = r_end - 1;
                glds16(pbase[j] + prow * PB + (int)(poff[j] % PB), st + pdst[j]);
            }
        }
    };

    // ---- per-lane LDS byte addresses (relative to the stage base)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    uint32_t a_xtr, a_xcl, a_ptr, a_pbf;
    {
        const int g4 = lane >> 4, sl = lane & 15;
        const int trow = 8 * (g4 >> 1) + (sl >> 2), tbit = (trow >> 1) & 1, lp = 32 * (g4 & 1) + 8 * (sl & 3);
        a_xtr = (uint32_t)(pp * 4096 + trow * 128 + 64 * (nt ^ tbit) + lp);                      // transpose reads of a row tile
        a_ptr = (uint32_t)(X_B + trow * PB + lp);                                                //   ... of a bottleneck tile (+ 64 ct)
        a_xcl = (uint32_t)(pp * 4096 + m * 128 + (((4 * nt + 2 * h) ^ (4 * ((m >> 1) & 1))) * 16));  // the lane's 16 columns of row m (32 B)
        a_pbf = (uint32_t)(X_B + m * PB + 16 * h);                                               // B fragment of row m (+ 32 ks)
    }
    const uint32_t a_bias = lds0 + (uint32_t)(BIAS_OFF + (32 * wc + 16 * h) * 4);
    const uint32_t a_dqtr = lds0 + (uint32_t)DQ_OFF + a_xtr, a_dqcl = lds0 + (uint32_t)DQ_OFF + a_xcl;

    f32x16 accU[RT], accGU[RT], accD[RT], accGD[RT];   // this wave's [32*RT x 32 columns] slices of dWu, dWgu, dWd, dWgd
    // column sums by ones-row MFMAs: slot k lives in MFMA row (k & 3) + 8 * (k >> 2) = register k of the lanes h = 0.
    // slots 0 / 1: dh / dq;  slots 2 + ct / 2 + RT + ct: dpre_a / dpre_g (one wave per row chunk)
    f32x16 sx = zero16();
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) { accU[ct] = zero16(); accGU[ct] = zero16(); accD[ct] = zero16(); accGD[ct] = zero16(); }
    auto ones_row = [&](int k) {          // A fragment whose row (slot k) is all ones: D[row][n] = column sums of the B operand
        const uint32_t w = (m == (k & 3) + 8 * (k >> 2)) ? 0x3f803f80u : 0u;
        const u32x4 v = {w, w, w, w};
        return __builtin_bit_cast(bf16x8, v);
    };
    const bool want_csp = cb == 0 && wc == 0;            // wave-uniform: the down-side bias sums, once per row chunk
    const float s2 = a.s2, sd = a.sd;

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // weights in registers, biases in LDS
    if (nsteps > 0) issue(0);
    if (nsteps > 1) issue(1);

    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using X2O = std::integral_constant<int, 8192>; using X1O = std::integral_constant<int, 16384>;

#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        const int64_t rb = r_begin + 32 * (int64_t)s;
        const int valid = (int)(r_end - rb) < 32 ? (int)(r_end - rb) : 32;
        const bool tail = valid < 32;
        const bool has1 = s + 1 < nsteps, has2 = s + 2 < nsteps;
        const uint32_t sb = lds0 + (uint32_t)((s % NSTG) * STG_B);
        const bool row_ok = m < valid;
        // byte offset of this lane's 16 columns of its row, relative to row rb (rows past the end: the last row)
        const uint32_t rowoff = (uint32_t)(row_ok ? m : valid - 1) * (uint32_t)ld2 + (uint32_t)((c0 + 16 * h) * 2);
        // Order of this wave's vector-memory operations: ... G(s+1) S(s-1) | L(s) [wait for G(s)] G(s+2) ... [wait for L(s)] ... S(s)
        // (G = the stage requests, L = the dx1_in register loads, S = the four output stores), all counted by hand.
        u32x4 din0 = {0u, 0u, 0u, 0u}, din1 = {0u, 0u, 0u, 0u};
        if constexpr (HAS_IN) {                                           // the incoming dx1 rows of this step, straight to registers
            const uint8_t* bp = sbase(DXIN + rb * ld2);
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(din0) : "v"(rowoff), "s"(bp) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(din1) : "v"(rowoff), "s"(bp) : "memory");
        }
        vm_wait((has1 ? NI : 0) + (s > 0 ? 4 : 0) + NL);
        __builtin_amdgcn_s_barrier();                                     // stage s has landed for every wave; stage s - 1 is free
        if (has2) issue(s + 2);
        if (tail) {                                                       // zero the bottleneck rows past the end (their products must vanish)
            const u32x4 z = {0u, 0u, 0u, 0u};
            uint8_t* pt = smem + (size_t)(s % NSTG) * STG_B + X_B;
            for (int q = tid; q < 4 * 32 * (PB / 16); q += 256) {
                const int rr = (q / (PB / 16)) & 31;
                if (rr >= valid) *reinterpret_cast<u32x4*>(pt + (size_t)q * 16) = z;
            }
            __syncthreads();
        }

        // projection of one chain: acc += W (resident A fragments) . B fragments of bottleneck tile T (lane = row)
        auto project = [&](auto TC, const bf16x8* w, f32x16& acc) {
            constexpr int T = decltype(TC)::value;
            u32x4 bf[KT];
            sfor<KT>([&](auto K) { lds_read16<T * PT_B + 32 * K.value>(bf[K.value], sb + a_pbf); });
            lgkm_fence(bf[0]);
#pragma unroll
            for (int k = 0; k < KT; ++k) { if (k) lgkm_tie(bf[k]); acc = mfma32(w[k], as_bf(bf[k]), acc); }
        };
        auto load_bias = [&](auto OC, f32x16& acc) {
            u32x4 bb[4];
            sfor<4>([&](auto Q) { lds_read16<decltype(OC)::value + 16 * Q.value>(bb[Q.value], a_bias); });
            lgkm_fence(bb[0]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q) lgkm_tie(bb[q]);
#pragma unroll
                for (int w = 0; w < 4; ++w) acc[4 * q + w] = __uint_as_float(bb[q][w]);
            }
        };
        // weight-gradient products of one job: acc[ct] += P^T (tile TP) . X (row tile at xaddr + XO), both 16-row k-steps
        auto wg_products = [&](auto TPC, auto XOC, uint32_t xaddr, f32x16* acc, int srow) {
            constexpr int TP = decltype(TPC)::value, XO = decltype(XOC)::value;
            TrOp bx[2], ap[2][RT];
            sfor<2>([&](auto KS) {
                constexpr int ks = KS.value;
                tr_read<XO + ks * 16 * 128, XO + (ks * 16 + 4) * 128>(bx[ks], xaddr);
                sfor<RT>([&](auto CT) {
                    tr_read<TP * PT_B + 64 * CT.value + ks * 16 * PB, TP * PT_B + 64 * CT.value + (ks * 16 + 4) * PB>(ap[ks][CT.value], sb + a_ptr);
                });
            });
            tr_fence(bx[0]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks) tr_tie(bx[ks]);
                const bf16x8 vx = tr_val(bx[ks]);
                if (srow >= 0) sx = mfma32(ones_row(srow), vx, sx);
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) { tr_tie(ap[ks][ct]); acc[ct] = mfma32(tr_val(ap[ks][ct]), vx, acc[ct]); }
            }
        };

        // ---- both up projections of the step's rows (accumulators start at the biases)
        f32x16 aA, aG;
        load_bias(I0{}, aA);
        project(I0{}, wU, aA);
        load_bias(std::integral_constant<int, 512>{}, aG);
        project(I1{}, wGU, aG);
        // ---- dh, dq (this lane: 16 columns of its row) -> the tiles the row contractions read; dh replaces dy in place
        u32x4 dhp[2];
        const float gsr = row_ok ? a.gs : 0.f;                            // rows past the end: dh = dq = 0
        sfor<2>([&](auto C) {
            constexpr int c = C.value;
            u32x4 dyv, x2v;
            lds_read16<16 * c>(dyv, sb + a_xcl);
            lds_read16<8192 + 16 * c>(x2v, sb + a_xcl);
            lgkm_fence(dyv); lgkm_tie(x2v);
            float dh[8], dq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = 8 * c + j;
                const float gt = sigm(aG[e]);
                const float dyp = gsr * bf_at(dyv, j);
                if constexpr (ADD) {
                    dh[j] = dyp;
                    dq[j] = dyp * gt * (1.0f - gt);
                } else {
                    const float hv = s2 * bf_at(x2v, j) + sd * aA[e];
                    dh[j] = dyp * gt;
                    dq[j] = dh[j] * hv * (1.0f - gt);
                }
            }
            dhp[c] = pack8(dh);
            const u32x4 dqp = pack8(dq);
            lds_write16<16 * c>(sb + a_xcl, dhp[c]);
            lds_write16<16 * c>(a_dqcl, dqp);
        });
        // ---- down side: dWd += dpre_a^T x2, dWgd += dpre_g^T x1 (+ their bias sums in one wave per row chunk), both projections
        wg_products(I2{}, X2O{}, sb + a_xtr, accD, -1);
        wg_products(I3{}, X1O{}, sb + a_xtr, accGD, -1);
        if (want_csp) {                                                   // its own block (inside the products it would make every accumulator a phi)
            sfor<2>([&](auto KS) {
                constexpr int ks = KS.value;
                sfor<2>([&](auto TT) {
                    TrOp ap[RT];
                    sfor<RT>([&](auto CT) {
                        tr_read<(2 + TT.value) * PT_B + 64 * CT.value + ks * 16 * PB, (2 + TT.value) * PT_B + 64 * CT.value + (ks * 16 + 4) * PB>(ap[CT.value], sb + a_ptr);
                    });
                    tr_fence(ap[0]);
#pragma unroll
                    for (int ct = 0; ct < RT; ++ct) {
                        if (ct) tr_tie(ap[ct]);
                        sx = mfma32(ones_row(2 + TT.value * RT + ct), tr_val(ap[ct]), sx);
                    }
                });
            });
        }
        f32x16 p2 = zero16(), p1 = zero16();
        project(I2{}, wD, p2);
        project(I3{}, wGD, p1);
        // ---- dx2 = s2*dh + p2, dx1 = p1 (+ dx1_in): 32 bytes per lane and tensor
        if constexpr (HAS_IN) {                                           // the dx1_in loads precede this step's stage requests
            if (has2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(din0), "+v"(din1) : "n"(NI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(din0), "+v"(din1) :: "memory");
        }
        {
            uint8_t* q2 = const_cast<uint8_t*>(sbase(reinterpret_cast<const uint8_t*>(a.dx2) + rb * ld2)) + rowoff;
            uint8_t* q1 = const_cast<uint8_t*>(sbase(reinterpret_cast<const uint8_t*>(a.dx1) + rb * ld2)) + rowoff;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const u32x4 din = c == 0 ? din0 : din1;
                float o2[8], o1[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    o2[j] = s2 * bf_at(dhp[c], j) + p2[8 * c + j];      // (the bf16-rounded dh, as the weight gradients see it)
                    o1[j] = p1[8 * c + j] + bf_at(din, j);               // (zeros without dx1_in)
                }
                const u32x4 v2 = pack8(o2), v1 = pack8(o1);
                if (row_ok) {
                    reinterpret_cast<u32x4*>(q2)[c] = v2;
                    reinterpret_cast<u32x4*>(q1)[c] = v1;
                }
            }
        }
        // ---- up side: dWu += z_a^T dh, dWgu += z_g^T dq and the column sums of dh, dq (this wave's own tile columns: its
        // ds_writes above are ordered before these reads by the lgkmcnt waits in between)
        wg_products(I0{}, I0{}, sb + a_xtr, accU, 0);
        wg_products(I1{}, I0{}, a_dqtr, accGU, 1);
    }

    // ---- this row chunk's partial sums, in wgrad.hip's workspace layout (wgrad_finalize_kernel sums the chunks)
    const int RC = a.row_chunks;
    const int col = c0 + m;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        float* t = a.part[jb] + (int64_t)rc * PR * d;
        const f32x16* acc = jb == 0 ? accD : jb == 1 ? accU : jb == 2 ? accGD : accGU;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int crow = 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h;
                t[(int64_t)crow * d + col] = acc[ct][i];
            }
    }
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
}

// Row chunks: (d / 128) column blocks x chunks workgroups, at most 32 per XCD (one per CU: the ring takes the LDS), i.e. at most
// 8 * floor(32 / NCB) chunks; a chunk is a multiple of 32 rows.
void k1_cols_plan(int64_t M, int d, int* row_chunks, int64_t* rows_per_chunk) {
    const int ncb = d / 128;
    int64_t rc = 8 * (32 / (ncb < 32 ? ncb : 32));
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
    constexpr int NSTG = 3;
    const size_t lds = ColzGeo<RT>::lds(NSTG);
    auto kern = k1_cols_kernel<RT, NSTG, ADD, HAS_IN>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int ncb = c.d / 128;
    const unsigned grid = 8u * (unsigned)ncb * (unsigned)((c.row_chunks + 7) / 8);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, c);
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
