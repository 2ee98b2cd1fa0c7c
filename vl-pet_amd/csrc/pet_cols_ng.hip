// Backward of the adapter WITHOUT a gate (K2: adapters/adapter_modeling.py:55-61 + adapter_controller.py:149-162; the
// adapter-only K1 of the small / middle gate scripts; K3 when it runs without dropout) in the two-pass shape of the gated K1
// backward (pet_dz2.hip + pet_cols.hip), bf16, r <= 192, with the forward's saved z / act'(pre):
//     dz[m,c]   = sd * sum_f Wu[f,c] dy[m,f]          dpre = dz * act'(pre)                       (pass 1, row-parallel)
//     dx[m,k]   = sum_c Wd[c,k] dpre[m,c]                                                        (pass 2, column-parallel)
//     dWu[f,c]  = sd * sum_m dy[m,f] z[m,c]      dbu = sd * sum_m dy         dWd[c,k] = sum_m dpre[m,c] x[m,k]      dbd = sum_m dpre
// Until round 3 this op was a row kernel (reads dy, writes dx and dpre) + the streaming weight-gradient kernel (reads dy and x
// again): 85 us at M = 30 k for 3 algorithmic units.  Here pass 1 reads dy and writes only the [M, 32*RT] dpre; pass 2 keeps
// Wd^T of a 128-column block in registers, reads dy and x once and produces dx and both weight gradients from that one read
// (row-chunk partials in wgrad.hip's layout, wgrad_finalize_kernel sums them: deterministic).  There is no elementwise block
// and no second chain, so both kernels are small relatives of the gated ones:
//   pass 1: 128-row workgroups, the two waves of a row group split the stage's 64 features; B operand = the dy fragment of the
//           row tile (ds_read_b128), A operand = transpose reads of the Wu image (gathered from the "up" pack as in pet_dz2.hip);
//   pass 2: 8 waves = 4 column quarters x {up side: dWu + column sums of dy; down side: dWd, dx}; stage = dy and x as two
//           128-byte pair tiles each + the z and dpre tiles, three stages, one barrier per step.
#include "cols_common.h"

struct NgArgs {
    const void* dy; const void* x; const void* z; const void* gp;     // gp = act'(pre) [M, 32*RT] or nullptr (identity activation)
    void* dp;                                                           // dpre [M, 32*RT] (workspace)
    void* dx;
    const uint8_t* pk;
    int64_t M;
    int d;
    float sd;
    const uint8_t* bits; float ks;                                      // K3 dropout on x: the forward's packed mask [M, d/8] (rng.h drop_pos) and 1 / (1 - p)
    int row_chunks; int64_t rows_per_chunk;
    float* part[2];                                                     // job 0: dWd (+ column sums of dpre), job 1: dWu (+ column sums of dy)
};

// ================================================================================================== pass 1
template <int RT> struct NgDzGeo {
    static constexpr int PB = 64 * RT, NPS = PB / 16;
    static constexpr int WT_B = 64 * PB;               // the Wu block of a stage [64 f x 32*RT c]
    static constexpr int XT_B = 128 * 128;             // the dy tile
    static constexpr int STG_B = WT_B + XT_B;
    static constexpr int NWW = (4 * RT + 7) / 8;       // weight pieces per wave (the last ones may be missing)
};

// NSTG = 4: one workgroup per CU, three stages ahead (M <= 32,768: one round of 128-row workgroups on 256 CUs);
// NSTG = 2: 56 KiB of LDS, two workgroups per CU (96 registers allow it): up to 65,536 rows in one round
template <int RT, int NSTG>
__global__ __launch_bounds__(512, 2) void ng_dz_kernel(NgArgs a) {
    using GEO = NgDzGeo<RT>;
    constexpr int PB = GEO::PB, NPS = GEO::NPS, WT_B = GEO::WT_B, STG_B = GEO::STG_B, NWW = GEO::NWW;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave & 3, fh = wave >> 2;
    const int m = lane & 31, h = lane >> 5;
    const int d = a.d, S = d >> 6;
    const int64_t ld2 = (int64_t)d * 2;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    const int64_t grow_raw = row0 + 32 * rg + m;
    const bool row_ok = grow_raw < a.M;
    const int64_t grow = row_ok ? grow_raw : a.M - 1;
    const PackGeom pg = pack_geom(RT, d, 1);

    // stage pieces of this wave: two 8-row pieces of the dy tile, up to NWW pieces of the Wu block (gathered from the "up" pack:
    // fragments (stage, v, ks), slot (i, hh, j) = W[f_of4(stage, v, i)][16 ks + 8 hh + j], into natural [f][c] order, gsw-swizzled)
    uint32_t xoff[2], woff[NWW];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 8 * (wave + 8 * j) + (lane >> 3);
        int64_t gr = row0 + row;
        if (gr >= a.M) gr = a.M - 1;
        xoff[j] = (uint32_t)((gr - row0) * ld2) + (uint32_t)(((lane & 7) ^ swz(row)) * 16);
    }
    int nww = 0;
#pragma unroll
    for (int j = 0; j < NWW; ++j) {
        const int piece = wave + 8 * j;
        const int sig = piece * 64 + lane, f = (sig / NPS) & 63, sl = (sig % NPS) ^ gsw(f);
        const int i = 8 * ((f >> 2) & 3) + 4 * (f >> 5) + (f & 3), v = (f >> 4) & 1;
        woff[j] = (uint32_t)((v * 2 * RT + (sl >> 1)) * 1024 + ((sl & 1) * 32 + i) * 16);
        if (piece < 4 * RT) nww = j + 1;
    }
    auto sbase = [](const uint8_t* p) {
        const uint64_t u = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const uint8_t*>(((uint64_t)hi << 32) | lo);
    };
    const uint8_t* dyp = reinterpret_cast<const uint8_t*>(a.dy) + row0 * ld2;
    const uint8_t* wp = a.pk + pg.pack_bytes;
    auto issue = [&](int s) {
        uint8_t* st = smem + (size_t)(s % NSTG) * STG_B;
#pragma unroll
        for (int j = 0; j < NWW; ++j)
            if (j < nww) glds16(sbase(wp + (int64_t)s * (4 * RT * 1024)) + woff[j], st + (wave + 8 * j) * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16_row(sbase(dyp + s * 128) + xoff[j], st + WT_B + (wave + 8 * j) * 1024);
    };
    const int NW = nww + 2;                             // this wave's requests per stage

    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    uint32_t a_wtr[2], a_b[2];
    {
        const int g4 = lane >> 4, sl = lane & 15, hp = g4 >> 1;
        const int tslot = 2 * (g4 & 1) + ((sl & 3) >> 1), thalf = 8 * (sl & 1);
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {                // transpose reads of the Wu block: rows 32 fh + 8 hp + 4 hi + (sl >> 2) (+ 16 kappa)
            const int r = 32 * fh + 8 * hp + 4 * hi + (sl >> 2);
            a_wtr[hi] = (uint32_t)(r * PB + ((tslot ^ gsw(r)) * 16) + thalf);
        }
        const int row = 32 * rg + m;
#pragma unroll
        for (int k = 0; k < 2; ++k)                     // B fragment: features 32 fh + 16 k + 8 h .. + 7 of the lane's row
            a_b[k] = (uint32_t)(WT_B + row * 128 + (((4 * fh + 2 * k + h) ^ swz(row)) * 16));
    }
    f32x16 dz[RT];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) dz[ct] = zero16();

#pragma unroll
    for (int s = 0; s < NSTG - 1; ++s)
        if (s < S) issue(s);
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
        int ahead = S - 1 - s;
        if (ahead > NSTG - 2) ahead = NSTG - 2;
        vm_wait(ahead * NW);
        __builtin_amdgcn_s_barrier();
        if (s + NSTG - 1 < S) issue(s + NSTG - 1);
        const uint32_t sb = lds0 + (uint32_t)((s % NSTG) * STG_B);
        u32x4 bf[2];
        TrOp ap[2][RT];
        lds_read16<0>(bf[0], sb + a_b[0]); lds_read16<0>(bf[1], sb + a_b[1]);
        sfor<2>([&](auto KP) {
            sfor<RT>([&](auto CT) { tr_read2<KP.value * 16 * PB + 64 * CT.value>(ap[KP.value][CT.value], sb + a_wtr[0], sb + a_wtr[1]); });
        });
        lgkm_fence(bf[0]); lgkm_tie(bf[1]);
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) {
                tr_tie(ap[kp][ct]);
                dz[ct] = mfma32(tr_val(ap[kp][ct]), as_bf(bf[kp]), dz[ct]);
            }
    }

    // ---- the two feature halves meet; the fh = 0 wave finishes: dpre = sd * dz * act'(pre)
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 gpv[RT][4];
    if (fh == 0 && a.gp != nullptr) {
        const __bf16* gp = reinterpret_cast<const __bf16*>(a.gp) + grow * (int64_t)(32 * RT) + 4 * h;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) gpv[ct][q] = *reinterpret_cast<const bf16x4*>(gp + 32 * ct + 8 * q);
    }
    __syncthreads();
    float* xch = reinterpret_cast<float*>(smem) + (size_t)rg * (RT * 16 * 64);
    if (fh == 1) {
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 t = {dz[ct][4 * q], dz[ct][4 * q + 1], dz[ct][4 * q + 2], dz[ct][4 * q + 3]};
                *reinterpret_cast<f32x4*>(xch + (size_t)((ct * 4 + q) * 64 + lane) * 4) = t;
            }
    }
    __syncthreads();
    if (fh == 0) {
        __bf16* out = reinterpret_cast<__bf16*>(a.dp) + grow * (int64_t)(32 * RT) + 4 * h;
        const bool has_gp = a.gp != nullptr;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(xch + (size_t)((ct * 4 + q) * 64 + lane) * 4);
                bf16x4 r4;
#pragma unroll
                for (int j = 0; j < 4; ++j) r4[j] = (__bf16)(a.sd * (dz[ct][4 * q + j] + o[j]) * (has_gp ? (float)gpv[ct][q][j] : 1.0f));
                if (row_ok) *reinterpret_cast<bf16x4*>(out + 32 * ct + 8 * q) = r4;
            }
    }
}

// ================================================================================================== pass 2
template <int RT> struct NgColGeo {
    static constexpr int KT = 2 * RT, PB = 64 * RT, PT_B = 32 * PB;
    static constexpr int X_B = 2 * 2 * 4096;           // dy, x: two pair tiles [32 rows x 128 B] each
    static constexpr int MASK_OFF = X_B + 2 * PT_B;    // + z, dpre, then (DROP) 32 rows x 16 mask bytes of the 128 columns
    static constexpr int STG_B = MASK_OFF + 1024;
    static constexpr int NSTG = 3;
};

// DROP (K3 with dropout): x enters the op as dropout(x) = x * mask / (1 - p), so dWd (= dA) sees the masked x and dx = mask / (1 - p) *
// Wd^T dpre.  The 16 mask bytes per row of the workgroup's 128 columns travel with the stage (one more request of the last wave);
// every down-side wave clears the dropped elements of ITS 32 columns of the x tile in LDS before its transpose reads (nobody else
// reads them) and masks its 16 outputs per row; 1 / (1 - p) goes into dx directly and into dWd through the finalize scale.
template <int RT, bool DROP>
__global__ __launch_bounds__(512, 2) void ng_cols_kernel(NgArgs a) {
    using GEO = NgColGeo<RT>;
    constexpr int KT = GEO::KT, PB = GEO::PB, PT_B = GEO::PT_B, X_B = GEO::X_B, STG_B = GEO::STG_B, NSTG = GEO::NSTG, MASK_OFF = GEO::MASK_OFF;
    constexpr int PR = 32 * RT;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int d = a.d, NCB = d >> 7;
    int rc, cb;
    cols_decode((int)blockIdx.x, NCB, rc, cb);
    if (rc >= a.row_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, wc = wave & 3;          // waves w and w + 4 share a SIMD: up side and down side of a column quarter
    const int pp = wc >> 1, nt = wc & 1;
    const int m = lane & 31, h = lane >> 5;
    const int64_t ld2 = (int64_t)d * 2;
    const int c0 = 128 * cb + 32 * wc;
    const int64_t r_begin = (int64_t)rc * a.rows_per_chunk;
    int64_t r_end = r_begin + a.rows_per_chunk;
    if (r_end > a.M) r_end = a.M;
    const int nsteps = r_end > r_begin ? (int)((r_end - r_begin + 31) >> 5) : 0;
    const PackGeom pg = pack_geom(RT, d, 1);

    // down side: Wd transposed from the "down_t" pack (slot = W[16ks + 8hh + j][f]); MFMA row i stands for column
    // c0 + 16*((i>>2)&1) + 4*(i>>3) + (i&3), so that a lane (m, h) ends with the 16 contiguous columns c0 + 16h .. of row m
    bf16x8 wD[KT];
    if (role == 1) {
        const int i = m, v = (i >> 2) & 1, ip = (i & 3) | (nt << 2) | ((i >> 3) << 3);
        const int64_t off = (int64_t)3 * pg.pack_bytes + (int64_t)(2 * cb + pp) * (4 * RT * 1024) + (int64_t)(v * KT) * 1024 + (ip + 32 * h) * 16;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) wD[ks] = *reinterpret_cast<const bf16x8*>(a.pk + off + ks * 1024);
    }

    // ---- stage pieces of this wave: pieces q = wave, wave + 8 of the 16 row pieces (tensor q / 8, pair (q / 4) % 2, rows
    // 8 (q % 4) ..), pieces q' = wave + 8 j < 4 RT of the bottleneck tiles (tensor q' / KT, piece q' % KT)
    constexpr int NPJ = (4 * RT + 7) / 8;
    const int xrow = 8 * (wave & 3) + (lane >> 3);
    const uint32_t xoff = (uint32_t)xrow * (uint32_t)ld2 + (uint32_t)((128 * cb + 64 * ((wave >> 2) & 1)) * 2 + (((lane & 7) ^ fsw(xrow)) * 16));
    const uint8_t* pbase[NPJ]; uint32_t poff[NPJ], pdst[NPJ]; int prow[NPJ];
    int npj = 0;
#pragma unroll
    for (int j = 0; j < NPJ; ++j) {
        const int q = wave + 8 * j, t = (q / KT) & 1, piece = q % KT;
        const int sig = piece * 64 + lane;
        prow[j] = sig / (PB / 16);
        pbase[j] = reinterpret_cast<const uint8_t*>(t == 0 ? a.z : a.dp);
        poff[j] = (uint32_t)(prow[j] * PB + ((sig % (PB / 16)) ^ gsw(prow[j])) * 16);
        pdst[j] = (uint32_t)(X_B + t * PT_B + piece * 1024);
        if (q < 4 * RT) npj = j + 1;
    }
    const bool mask_wave = DROP && wave == 7;
    const int NW = 2 + npj + (mask_wave ? 1 : 0);
    auto sbase = [](const uint8_t* p) {
        const uint64_t u = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const uint8_t*>(((uint64_t)hi << 32) | lo);
    };
    const uint8_t* dyb = reinterpret_cast<const uint8_t*>(a.dy);
    const uint8_t* xb_ = reinterpret_cast<const uint8_t*>(a.x);
    auto issue = [&](int s) {
        const int64_t rb = r_begin + 32 * (int64_t)s;
        uint8_t* st = smem + (size_t)(s % NSTG) * STG_B;
        const int last = (int)(r_end - rb) - 1;         // rows past the end re-read the last row
        const uint32_t xo = xoff - (uint32_t)(xrow > last ? xrow - last : 0) * (uint32_t)ld2;
        // row pieces: this wave brings rows 8 (wave & 3) .. of pair (wave >> 2) & 1 of BOTH tensors
        glds16_row(sbase(dyb + rb * ld2) + xo, st + (0 * 2 + ((wave >> 2) & 1)) * 4096 + (wave & 3) * 1024);
        glds16_row(sbase(xb_ + rb * ld2) + xo, st + (1 * 2 + ((wave >> 2) & 1)) * 4096 + (wave & 3) * 1024);
#pragma unroll
        for (int j = 0; j < NPJ; ++j)
            if (j < npj)
                glds16(sbase(pbase[j] + rb * PB) + poff[j] - (uint32_t)(prow[j] > last ? prow[j] - last : 0) * PB, st + pdst[j]);
        if constexpr (DROP) {
            if (mask_wave) {                            // lane -> row lane & 31 (both halves of the wave bring the same 512 bytes)
                const int mr = (lane & 31) > last ? last : (lane & 31);
                glds16(sbase(a.bits + rb * (int64_t)(d >> 3) + 16 * cb) + (uint32_t)mr * (uint32_t)(d >> 3), st + MASK_OFF);
            }
        }
    };

    // ---- per-lane LDS byte addresses (relative to the stage base)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    uint32_t a_xtr[2], a_ptr[2], a_pbf[2];
    {
        const int g4 = lane >> 4, sl = lane & 15;
        const int trow = 8 * (g4 >> 1) + (sl >> 2);
        const int tslot = 2 * (g4 & 1) + ((sl & 3) >> 1), thalf = 8 * (sl & 1);
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int r = trow + 4 * hi;
            a_xtr[hi] = (uint32_t)(pp * 4096 + r * 128 + (((4 * nt + tslot) ^ fsw(r)) * 16) + thalf);     // row tiles (+ 8192 per tensor)
            a_ptr[hi] = (uint32_t)(X_B + r * PB + ((tslot ^ gsw(r)) * 16) + thalf);                        // bottleneck tiles (+ 64 ct, + PT_B per tensor)
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) a_pbf[k] = (uint32_t)(X_B + PT_B + m * PB + (((2 * k + h) ^ gsw(m)) * 16));   // B fragment of dpre row m
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

    f32x16 acc[RT], sx = zero16();
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) acc[ct] = zero16();
    // which bottleneck tile's column sums (dbd) this up-side wave takes: tile k = c-tile k of dpre goes to wave number wc * NCB + cb
    int csp_tile[2] = {-1, -1};
    if (role == 0) {
        const int w = wc * NCB + cb;
        if (w < RT) csp_tile[0] = w;
        if (w + 4 * NCB < RT) csp_tile[1] = w + 4 * NCB;
    }

    auto step_top = [&](int s, int extra) {
        int ahead = nsteps - 1 - s;
        if (ahead > NSTG - 2) ahead = NSTG - 2;
        vm_wait(ahead * NW + extra);
        __builtin_amdgcn_s_barrier();
        if (s + NSTG - 1 < nsteps) issue(s + NSTG - 1);
        const int valid = (int)(r_end - (r_begin + 32 * (int64_t)s));
        if (valid < 32) {                               // zero the bottleneck rows past the end (their products must vanish)
            const u32x4 z = {0u, 0u, 0u, 0u};
            uint8_t* pt = smem + (size_t)(s % NSTG) * STG_B + X_B;
            for (int q = tid; q < 2 * 32 * (PB / 16); q += 512) {
                const int rr = (q / (PB / 16)) & 31;
                if (rr >= valid) *reinterpret_cast<u32x4*>(pt + (size_t)q * 16) = z;
            }
            // ... and the dy rows past the end (re-reads of the last row): they would count in the column sums of dy
            if (((tid >> 3) & 31) >= valid) *reinterpret_cast<u32x4*>(smem + (size_t)(s % NSTG) * STG_B + (size_t)tid * 16) = z;
            __syncthreads();
        }
    };
    // acc[ct] += P^T (bottleneck tile TP) . X (row tensor TX, this wave's 32 columns), both 16-row k-steps in one batch
    auto wg_products = [&](uint32_t sb, auto TPC, auto TXC, int slot) {
        constexpr int TP = decltype(TPC)::value, TX = decltype(TXC)::value;
        TrOp bx[2], ap[2][RT];
        sfor<2>([&](auto KS) {
            constexpr int ks = KS.value;
            tr_read2<TX * 8192 + ks * 16 * 128>(bx[ks], sb + a_xtr[0], sb + a_xtr[1]);
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
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int s0 = 0; s0 < NSTG - 1; ++s0)
        if (s0 < nsteps) issue(s0);

    if (role == 0) {
        // ================================================================ up side: dWu += z^T dy, column sums of dy and of dpre
#pragma unroll 1
        for (int s = 0; s < nsteps; ++s) {
            const uint32_t sb = lds0 + (uint32_t)((s % NSTG) * STG_B);
            step_top(s, 0);
            wg_products(sb, I0{}, I0{}, 0);
            if (csp_tile[0] >= 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = csp_tile[j];
                    if (k < 0) break;
                    const uint32_t off = (uint32_t)(PT_B + 64 * k);
                    TrOp ap[2];
                    tr_read2<0>(ap[0], sb + a_ptr[0] + off, sb + a_ptr[1] + off);
                    tr_read2<16 * PB>(ap[1], sb + a_ptr[0] + off, sb + a_ptr[1] + off);
                    tr_fence(ap[0]);
                    sx = mfma32(ones_row(2 + j), tr_val(ap[0]), sx);
                    tr_tie(ap[1]);
                    sx = mfma32(ones_row(2 + j), tr_val(ap[1]), sx);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (h == 0) {
            a.part[1][(int64_t)RC * PR * d + (int64_t)rc * d + col] = sx[0];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = csp_tile[j];
                if (k < 0) break;
                a.part[0][(int64_t)RC * PR * d + (int64_t)RC * d + (int64_t)rc * PR + 32 * k + m] = sx[2 + j];
            }
        }
    } else {
        // ================================================================ down side: dWd += dpre^T x, dx = Wd^T dpre
#pragma unroll 1
        for (int s = 0; s < nsteps; ++s) {
            const uint32_t sb = lds0 + (uint32_t)((s % NSTG) * STG_B);
            step_top(s, s >= 1 ? 2 : 0);                // (its two output stores of the previous step may stay in flight)
            const int64_t rb = r_begin + 32 * (int64_t)s;
            const int valid = (int)(r_end - rb) < 32 ? (int)(r_end - rb) : 32;
            uint32_t mlo = 0xffffffffu, mhi = 0xffffffffu;                 // row m's keep flags of this wave's pair of 64 columns
            if constexpr (DROP) {
                // this lane clears row lane >> 1, groups 4 nt + 2 (lane & 1) + {0, 1} of the pair tile (two 16-byte slots)
                const int xr = lane >> 1, par = lane & 1;
                u32x2 mk;
                lds_read8<0>(mk, sb + (uint32_t)(MASK_OFF + xr * 16 + 8 * pp));
                u32x4 v0, v1;
                const uint32_t xa0 = sb + (uint32_t)(8192 + pp * 4096 + xr * 128 + (((4 * nt + 2 * par) ^ fsw(xr)) * 16));
                const uint32_t xa1 = sb + (uint32_t)(8192 + pp * 4096 + xr * 128 + (((4 * nt + 2 * par + 1) ^ fsw(xr)) * 16));
                lds_read16<0>(v0, xa0); lds_read16<0>(v1, xa1);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mk), "+v"(v0), "+v"(v1) :: "memory");
                const uint32_t k0 = mk[0] >> (8 * (2 * nt + par)), k1 = mk[1] >> (8 * (2 * nt + par));   // even group, odd group
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int l0 = ((int)(k0 << (31 - 2 * q))) >> 31, h0 = ((int)(k0 << (30 - 2 * q))) >> 31;
                    const int l1 = ((int)(k1 << (31 - 2 * q))) >> 31, h1 = ((int)(k1 << (30 - 2 * q))) >> 31;
                    v0[q] &= __builtin_amdgcn_perm((uint32_t)h0, (uint32_t)l0, 0x07060100u);
                    v1[q] &= __builtin_amdgcn_perm((uint32_t)h1, (uint32_t)l1, 0x07060100u);
                }
                lds_write16<0>(xa0, v0); lds_write16<0>(xa1, v1);
                u32x2 mo;                                                 // ... and the flags of its own 16 output columns of row m
                lds_read8<0>(mo, sb + (uint32_t)(MASK_OFF + m * 16 + 8 * pp));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mo) :: "memory");
                mlo = mo[0] >> (8 * (2 * nt + h)); mhi = mo[1] >> (8 * (2 * nt + h));
            }
            wg_products(sb, I1{}, I1{}, -1);
            f32x16 p2 = zero16();
            {
                u32x4 bf[KT];
                sfor<KT>([&](auto K) { lds_read16<64 * (K.value >> 1)>(bf[K.value], sb + a_pbf[K.value & 1]); });
                lgkm_fence(bf[0]);
#pragma unroll
                for (int k = 0; k < KT; ++k) { if (k) lgkm_tie(bf[k]); p2 = mfma32(wD[k], as_bf(bf[k]), p2); }
            }
            {
                float o[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if constexpr (DROP) o[e] = (((e < 8 ? mlo : mhi) >> (e & 7)) & 1u) ? p2[e] * a.ks : 0.f;
                    else o[e] = p2[e];
                }
                const u32x4 v0 = pack8(o), v1 = pack8(o + 8);
                if (m < valid) {
                    const uint32_t rowoff = (uint32_t)m * (uint32_t)ld2 + (uint32_t)((c0 + 16 * h) * 2);
                    uint8_t* q2 = const_cast<uint8_t*>(sbase(reinterpret_cast<const uint8_t*>(a.dx) + rb * ld2)) + rowoff;
                    reinterpret_cast<u32x4*>(q2)[0] = v0;
                    reinterpret_cast<u32x4*>(q2)[1] = v1;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    // ---- this row chunk's partial sums (wgrad.hip's workspace layout)
    {
        float* t = a.part[role == 0 ? 1 : 0] + (int64_t)rc * PR * d;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int crow = 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h;
                t[(int64_t)crow * d + col] = acc[ct][i];
            }
    }
}

// ================================================================================================== host side
bool ng_two_pass_applies(const PetBwdArgs& a, int io_fp32) {
    if (io_fp32 || (a.flags & PET_GATE) || a.saved == nullptr) return false;
    // dropout (K3): only with the forward's packed mask (the training form), never an explicit byte mask or the generator
    if (drop_active(a.drop) && (a.drop.bits == nullptr || a.drop.keep != nullptr || !(a.flags & PET_ACT_IDENTITY))) return false;
    if (!(a.RT == 1 || a.RT == 3 || a.RT == 6) || a.d % 128 != 0 || a.d / 128 > 32) return false;
    return true;
}
void ng_cols_plan(int64_t M, int d, int* row_chunks, int64_t* rows_per_chunk) {
    const int ncb = d >= 128 ? d / 128 : 1;
    int64_t rc = cols_groups_max(ncb < 32 ? ncb : 32);
    const int64_t blocks32 = (M + 31) / 32;
    if (rc > blocks32) rc = blocks32;
    if (rc < 1) rc = 1;
    const int64_t per = (blocks32 + rc - 1) / rc;
    rc = (blocks32 + per - 1) / per;
    *row_chunks = (int)rc; *rows_per_chunk = per * 32;
}

template <int RT>
static hipError_t launch_ng_rt(const NgArgs& a, int passes, hipStream_t stream) {
    if (passes & 1) {
        const bool two_per_cu = RT <= 3 && a.M > 128 * 256;   // more 128-row workgroups than CUs: the two-slot form, two per CU (96 registers)
        constexpr int NSTG1 = RT <= 3 ? 4 : 3;          // (six tiles: 40-KiB stages)
        const size_t lds = (size_t)(two_per_cu ? 2 : NSTG1) * NgDzGeo<RT>::STG_B;
        const void* kern = two_per_cu ? reinterpret_cast<const void*>(ng_dz_kernel<RT, 2>) : reinterpret_cast<const void*>(ng_dz_kernel<RT, NSTG1>);
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        const dim3 grid((unsigned)((a.M + 127) / 128));
        if (two_per_cu) hipLaunchKernelGGL((ng_dz_kernel<RT, 2>), grid, dim3(512), lds, stream, a);
        else hipLaunchKernelGGL((ng_dz_kernel<RT, NSTG1>), grid, dim3(512), lds, stream, a);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (passes & 2) {
        const size_t lds = (size_t)NgColGeo<RT>::NSTG * NgColGeo<RT>::STG_B;
        const bool drop = a.bits != nullptr;
        const void* kern = drop ? reinterpret_cast<const void*>(ng_cols_kernel<RT, true>) : reinterpret_cast<const void*>(ng_cols_kernel<RT, false>);
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        const dim3 grid(cols_grid(a.d / 128, a.row_chunks));
        if (drop) hipLaunchKernelGGL((ng_cols_kernel<RT, true>), grid, dim3(512), lds, stream, a);
        else hipLaunchKernelGGL((ng_cols_kernel<RT, false>), grid, dim3(512), lds, stream, a);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// passes: bit 0 = pass 1 (dpre), bit 1 = pass 2 (dx + the row-chunk partials of dWd, dWu; the caller runs the finalize)
hipError_t launch_ng_two_pass(const PetBwdArgs& b, const WgradArgs& g, int passes, hipStream_t stream) {
    NgArgs a{};
    a.dy = b.dy; a.x = b.xa; a.z = b.z_a;
    a.gp = (b.flags & PET_ACT_IDENTITY) ? nullptr : reinterpret_cast<const uint8_t*>(b.saved) + b.saved_stride;
    a.dp = b.dp_a; a.dx = b.dxa; a.pk = b.pk_a;
    a.M = b.M; a.d = b.d; a.sd = b.sd;
    a.bits = drop_active(b.drop) ? b.drop.bits : nullptr; a.ks = drop_active(b.drop) ? b.drop.keep_scale : 1.f;
    a.row_chunks = g.row_chunks; a.rows_per_chunk = g.rows_per_chunk;
    const WgradLayout L = wgrad_layout(g);
    a.part[0] = g.partial + L.off[0]; a.part[1] = g.partial + L.off[1];
    return b.RT == 1 ? launch_ng_rt<1>(a, passes, stream) : b.RT == 3 ? launch_ng_rt<3>(a, passes, stream) : launch_ng_rt<6>(a, passes, stream);
}
