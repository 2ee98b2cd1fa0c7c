// K4 forward, round 5: the visual-feature projection as a tiled GEMM whose column tiles exchange the LayerNorm statistics
//
//     out = LN( feats . W^T + b ) * gamma + beta  (+ R)          feats [M, F] bf16, W [d_out, F] bf16 (row-major), d_out = NT * 256
//
// Reference: the feat_embedding branch of VisualEmbedding.forward (src/modeling_bart.py:157; nn.Sequential(Linear(feat_dim, d_model),
// LayerNorm) built at :91-110; T5: src/modeling_t5.py:56-66 with T5LayerNorm = rms form); R = position branch + order embeddings
// (:162-183), added after the norm.
//
// Why a new kernel.  The round-2 kernel (visproj.hip) gives a workgroup 128 rows and ALL 768 output features so that the LayerNorm
// is local: 146 workgroups at 18,700 rows (57 % of the CUs), each streaming the whole 3.1 MB weight through its LDS (460 MB of
// L2 -> LDS traffic) behind a barrier per 32 input features -- 106 us, 0.22 of the MFMA peak; round 4 replaced it by the library GEMM
// + a norm pass (48 + 24 us).  Here the product is tiled like a GEMM -- a workgroup owns BM rows x 256 output features, K in stages of
// BK through an NSLOT-deep LDS ring (global_load_lds, counted vmcnt, one raw barrier per stage), both operands as [rows][BK] images
// with the 16-byte slots XOR-swizzled on the source side so that the ds_read_b128 fragment reads are conflict-free -- and the NT = 3
// workgroups of a row block (a TEAM, placed on one XCD so that the feature rows they all read are L2 hits) exchange per-row partial
// statistics once, through L2, after their K loops:
//   * statistics: in registers per wave (64 features of a row live in lanes n, n + 32: sum -> mean, then centred squares), combined
//     over the 4 feature waves in LDS and over the NT workgroups in L2 by Chan's parallel-variance formula (exact and stable: no
//     E[x^2] - mean^2 cancellation); the rms form exchanges the sum of squares only;
//   * hand-over (cdna_hip_programming.md, Guideline 16, form R2): 8-byte {tag = epoch, value} granules written by ONE agent-scope
//     (write-through) store each and polled relaxed by the threads that need them -- no flag, no fence; tags count the row blocks of
//     THIS launch (epoch = iteration + 1), two slots by parity (a partner is at most one row block ahead); every consumer has its own
//     copy of a producer's granules and zeroes it after reading, so a zeroed exchange area is left zeroed by every launch and needs
//     no memset node between launches (a per-launch salt in the tags would be frozen under graph replay);
//   * residency: the grid is at most one workgroup per CU (128 KiB of LDS each) and never more than the chip's CUs, teams walk the
//     row blocks persistently; a team's members therefore run concurrently -- AS LONG AS the GPU is this launch's alone.  It need not
//     be (a collective on another stream, a second process on the device), so every spin is bounded and a timeout is REPAIRED inside
//     the same call (round 6): every workgroup also leaves its per-row partial statistics in a plain array of the workspace; a
//     workgroup that gives up stores its PRE-norm tile (bias added, bf16) where xhat goes, flags its (row block, column tile) and
//     bumps the workspace's timeout counter; visproj_gemm_repair_kernel, launched right behind the main kernel by the same call,
//     returns at once unless that counter has moved -- then it normalises the flagged tiles from the (by then complete) statistics
//     array, in the main kernel's order of combination, and zeroes the exchange area (a late producer writes its granules after
//     the consumer that gave up has cleared them).  The caller never sees rows normalised with partial statistics, and the next
//     launch starts from a clean area; the status word only reports that it happened.
// Epilogue: R arrives by global_load_lds into the wave's private staging area while the statistics are exchanged; out and xhat leave
// through the same area as whole 128-byte lines.
#include "cols_common.h"
#include "kernels.h"

#define VISG_MAX_SPIN (1u << 20)
#define VISG_FLAGS_B 65536       // give-up flags of the tiles: one fixed-size area (see visproj_gemm_workspace_bytes)
#ifndef VISG_DEFAULT_FORM
#define VISG_DEFAULT_FORM 4      // BK 64, two slots, spread requests: profiles/r05_k4bench.txt
#endif

template <int N, int RTN>
__device__ __forceinline__ void visg_frag_wait(u32x4 (&w)[2], u32x4 (&x)[RTN]) {
    static_assert(RTN >= 2 && RTN <= 4, "");
    if constexpr (RTN == 4)
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(w[0]), "+v"(w[1]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "n"(N) : "memory");
    else if constexpr (RTN == 3)
        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(w[0]), "+v"(w[1]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]) : "n"(N) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(w[0]), "+v"(w[1]), "+v"(x[0]), "+v"(x[1]) : "n"(N) : "memory");
}

// (inline asm like every LDS access of the epilogue: hipcc drains the LDS-DMA queue in front of an LDS access it can see)
__device__ __forceinline__ void visg_lds_write4(uint32_t addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned visg_lds_read4(uint32_t addr) {
    unsigned o;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(o) : "v"(addr) : "memory");
    return o;
}

template <int BM, int BK, int NSLOT> struct VisGemmGeo {
    static constexpr int ROWB = BK * 2;                 // bytes of a tile row
    static constexpr int SLOTS = ROWB / 16;             // 16-byte slots of a row (8 / 4)
    static constexpr int RPB = 256 / ROWB;              // rows per 256-byte bank row (2 / 4)
    static constexpr int RPP = 1024 / ROWB;             // rows per 1-KiB piece (8 / 16)
    static constexpr int XT_B = BM * ROWB, WT_B = 256 * ROWB, STG_B = XT_B + WT_B;
    static constexpr int PX = XT_B / 8192, PW = WT_B / 8192, NPW = PX + PW;       // pieces per wave and stage
    static constexpr int KU = BK / 16;
    static constexpr int RTN = BM / 64;                 // 32-row tiles per wave (a wave owns BM / 2 rows x 64 features)
    static constexpr int RING_B = NSLOT * STG_B;
    static constexpr int STAGE_B = BM * 512;            // epilogue staging: 8 waves x (BM / 2) rows x 128 B
    static constexpr int MAIN_B = RING_B > STAGE_B ? RING_B : STAGE_B;
    static constexpr int PRM_OFF = MAIN_B;              // bias | gamma | beta of the workgroup's 256 features (fp32)
    static constexpr int WST_OFF = PRM_OFF + 3 * 256 * 4;        // per-wave statistics [BM][4] x (mean, M2)
    static constexpr int RST_OFF = WST_OFF + BM * 32;            // row statistics [BM] x (mean, rstd)
    static constexpr int TMO_OFF = RST_OFF + BM * 8;             // != 0: a wave of this workgroup gave up on a partner's statistics (this row block)
    static constexpr int LDS_B = TMO_OFF + 16;
    static_assert(PX >= 1 && PW >= 1 && XT_B % 8192 == 0 && LDS_B <= 160 * 1024, "");
    static_assert(BK == 64 || BK == 32, "");
};

// SPREAD: the LDS-DMA requests of the stage ahead are issued between the MFMA groups of the current stage (two per k-step) instead
// of in one burst after the barrier (each costs 60-100 cycles of issue during which this wave feeds no MFMA).
template <int BM, int BK, int NSLOT, bool SPREAD>
__global__ __launch_bounds__(512, 2) void visproj_gemm_kernel(VisGemmArgs a) {
    using GEO = VisGemmGeo<BM, BK, NSLOT>;
    constexpr int ROWB = GEO::ROWB, SLOTS = GEO::SLOTS, RPB = GEO::RPB, RPP = GEO::RPP, XT_B = GEO::XT_B, STG_B = GEO::STG_B;
    constexpr int PX = GEO::PX, PW = GEO::PW, NPW = GEO::NPW, KU = GEO::KU, RTN = GEO::RTN, HB = BM / 2;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int NT = a.d_out >> 8;
    int team, member;
    cols_decode((int)blockIdx.x, NT, team, member);
    if (team >= a.nteams) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;            // row half, feature quarter (64 features)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    const int F2 = a.F * 2, D2 = a.d_out * 2;
    const int T = a.F / BK;

    // parameters of this workgroup's 256 features -> LDS (fp32)
    {
        float* prm = reinterpret_cast<float*>(smem + GEO::PRM_OFF);
        if (tid < 256) {
            const int f = member * 256 + tid;
            prm[tid] = a.bias ? a.bias[f] : 0.f;
            prm[256 + tid] = a.gamma[f];
            prm[512 + tid] = a.beta ? a.beta[f] : 0.f;
        }
    }
    auto sbase = [](const uint8_t* p) {     // a wave-uniform pointer as a fresh scalar (keeps the per-lane part a 32-bit loop invariant)
        const uint64_t u = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const uint8_t*>(((uint64_t)hi << 32) | lo);
    };
    const uint8_t* Xp = reinterpret_cast<const uint8_t*>(a.feats);
    const uint8_t* Wp = reinterpret_cast<const uint8_t*>(a.w);
    const uint8_t* Rp = reinterpret_cast<const uint8_t*>(a.R);
    uint8_t* Op = reinterpret_cast<uint8_t*>(a.out);
    uint8_t* Hp = reinterpret_cast<uint8_t*>(a.xhat);
    const float inv_d = 1.0f / (float)a.d_out;

    // wall-clock stamps of workgroup 0 (10 ns units) in the workspace header behind the status word: start, first stage requested, K loop
    // done, statistics published, statistics combined, out stored, end (tools/k4bench.py prints the differences)
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(reinterpret_cast<uint8_t*>(a.status) + 64);
    const bool stamping = blockIdx.x == 0 && tid == 0;
#define VISG_STAMP(k) { if (stamping) stamps[k] = wall_clock64(); }
    VISG_STAMP(0)
    int it = 0;
    for (int rb = team; rb < a.row_blocks; rb += a.nteams, ++it) {
        const int64_t row0 = (int64_t)rb * BM;
        if (tid == 0) visg_lds_write4(lds0 + (uint32_t)GEO::TMO_OFF, 0u);                   // (several barriers before anyone sets or reads it)
        // every per-lane constant is re-derived per row block from an opaque copy of the lane id: as loop invariants of the persistent
        // loop they stayed live through the epilogue (128 accumulator registers + its temporaries) and hipcc spilled 14-18 of them
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));
        const int n = lane_v & 31, kh = lane_v >> 5;
    // ---- stage pieces of this wave (1 KiB = RPP rows of a tile): piece p = 8 i + wave; lane -> (row lr, slot ls), slot swizzled at the source
    const int lr = lane_v / SLOTS, ls = lane_v % SLOTS;
    auto sw = [](int row) { return (row / RPB) % SLOTS; };
    uint32_t woff[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int row = RPP * (8 * i + wave) + lr;                       // feature row of the workgroup's 256
        woff[i] = (uint32_t)(member * 256 + row) * (uint32_t)F2 + (uint32_t)((ls ^ sw(row)) * 16);
    }
    // fragment reads: lane (n, kh) reads 16 bytes at slot (2 u + kh) ^ sw of row n (+ 32 per tile: the swizzle repeats every 32 rows)
    uint32_t foff[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) foff[u] = (uint32_t)(((2 * u + kh) ^ sw(n)) * 16);
    const uint32_t xbase = (uint32_t)((wm * HB + n) * ROWB);
    const uint32_t wbase = (uint32_t)(XT_B + (wn * 64 + n) * ROWB);
        uint32_t xoff[PX];
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            const int row = RPP * (8 * i + wave) + lr;
            int64_t rg = row0 + row;
            if (rg >= a.M) rg = a.M - 1;
            xoff[i] = (uint32_t)rg * (uint32_t)F2 + (uint32_t)((ls ^ sw(row)) * 16);
        }
        auto issue_piece = [&](int s, auto IC) {             // piece I of stage s: the X pieces first, then the W pieces
            constexpr int I = decltype(IC)::value;
            if constexpr (I < NPW) {
                uint8_t* st = smem + (size_t)(s % NSLOT) * STG_B;
                const int kb = s * ROWB;
                if constexpr (I < PX) glds16(sbase(Xp + kb) + xoff[I], st + (8 * I + wave) * 1024);
                else glds16(sbase(Wp + kb) + woff[I - PX], st + XT_B + (8 * (I - PX) + wave) * 1024);
            }
        };
        auto issue = [&](int s) { sfor<NPW>([&](auto IC) { issue_piece(s, IC); }); };
        f32x16 acc[2][RTN];
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int rt = 0; rt < RTN; ++rt) acc[ft][rt] = zero16();

        // K loop with a SKEWED barrier.  The hand-over to stage t + 1 (counted vmcnt, barrier, first fragment reads of t + 1, request of
        // stage t + NSLOT into the slot of stage t) sits in front of the LAST k-step's MFMAs of stage t, whose operands are already in
        // registers: every wave has completed its reads of stage t when it arrives, and it leaves with 2 x RTN MFMAs queued that
        // cover the barrier skew and the LDS latency of the first reads (with the hand-over between two stages the matrix pipes of a
        // CU drained at every barrier: 68 % of the MFMA rate in the loop of a lone workgroup).
#pragma unroll
        for (int s = 0; s < NSLOT; ++s)
            if (s < T) issue(s);
        vm_wait(((T < NSLOT ? T : NSLOT) - 1) * NPW);
        __builtin_amdgcn_s_barrier();
        u32x4 wf[2][2], xf[2][RTN];
        auto read = [&](int t, auto UC) {
            constexpr int u = decltype(UC)::value, b = u & 1;
            const uint32_t sb = lds0 + (uint32_t)((t % NSLOT) * STG_B);
            sfor<2>([&](auto FT) { lds_read16<FT.value * 32 * ROWB>(wf[b][FT.value], sb + wbase + foff[u]); });
            sfor<RTN>([&](auto RT_) { lds_read16<RT_.value * 32 * ROWB>(xf[b][RT_.value], sb + xbase + foff[u]); });
        };
        read(0, std::integral_constant<int, 0>{});
        VISG_STAMP(1)
        constexpr int PPU = (NPW + 2 * KU - 1) / (2 * KU);                 // pieces per half k-step under SPREAD
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            const bool next = t + 1 < T;
            const bool req_new = next && t + NSLOT < T;                    // stage t + NSLOT is requested from the hand-over on
            const bool req_old = t >= 1 && t - 1 + NSLOT < T;              // ... and stage t - 1 + NSLOT is still being requested (SPREAD)
            sfor<KU>([&](auto UC) {
                constexpr int u = decltype(UC)::value, b = u & 1;
                if constexpr (u + 1 < KU) {
                    read(t, std::integral_constant<int, u + 1>{});
                    visg_frag_wait<2 + RTN, RTN>(wf[b], xf[b]);
                } else {
                    visg_frag_wait<0, RTN>(wf[b], xf[b]);
                    if (next) {
                        int inflight = T - 1 - t;                          // stages t + 1 .. requested so far
                        if (inflight > NSLOT - 1) inflight = NSLOT - 1;
                        vm_wait((inflight - 1) * NPW);
                        __builtin_amdgcn_s_barrier();                      // stage t + 1 is complete; every wave is done reading stage t
                        read(t + 1, std::integral_constant<int, 0>{});
                        if (!SPREAD && req_new) issue(t + NSLOT);
                    }
                }
                sfor<2>([&](auto FT) {
                    constexpr int ft = FT.value;
#pragma unroll
                    for (int rt = 0; rt < RTN; ++rt) acc[ft][rt] = mfma32(as_bf(wf[b][ft]), as_bf(xf[b][rt]), acc[ft][rt]);
                    if constexpr (SPREAD) {
                        // request slots j = 0, 1: the two halves of the hand-over k-step; j = 2 + 2 u + ft: k-step u of the next stage
                        if constexpr (u == KU - 1) {
                            if (req_new) sfor<PPU>([&](auto J) { issue_piece(t + NSLOT, std::integral_constant<int, ft * PPU + J.value>{}); });
                        } else {
                            if (req_old) sfor<PPU>([&](auto J) { issue_piece(t - 1 + NSLOT, std::integral_constant<int, (2 + 2 * u + ft) * PPU + J.value>{}); });
                        }
                    }
                });
                // (without it hipcc sinks this step's MFMAs below the NEXT step's wait -- nothing but data ties an MFMA to an asm
                //  statement -- and the fragment reads running one step ahead would cover nothing)
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        __builtin_amdgcn_s_barrier();                                       // the ring is free: it becomes the staging area
        VISG_STAMP(2)

        // ---- epilogue.  Register r = 4 q + j of tile (ft, rt) of lane (n, kh) is feature 64 wn + 32 ft + 8 q + 4 kh + j of row HB wm + 32 rt + n.
        const uint32_t stg = lds0 + (uint32_t)(wave * HB * 128);
        if (Rp) {
#pragma unroll
            for (int j = 0; j < HB / 8; ++j) {
                const int rw = 8 * j + (lane_v >> 3);
                int64_t rg = row0 + wm * HB + rw;
                if (rg >= a.M) rg = a.M - 1;
                const uint32_t off = (uint32_t)rg * (uint32_t)D2 + (uint32_t)((member * 256 + wn * 64) * 2 + (((lane_v & 7) ^ (rw & 7)) * 16));
                glds16_row(sbase(Rp) + off, smem + (size_t)wave * HB * 128 + j * 1024);
            }
        }
        const float* prm = reinterpret_cast<const float*>(smem + GEO::PRM_OFF);
        const int fl0 = wn * 64 + 4 * kh;                                   // + 32 ft + 8 q: the lane's four contiguous features
        sfor<2>([&](auto FT) {
            sfor<4>([&](auto Q) {
                u32x4 braw;             // (inline asm: hipcc would drain the R rows in flight before an LDS read it can see)
                lds_read16<(32 * FT.value + 8 * Q.value) * 4>(braw, lds0 + (uint32_t)(GEO::PRM_OFF + fl0 * 4));
                lgkm_fence(braw);
                const f32x4 b4 = __builtin_bit_cast(f32x4, braw);
#pragma unroll
                for (int rt = 0; rt < RTN; ++rt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[FT.value][rt][4 * Q.value + j] += b4[j];
            });
        });
#pragma unroll
        for (int rt = 0; rt < RTN; ++rt) {
            float mw = 0.f, q = 0.f;
            if (!a.rms) {
                float s = 0.f;
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += acc[ft][rt][r];
                s += __shfl_xor(s, 32, 64);
                mw = s * (1.0f / 64.0f);
            }
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float c = acc[ft][rt][r] - mw; q += c * c; }
            q += __shfl_xor(q, 32, 64);
            if (kh == 0) {              // (inline asm, like every LDS access up to the second barrier: the R rows are in flight, and hipcc
                const u32x2 v2 = {__float_as_uint(mw), __float_as_uint(q)};     //  drains them before an LDS access it can see)
                lds_write8<0>(lds0 + (uint32_t)(GEO::WST_OFF + ((wm * HB + 32 * rt + n) * 4 + wn) * 8), v2);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        VISG_STAMP(3)
        const float2* rst = reinterpret_cast<const float2*>(smem + GEO::RST_OFF);
        if (tid < BM) {
            // this workgroup's statistics of row tid over its 256 features (Chan's combination of the four waves' 64)
            float mj = 0.f, m2 = 0.f;
            float2 w4[4];
            {
                u32x4 ra, rb2;
                lds_read16<0>(ra, lds0 + (uint32_t)(GEO::WST_OFF + tid * 32));
                lds_read16<16>(rb2, lds0 + (uint32_t)(GEO::WST_OFF + tid * 32));
                lgkm_fence(ra); lgkm_tie(rb2);
                w4[0] = make_float2(__uint_as_float(ra[0]), __uint_as_float(ra[1])); w4[1] = make_float2(__uint_as_float(ra[2]), __uint_as_float(ra[3]));
                w4[2] = make_float2(__uint_as_float(rb2[0]), __uint_as_float(rb2[1])); w4[3] = make_float2(__uint_as_float(rb2[2]), __uint_as_float(rb2[3]));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { mj += w4[k].x; m2 += w4[k].y; }
            mj *= 0.25f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float dm = w4[k].x - mj; m2 += 64.0f * dm * dm; }
            float mean = mj, M2 = m2;
            if (NT > 1) {
                typedef __attribute__((address_space(1))) unsigned long long gu64;
                const unsigned epoch = (unsigned)it + 1u;
                // granules [parity][team][consumer][producer][2 BM]: a producer writes one copy per consumer, a consumer zeroes its copies
                // once it has read them -- a clean (all-zero) area is left clean by every launch, so nothing has to be zeroed between
                // launches (the memset node + its boundary cost 3-4 us of a 70 us call)
                gu64* base = (gu64*)(a.xch) + ((size_t)((it & 1) * a.nteams + team) * NT * NT) * (BM * 2);
                // the same pair, once more, where nobody clears it: what the repair kernel combines for a workgroup that gave up
                a.stats[(size_t)(row0 + tid) * NT + member] = ((unsigned long long)__float_as_uint(m2) << 32) | __float_as_uint(mj);
                const unsigned long long g0 = ((unsigned long long)epoch << 32) | __float_as_uint(mj);
                const unsigned long long g1 = ((unsigned long long)epoch << 32) | __float_as_uint(m2);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (k < NT - 1) {
                        const int other = member + 1 + k < NT ? member + 1 + k : member + 1 + k - NT;
                        gu64* dst = base + ((size_t)other * NT + member) * (BM * 2) + 2 * tid;
                        __hip_atomic_store(dst, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                float pm[3] = {0.f, 0.f, 0.f}, pq[3] = {0.f, 0.f, 0.f};
                bool timed_out = false;
                for (unsigned spins = 0;; ++spins) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        if (k < NT - 1) {
                            const int other = member + 1 + k < NT ? member + 1 + k : member + 1 + k - NT;
                            gu64* g = base + ((size_t)member * NT + other) * (BM * 2) + 2 * tid;
                            const unsigned long long x0 = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const unsigned long long x1 = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = ok && (unsigned)(x0 >> 32) == epoch && (unsigned)(x1 >> 32) == epoch;
                            pm[k] = __uint_as_float((unsigned)x0); pq[k] = __uint_as_float((unsigned)x1);
                        }
                    }
                    if (a.spin_limit == 0xffffffffu) { timed_out = true; break; }            // (tests: every workgroup takes the give-up path)
                    if (__all(ok)) break;
                    if (spins >= a.spin_limit) { timed_out = true; break; }                  // (wave-uniform: every lane counts the same)
                    __builtin_amdgcn_s_sleep(4);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {                                               // consumed: this consumer's copies go back to zero
                    if (k < NT - 1) {
                        const int other = member + 1 + k < NT ? member + 1 + k : member + 1 + k - NT;
                        gu64* g = base + ((size_t)member * NT + other) * (BM * 2) + 2 * tid;
                        __hip_atomic_store(g, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(g + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (timed_out && lane == 0) visg_lds_write4(lds0 + (uint32_t)GEO::TMO_OFF, 1u);
                float ms = mj;
#pragma unroll
                for (int k = 0; k < 3; ++k) if (k < NT - 1) ms += pm[k];
                mean = ms / (float)NT;
                { const float dm = mj - mean; M2 = m2 + 256.0f * dm * dm; }
#pragma unroll
                for (int k = 0; k < 3; ++k) if (k < NT - 1) { const float dm = pm[k] - mean; M2 += pq[k] + 256.0f * dm * dm; }
            }
            if (a.rms) mean = 0.f;
            const float rstd = rsqrtf(M2 * inv_d + a.eps);
            {
                const u32x2 v2 = {__float_as_uint(mean), __float_as_uint(rstd)};
                lds_write8<0>(lds0 + (uint32_t)(GEO::RST_OFF + tid * 8), v2);
            }
            if (member == 0 && row0 + tid < a.M) {
                if (a.rstd) a.rstd[row0 + tid] = rstd;
                if (a.mean) a.mean[row0 + tid] = mean;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");         // this wave's R rows have landed in its staging area
        __builtin_amdgcn_s_barrier();
        VISG_STAMP(4)
        // A workgroup one of whose waves gave up: (mean, rstd) = (0, 1) for every row, so that what goes where xhat goes is the PRE-norm
        // tile (what goes to out is overwritten by the repair kernel); the tile is flagged and the call's timeout counter bumped.
        const bool gave_up = __builtin_amdgcn_readfirstlane(visg_lds_read4(lds0 + (uint32_t)GEO::TMO_OFF)) != 0u;
        if (gave_up && tid == 0) {
            typedef __attribute__((address_space(1))) unsigned gu32;
            __hip_atomic_store((gu32*)(a.flags + (size_t)rb * NT + member), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(a.status, 1u);
            atomicAdd(a.status + 1, 1u);
        }
        // normalise in registers; out = xhat * gamma + beta + R goes through the staging area (in place over R)
        const uint32_t a_st = stg + (uint32_t)(n * 128 + 8 * kh);           // + 32 rt rows, slot (4 ft + q) ^ (n & 7)
        sfor<RTN>([&](auto RT_) {
            constexpr int rt = RT_.value;
            float2 ms = rst[wm * HB + 32 * rt + n];
            if (gave_up) ms = make_float2(0.f, 1.f);
            sfor<2>([&](auto FT) {
                sfor<4>([&](auto Q) {
                    constexpr int ft = FT.value, q = Q.value;
                    const uint32_t ad = a_st + (uint32_t)(rt * 32 * 128) + (uint32_t)((((4 * ft + q) ^ (n & 7))) * 16);
                    u32x2 r2 = {0u, 0u};
                    if (Rp) { lds_read8<0>(r2, ad); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r2) :: "memory"); }
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(prm + 256 + fl0 + 32 * ft + 8 * q);
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(prm + 512 + fl0 + 32 * ft + 8 * q);
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xh = (acc[ft][rt][4 * q + j] - ms.x) * ms.y;
                        acc[ft][rt][4 * q + j] = xh;
                        const float rv = (j & 1) ? bf_hi(r2[j >> 1]) : bf_lo(r2[j >> 1]);
                        o[j] = xh * g4[j] + b4[j] + rv;
                    }
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                    const bf16x4 t4 = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
                    lds_write8<0>(ad, __builtin_bit_cast(u32x2, t4));
                });
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        auto store_rows = [&](uint8_t* dstp) {                              // the wave's HB x 128 B of the staging area as whole lines
            const uint64_t ub = reinterpret_cast<uint64_t>(dstp);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)ub), hi = __builtin_amdgcn_readfirstlane((uint32_t)(ub >> 32));
            const uint64_t gb = ((uint64_t)hi << 32) | lo;
            sfor<HB / 32>([&](auto JB) {                                    // four 8-row pieces per batch: one LDS wait for four reads
                u32x4 v[4];
                sfor<4>([&](auto K) { lds_read16<(JB.value * 4 + K.value) * 1024>(v[K.value], stg + (uint32_t)(lane_v * 16)); });
                lgkm_fence(v[0]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k) lgkm_tie(v[k]);
                    const int rw = 8 * (JB.value * 4 + k) + (lane_v >> 3);
                    const int64_t rg = row0 + wm * HB + rw;
                    if (rg < a.M) {
                        const uint32_t off = (uint32_t)rg * (uint32_t)D2 + (uint32_t)((member * 256 + wn * 64) * 2 + (((lane_v & 7) ^ (rw & 7)) * 16));
                        typedef __attribute__((address_space(1))) u32x4 g_u32x4;
                        *reinterpret_cast<g_u32x4*>(gb + off) = v[k];
                    }
                }
            });
        };
        store_rows(Op);
        VISG_STAMP(5)
        if (Hp) {
            sfor<RTN>([&](auto RT_) {
                constexpr int rt = RT_.value;
                sfor<2>([&](auto FT) {
                    sfor<4>([&](auto Q) {
                        constexpr int ft = FT.value, q = Q.value;
                        const uint32_t ad = a_st + (uint32_t)(rt * 32 * 128) + (uint32_t)((((4 * ft + q) ^ (n & 7))) * 16);
                        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                        const bf16x4 t4 = {(__bf16)acc[ft][rt][4 * q], (__bf16)acc[ft][rt][4 * q + 1], (__bf16)acc[ft][rt][4 * q + 2], (__bf16)acc[ft][rt][4 * q + 3]};
                        lds_write8<0>(ad, __builtin_bit_cast(u32x2, t4));
                    });
                });
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            store_rows(Hp);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                                    // the staging area becomes the ring of the next row block
        VISG_STAMP(6)
    }
}

// ------------------------------------------------------------------------------------------------ repair (round 6)
// Launched behind every main kernel by the same call.  Header words of the workspace: [0] status (sticky, for the host), [1] tiles that
// gave up so far (monotonic), [2] that count at the last repair, [3] repair workgroups finished.  Normal case: [1] == [2], every
// workgroup returns after two loads.  Otherwise the flagged tiles are normalised from the pre-norm rows the main kernel left where
// xhat goes (bf16: the rounding point of the library composition), with the statistics of ALL column tiles from the plain array --
// complete by now, combined in the order the main kernel uses for that member -- and the exchange area is zeroed; the last
// workgroup to finish records the count.
__global__ __launch_bounds__(256) void visproj_gemm_repair_kernel(VisGemmArgs a, int BM, size_t xch_words) {
    typedef __attribute__((address_space(1))) unsigned gu32;
    gu32* hdr = (gu32*)a.status;
    const unsigned given_up = __hip_atomic_load(hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned repaired = __hip_atomic_load(hdr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (given_up == repaired) return;
    const int NT = a.d_out >> 8, tid = threadIdx.x;
    const float inv_d = 1.0f / (float)a.d_out;
    for (int t = blockIdx.x; t < a.row_blocks * NT; t += gridDim.x) {
        if (a.flags[t] == 0u) continue;
        const int rb = t / NT, member = t % NT;
        const int64_t row0 = (int64_t)rb * BM;
        const int c8 = (tid & 31) * 8, f0 = member * 256 + c8;            // this thread's 8 features (16 bytes) of a row
        float g8[8], b8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { g8[j] = a.gamma[f0 + j]; b8[j] = a.beta ? a.beta[f0 + j] : 0.f; }
        for (int r = tid >> 5; r < BM; r += 8) {
            const int64_t row = row0 + r;
            if (row >= a.M) break;
            float pmj[4], pm2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < NT) {
                    const int who = member + k < NT ? member + k : member + k - NT;     // self first, then member + 1, .. (the main kernel's order)
                    const unsigned long long v = a.stats[(size_t)row * NT + who];
                    pmj[k] = __uint_as_float((unsigned)v); pm2[k] = __uint_as_float((unsigned)(v >> 32));
                }
            float ms = pmj[0];
#pragma unroll
            for (int k = 1; k < 4; ++k) if (k < NT) ms += pmj[k];
            float mean = ms / (float)NT, M2;
            { const float dm = pmj[0] - mean; M2 = pm2[0] + 256.0f * dm * dm; }
#pragma unroll
            for (int k = 1; k < 4; ++k) if (k < NT) { const float dm = pmj[k] - mean; M2 += pm2[k] + 256.0f * dm * dm; }
            if (a.rms) mean = 0.f;
            const float rstd = rsqrtf(M2 * inv_d + a.eps);
            const size_t off = (size_t)row * a.d_out + f0;
            const bf16x8 pre = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.xhat) + off);
            bf16x8 rr;
#pragma unroll
            for (int j = 0; j < 8; ++j) rr[j] = (__bf16)0.f;
            if (a.R) rr = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.R) + off);
            bf16x8 xo, oo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = ((float)pre[j] - mean) * rstd;
                xo[j] = (__bf16)xh;
                oo[j] = (__bf16)(xh * g8[j] + b8[j] + (float)rr[j]);
            }
            *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(a.xhat) + off) = xo;
            *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(a.out) + off) = oo;
            if (member == 0 && c8 == 0) {
                if (a.rstd) a.rstd[row] = rstd;
                if (a.mean) a.mean[row] = mean;
            }
        }
        __syncthreads();
        if (tid == 0) a.flags[t] = 0u;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < xch_words; i += (size_t)gridDim.x * 256) a.xch[i] = 0ull;
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        if (atomicAdd(a.status + 3, 1u) == gridDim.x - 1) {
            __hip_atomic_store(hdr + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(hdr + 2, given_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
bool visproj_gemm_applies(int64_t M, int F, int d_out, int io_fp32) {
    if (io_fp32 || M <= 0 || d_out % 256 != 0 || d_out / 256 > 4 || F % 64 != 0 || F < 64) return false;
    const int64_t wide = F > d_out ? F : d_out;
    if ((M + 127) / 128 * (d_out / 256) > 65536 / 4) return false;      // (one give-up flag per 128-row tile: VISG_FLAGS_B)
    return M * wide * 2 < ((int64_t)1 << 32);           // (32-bit per-lane byte offsets)
}
static int visg_teams_max(int d_out) { return cols_groups_max(d_out / 256); }
// rows per workgroup (128 / 192 / 256): the one that needs the least (weighted) passes of the teams over the row blocks -- a workgroup's
// time is close to affine in its rows (profiles/r05_k4bench.txt: 18 us + 0.2 us per row), and a pass costs a whole workgroup time
// whatever its fill.  Ties go to the larger tile (less L2 -> LDS traffic per flop).
static int visg_pick_bm(int64_t M, int d_out, int forced) {
    if (forced == 128 || forced == 192 || forced == 256) return forced;
    const int tmax = visg_teams_max(d_out);
    int best = 256; double bc = 1e30;
    for (int bm : {256, 192, 128}) {
        const double passes = (double)(((M + bm - 1) / bm + tmax - 1) / tmax);
        const double c = passes * (18.0 + 0.2 * bm);
        if (c < bc - 1e-9) { bc = c; best = bm; }
    }
    return best;
}
// workspace: [0, 256) header (status, counters, stamps) | exchange area | tile flags [<= 16,384] u32 (64 KiB) | statistics [M + 256][NT] x 8 B
size_t visproj_gemm_exchange_bytes(int d_out) {
    const size_t NT = (size_t)(d_out / 256);
    return (size_t)2 * visg_teams_max(d_out) * NT * NT * 256 * 2 * 8;
}
// (the flag area has ONE size whatever M: callers share a workspace between launches of different sizes, and a smaller launch's
//  statistics must never land where a larger launch looks for its flags)
size_t visproj_gemm_workspace_bytes(int64_t M, int F, int d_out) {
    if (!visproj_gemm_applies(M, F, d_out, 0)) return 0;
    return 256 + visproj_gemm_exchange_bytes(d_out) + VISG_FLAGS_B + (size_t)(M + 256) * (size_t)(d_out / 256) * 8;
}

template <int BM, int BK, int NSLOT, bool SPREAD>
static hipError_t launch_visg(VisGemmArgs& a, uint8_t* ws, hipStream_t stream) {
    using GEO = VisGemmGeo<BM, BK, NSLOT>;
    const int NT = a.d_out / 256, tmax = visg_teams_max(a.d_out);
    a.row_blocks = (int)((a.M + BM - 1) / BM);
    a.nteams = a.row_blocks < tmax ? a.row_blocks : tmax;
    a.status = reinterpret_cast<unsigned*>(ws);
    a.xch = reinterpret_cast<unsigned long long*>(ws + 256);
    const size_t xch_b = visproj_gemm_exchange_bytes(a.d_out);
    a.flags = reinterpret_cast<unsigned*>(ws + 256 + xch_b);
    a.stats = reinterpret_cast<unsigned long long*>(ws + 256 + xch_b + VISG_FLAGS_B);
    if (a.spin_limit == 0) a.spin_limit = VISG_MAX_SPIN;
    auto kern = visproj_gemm_kernel<BM, BK, NSLOT, SPREAD>;
    // residency: a team's members must run concurrently, which one workgroup per CU and a grid no larger than the device's CU count
    // guarantee (a smaller device gets fewer teams and more passes, never a grid it cannot hold)
    {
        int dev = 0, cus = 0;
        hipError_t eq = hipGetDevice(&dev);
        if (eq == hipSuccess) eq = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (eq != hipSuccess) return eq;
        while (a.nteams > 1 && (int)cols_grid(NT, a.nteams) > cus) --a.nteams;
        if ((int)cols_grid(NT, a.nteams) > cus) return hipErrorInvalidConfiguration;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GEO::LDS_B);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(cols_grid(NT, a.nteams)), dim3(512), GEO::LDS_B, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess || NT == 1) return e;
    // the repair pass of the same call: two loads and out unless a workgroup gave up (see the kernel)
    const int tiles = a.row_blocks * NT;
    // (16 workgroups: the normal case is "two loads and out", and a 213-workgroup launch of that took 4.8 us -- profiles/r06_pmc_traffic.md)
    hipLaunchKernelGGL(visproj_gemm_repair_kernel, dim3(tiles < 16 ? tiles : 16), dim3(256), 0, stream, a, BM, xch_b / 8);
    return hipGetLastError();
}

// form: 0 = default, 1 = BK 64 / two slots, 2 = BK 32 / four slots, 3 = BK 32 / three slots, 4-6 = the same with the requests of
// the stage ahead spread between the MFMA groups; bm: 0 = by shape
hipError_t launch_visproj_gemm(VisGemmArgs& a, void* ws, int form, int bm, hipStream_t stream) {
    uint8_t* w8 = reinterpret_cast<uint8_t*>(ws);
    // (tests of the give-up path: bits 8.. = log2 of the polls before a wave gives up; 31: one poll; 30: every workgroup gives up at once)
    if (form >> 8) a.spin_limit = ((form >> 8) & 31) == 31 ? 1u : ((form >> 8) & 31) == 30 ? 0xffffffffu : 1u << ((form >> 8) & 31);
    form &= 255;
    if (form == 0) form = VISG_DEFAULT_FORM;
    int BMv = visg_pick_bm(a.M, a.d_out, bm);
    if (BMv == 192 && form != 1 && form != 4) {         // (192-row tiles exist for the 64-feature stages only)
        if (bm == 192) return hipErrorInvalidValue;
        BMv = visg_pick_bm(a.M, a.d_out, 0) == 192 ? 256 : BMv;
    }
#define VISG_GO(BK_, NS_, SP_) return BMv == 256 ? launch_visg<256, BK_, NS_, SP_>(a, w8, stream) : launch_visg<128, BK_, NS_, SP_>(a, w8, stream)
#define VISG_GO3(NS_, SP_) return BMv == 256 ? launch_visg<256, 64, NS_, SP_>(a, w8, stream) : BMv == 192 ? launch_visg<192, 64, NS_, SP_>(a, w8, stream) : launch_visg<128, 64, NS_, SP_>(a, w8, stream)
    switch (form) {
        case 1: VISG_GO3(2, false);
        case 2: VISG_GO(32, 4, false);
        case 3: VISG_GO(32, 3, false);
        case 4: VISG_GO3(2, true);
        case 5: VISG_GO(32, 4, true);
        case 6: VISG_GO(32, 3, true);
        default: return hipErrorInvalidValue;
    }
#undef VISG_GO
#undef VISG_GO3
}
