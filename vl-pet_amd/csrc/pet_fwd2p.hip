// K1 forward (training form) as two passes that hold the weights still (round 4):
//     out = ( s2*x2 + sd*up_A(gelu_new(down_A(x2))) ) (*|+) sigmoid( up_G(gelu_new(down_G(x1))) ) * gs
// (my_transformers/modeling_bart.py:1147-1155, 1195-1209, 1256-1257; T5: my_transformers/modeling_t5.py:366-390, 782-806).
//
// Why.  The one-kernel forward (pet_gate_fwd.hip) is row-parallel: every 128-row workgroup streams both chains' packed weights
// (0.59 MB at r = 96, 1.2 MB at r = 192) through its LDS and synchronises at each of its 25 (49) stages, so it is a latency chain
// whose length does not depend on its rows -- 31 us for 28 workgroups, 117 us at r = 192.  The forward
// of the training path has to leave z = gelu_new(pre) and gelu_new'(pre) of both chains for the backward anyway, and everything
// after z is separable by COLUMN, so the op is cut there:
//
//   pass A (k1_down_kernel): z_c = gelu_new(b_c + W_c x_c) for chain c = adapter (x2) / gate (x1).  A workgroup owns ONE chain and
//     a block of rows; wave ct holds the [32 x d] slice of the down weight of c-tile ct as MFMA A fragments in registers for the
//     whole launch (192 registers at d = 768) and the rows stream past it: a loader wave moves [32 rows x d/2] sub-steps into an
//     LDS ring with global_load_lds (whole 128-byte lines, two sub-steps ahead), the compute waves read B fragments (lane = row)
//     and run 24 MFMAs per sub-step, bias + gelu_new (+ derivative) + the z / gelu' stores after every second one.  No weight
//     stream, one barrier per sub-step, row blocks of any multiple of 32 rows (no 128-row round quantisation).
//   pass B (k1_up_kernel): column-parallel.  A wave owns 32 output columns and keeps Wu / Wgu of them in registers (48 at r = 96);
//     a workgroup = 8 waves = 256 columns and a chunk of rows; per 32-row step the x2 columns and the two z tiles arrive by
//     global_load_lds, each wave runs its two up projections (2 KT MFMAs), the sigmoid / residual / product in registers, and
//     stores its 64 bytes per row.  The column groups of a row chunk share an XCD (they re-read the same z rows through its L2).
//
// Traffic: x1, x2 once in pass A, z / gelu' written (they are outputs of the training form), x2 again + z back in pass B, out
// written: 4 row units + 2.5 bottleneck units instead of 3 + 1 -- but no per-workgroup weight stream (129 MB of L2->LDS traffic at
// M = 28,000, 169 MB at r = 192 / 18,250 rows) and no 25-stage chain.  bf16, d = 768; everything else stays on
// pet_gate_fwd.hip / pet_fwd.hip.
//
// The forms without a gate take the same two passes (round 4, second session): K2 (adapters/adapter_modeling.py:55-61,
// adapter_controller.py:149-162: out = y + s*up(gelu_new(down(x)))) and K3 (lora/controller.py:56-70: out = base +
// (dropout(x) A^T B^T) * alpha / r).  Pass A runs ONE chain (grid = row blocks), with the identity instead of gelu_new for K3;
// pass B has no gate projection and adds the residual input (y / base) instead of x2.  For them the cut costs almost nothing in
// traffic -- x is not needed after pass A, so the op still moves 3 row units + z twice -- and it removes what made the
// one-kernel forward slow outside one well-filled round: the 24-stage chain of a 128-row workgroup (decoder-side LoRA calls of
// 2,500-10,000 rows: 18 of the 36 calls of a step) and the second round above 32,768 rows.
// K3's dropout: the keep flags no longer come out of the Philox generator INSIDE the forward (13 us of a 42 us kernel at
// M = 28,000: ~100 VALU instructions per 8 elements in the one wave that also runs the MFMAs).  drop_bits_kernel writes the
// packed mask the backward wants anyway (rng.h drop_pos layout, 1 bit per element, M*d/8 bytes) as a plain elementwise launch
// over the whole chip; pass A's loader brings each sub-step's mask words along with the rows (one 4-byte global_load_lds per
// lane and stage) and the compute waves clear the dropped elements of their B fragments; 1/(1-p) goes onto the sums.
#include "cols_common.h"
#include "rng.h"

namespace {

constexpr int F2_D = 768;
constexpr int F2_G = F2_D / 16;          // MFMA k-steps of the down projection
constexpr int F2_NST = F2_D / 64;         // 64-feature stages of a row

// diagnosis build (-DVLPET_F2_STAMPS): shader-clock time per phase of one wave of a few workgroups, summed over its steps
#ifdef VLPET_F2_STAMPS
#define F2_STAMP_DECL unsigned long long f2_acc[4] = {0, 0, 0, 0}, f2_last = __builtin_readcyclecounter(); const unsigned long long f2_t0 = f2_last;
#define F2_STAMP(k) { const unsigned long long tn = __builtin_readcyclecounter(); f2_acc[k] += tn - f2_last; f2_last = tn; }
#define F2_STAMP_PRINT(tag, fmt) if (lane == 0 && (wave == 0 || wave == (int)(blockDim.x >> 6) - 1) && (blockIdx.x == 0 || blockIdx.x == 101 || blockIdx.x == 303)) \
    printf("f2 stamps " tag " blk %d wave %d (cycles, total %llu): " fmt "\n", (int)blockIdx.x, wave, __builtin_readcyclecounter() - f2_t0, f2_acc[0], f2_acc[1], f2_acc[2]);
#else
#define F2_STAMP_DECL
#define F2_STAMP(k)
#define F2_STAMP_PRINT(tag, fmt)
#endif

template <int N> __device__ __forceinline__ void vmw() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// cols_decode / cols_grid of cols_common.h with S workgroup slots per XCD instead of 32 (S = 64: two workgroups per CU): a group =
// the U column groups of a row chunk, placed on one XCD (block b runs on XCD b % 8) so that they re-read its z rows through one L2
__host__ __device__ inline int f2_groups_max(int U, int S) { const int G = S / U; return 8 * G + ((S - U * G) * 8) / U; }
__host__ __device__ inline void f2_decode(int b, int U, int S, int& group, int& member) {
    const int G = S / U, main = U * G, x = b & 7, j = b >> 3;
    if (j < main) { member = j % U; group = (j / U) * 8 + x; }
    else { const int l = (j - main) * 8 + x; group = 8 * G + l / U; member = l % U; }
}
__host__ __device__ inline unsigned f2_grid(int U, int S, int ngroups) {
    const int G = S / U;
    if (ngroups <= 8 * G) return 8u * (unsigned)U * (unsigned)((ngroups + 7) / 8);
    return 8u * (unsigned)(U * G + ((ngroups - 8 * G) * U + 7) / 8);
}

// ------------------------------------------------------------------------------------------------ pass A
// SSN = stages per sub-step (ring slot = 32 rows x 64 SSN features): 6 -> two sub-steps per row tile, 3 or 4 slots; 3 -> four sub-steps,
// 6 slots (the same LDS, 5/6 of it in flight instead of 2/3)
template <int RT, int SSN, bool DROP = false> struct DownGeo {
    static constexpr int SUB_B = SSN * 4096;
    static constexpr int NPT = F2_NST / SSN;            // sub-steps per row tile
    static constexpr int NSLOT = SSN == 6 ? (RT == 6 ? 4 : 2) : (RT == 6 ? 6 : 5);
    static constexpr int NI = (4 + (DROP ? 1 : 0)) * SSN;   // global_load_lds instructions per sub-step (DROP: + the mask words of each stage)
    static constexpr int PB = 64 * RT;                  // bytes of a bottleneck row
    static constexpr int STG_OFF = NSLOT * SUB_B;       // staging of the z and gelu' tiles of a row tile: 2 x [32 rows x PB]
    static constexpr int BIAS_OFF = STG_OFF + 2 * 32 * PB;
    static constexpr int BITS_OFF = BIAS_OFF + 32 * RT * 4;          // DROP: NSLOT x SSN x [32 rows x 8 mask bytes]
    static constexpr int LDS = BITS_OFF + (DROP ? NSLOT * SSN * 256 : 0);
    static constexpr int NLD = RT == 6 ? 2 : 1;         // loader waves: at six tiles (one workgroup per CU, 13 of a wave's 41 k cycles spent at the
                                                        // stage barrier behind ONE wave issuing 24 LDS-DMA instructions per sub-step) two share the rows
    static constexpr int THREADS = (RT + NLD) * 64;
};

// NCH = chains (2: the gated K1, block parity = chain; 1: K2 / K3); ACT_ID: identity instead of gelu_new (K3; z only is saved);
// DROP: a.drop.bits = packed keep flags of the chain input (K3)
template <int RT, int SSN, int NCH, bool ACT_ID, bool DROP>
__global__ __launch_bounds__((RT + (RT == 6 ? 2 : 1)) * 64, 2) void k1_down_kernel(PetFwdArgs a, int rows_per_block, int write_grad) {
    using GEO = DownGeo<RT, SSN, DROP>;
    constexpr int NSLOT = GEO::NSLOT, NPT = GEO::NPT, SUB_B = GEO::SUB_B, NI = GEO::NI, PB = GEO::PB, NLD = GEO::NLD, QN = 4 / NLD;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chain = NCH == 2 ? (int)(blockIdx.x & 1) : 0, rb = NCH == 2 ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    const int64_t r_begin = (int64_t)rb * rows_per_block;
    int64_t r_end = r_begin + rows_per_block;
    if (r_end > a.M) r_end = a.M;
    const int ntiles = (int)((r_end - r_begin + 31) >> 5);
    const int nss = NPT * ntiles;
    const uint8_t* x = reinterpret_cast<const uint8_t*>(chain == 0 ? a.xa : a.xg);
    const uint8_t* pk = chain == 0 ? a.pk_a : a.pk_g;
    const PackGeom pg = pack_geom(RT, F2_D, 1);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;

    if (wave >= RT) {
        // ---------------------------------------------------------------- loader wave(s): every global_load_lds of the workgroup
        // (NLD = 2: loader ld brings the row groups q = 2 ld, 2 ld + 1 of every stage; the mask words travel with loader 0)
        const int ld = wave - RT;
        // sub-step i = (row tile i / NPT, feature part i % NPT): SSN stages x 4 instructions of 8 rows x 128 B; slot = piece ^ swz(row)
        uint32_t roff[QN];
#pragma unroll
        for (int i = 0; i < QN; ++i) {
            const int row = 8 * (ld * QN + i) + (lane >> 3);
            roff[i] = (uint32_t)row * (uint32_t)(F2_D * 2) + (uint32_t)(((lane & 7) ^ swz(row)) * 16);
        }
        auto issue = [&](int i) {
            const int64_t row0 = r_begin + 32 * (int64_t)(i / NPT);
            const int last = (int)(r_end - row0) - 1;                     // rows past the end re-read the last row
            const uint8_t* src = x + row0 * (F2_D * 2) + (i % NPT) * (SSN * 128);
            uint8_t* dst = smem + (size_t)(i % NSLOT) * SUB_B;
#pragma unroll
            for (int qq = 0; qq < QN; ++qq) {
                const int q = ld * QN + qq;
                const int row = 8 * q + (lane >> 3);
                const uint32_t o = roff[qq] - (uint32_t)(row > last ? row - last : 0) * (uint32_t)(F2_D * 2);
#pragma unroll
                for (int st = 0; st < SSN; ++st) glds16_row(src + o + st * 128, dst + st * 4096 + q * 1024);
            }
            if (DROP && ld == 0) {
                // mask words of the sub-step: lane L brings the dword of (row L / 2, half L % 2) of each stage -- the four bytes
                // of the k-steps u = 0..3 of compute lane (m = row, h = half) in the packed layout (rng.h drop_pos)
                int brow = lane >> 1;
                if (brow > last) brow = last;
                const uint8_t* bsrc = a.drop.bits + (row0 + brow) * (int64_t)(F2_D / 8) + (i % NPT) * (SSN * 8) + (lane & 1) * 4;
                uint8_t* bdst = smem + GEO::BITS_OFF + (size_t)(i % NSLOT) * (SSN * 256);
#pragma unroll
                for (int st = 0; st < SSN; ++st)
                    __builtin_amdgcn_global_load_lds((gmem_cv*)(bsrc + st * 8), (lmem_v*)(bdst + st * 256), 4, 0, 0);
            }
        };
#pragma unroll
        for (int i = 0; i < NSLOT - 1; ++i)
            if (i < nss) issue(i);
        F2_STAMP_DECL
        for (int i = 0; i < nss; ++i) {
            int ahead = nss - 1 - i;                                      // sub-steps already requested beyond i
            if (ahead > NSLOT - 2) ahead = NSLOT - 2;
            if constexpr (NLD > 1) vm_wait(ahead * (QN * SSN + ((DROP && ld == 0) ? SSN : 0)));    // (this loader's instructions per sub-step)
            else switch (ahead) {
                case 0: vmw<0>(); break;
                case 1: vmw<NI>(); break;
                case 2: vmw<2 * NI>(); break;
                case 3: vmw<(3 * NI <= 63 ? 3 * NI : 63)>(); break;
                default: vmw<(4 * NI <= 63 ? 4 * NI : 63)>(); break;
            }
            F2_STAMP(0)
            __builtin_amdgcn_s_barrier();                                 // sub-step i has landed; the slot of sub-step i - 1 is free
            F2_STAMP(1)
            if (i + NSLOT - 1 < nss) issue(i + NSLOT - 1);
            F2_STAMP(2)
        }
        __builtin_amdgcn_s_barrier();                                     // (the compute waves' hand-over of the last staged tile)
        F2_STAMP_PRINT("down loader", "vmcnt wait %llu  barrier %llu  issue %llu")
        return;
    }

    // -------------------------------------------------------------------- compute wave ct: resident A fragments of its c-tile
    const int ct = wave, m = lane & 31, h = lane >> 5;
    bf16x8 wf[F2_G];
#pragma unroll
    for (int g = 0; g < F2_G; ++g)
        wf[g] = *reinterpret_cast<const bf16x8*>(pk + ((size_t)(g * RT + ct) * 64 + lane) * 16);
    {   // down bias of the chain -> LDS (fp32)
        float* sb = reinterpret_cast<float*>(smem + GEO::BIAS_OFF);
        const float* bsrc = reinterpret_cast<const float*>(pk + pg.bias_off);
        for (int i = tid; i < 32 * RT; i += 64 * RT) sb[i] = bsrc[i];
    }
    // B fragment of k-step u of a stage: 16 bytes at piece 2u + h of row m
    uint32_t a_b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a_b[u] = lds0 + (uint32_t)(m * 128 + (((2 * u + h) ^ swz(m)) * 16));
    const uint32_t a_bias = lds0 + (uint32_t)(GEO::BIAS_OFF + (32 * ct + 8 * h) * 4);
    const uint32_t a_kw = lds0 + (uint32_t)(GEO::BITS_OFF + (2 * m + h) * 4);      // (DROP) the lane's mask word of a stage
    const float kscale = DROP ? a.drop.keep_scale : 1.0f;
    __bf16* sv_z = reinterpret_cast<__bf16*>(reinterpret_cast<uint8_t*>(a.save) + (size_t)(chain == 0 ? 0 : 2) * a.save_stride);
    __bf16* sv_g = reinterpret_cast<__bf16*>(reinterpret_cast<uint8_t*>(a.save) + (size_t)(chain == 0 ? 1 : 3) * a.save_stride);

    // The z / gelu' values of a row tile are staged in LDS as the two [32 rows x PB] tiles they are in memory and leave, after the
    // next barrier, as contiguous 1 KiB chunks (whole 128-byte lines): written straight from the accumulator layout they were
    // 16-byte pieces of 32 rows per instruction -- as many memory transactions per tile as the whole row stream.
    const uint32_t a_stg = lds0 + (uint32_t)(GEO::STG_OFF + m * PB + (32 * ct + 8 * h) * 2);
    const int nchunk = write_grad ? 4 : 2;              // chunks of this wave: c = wave + RT k; tensor c / (2 RT), piece c % (2 RT)
    auto store_out = [&](int t) {
#ifdef VLPET_F2_ABL
        if (VLPET_F2_ABL & 4) return;
#endif
        const int64_t row0 = r_begin + 32 * (int64_t)t;
        const int valid = (int)(r_end - row0);
        sfor<2>([&](auto H) {                             // two chunks at a time (the resident weights leave few registers)
            constexpr int k0 = 2 * H.value;
            if (k0 < nchunk) {
                u32x4 v[2];
                sfor<2>([&](auto K) {
                    constexpr int k = k0 + K.value;
                    lds_read16<0>(v[K.value], lds0 + (uint32_t)(GEO::STG_OFF + (wave + RT * k) * 1024 + lane * 16));
                });
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]) :: "memory");
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int c = wave + RT * (k0 + kk), piece = c % (2 * RT);
                    const int off = piece * 1024 + lane * 16;
                    // (a fresh scalar base per use: otherwise hipcc hoists the 64-bit per-lane addresses out of the loop and spills them)
                    const uint64_t ub = reinterpret_cast<uint64_t>(reinterpret_cast<uint8_t*>(k0 < 2 ? sv_z : sv_g) + row0 * PB);
                    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)ub), hi = __builtin_amdgcn_readfirstlane((uint32_t)(ub >> 32));
                    typedef __attribute__((address_space(1))) u32x4 g_u32x4;           // (global, not flat: the pointer went through an integer)
                    g_u32x4* dst = reinterpret_cast<g_u32x4*>((((uint64_t)hi << 32) | lo) + (uint32_t)off);
                    if (off / PB < valid) *dst = v[kk];
                }
            }
        });
    };

    f32x16 acc = zero16();
    F2_STAMP_DECL
    int t = 0;
    auto substep = [&](int i, auto PART) {
        constexpr int hf = decltype(PART)::value;
        __builtin_amdgcn_s_barrier();
        F2_STAMP(0)
        if constexpr (hf == 0) { if (t > 0) store_out(t - 1); }
        const uint32_t sb = (uint32_t)((i % NSLOT) * SUB_B);
#ifdef VLPET_F2_ABL
        if (!(VLPET_F2_ABL & 1))
#endif
        sfor<SSN>([&](auto ST) {
            constexpr int st = ST.value;
            uint32_t kw = 0;
            if constexpr (DROP)
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(kw) : "v"(a_kw + (uint32_t)((i % NSLOT) * (SSN * 256))), "n"(st * 256) : "memory");
            // B fragments in batches: all four of a stage at once, two at a time under DROP (the masking temporaries next to the 192
            // resident weight registers spilled 8 registers into the loop with four in flight)
            constexpr int GB = DROP ? 2 : 4;
            sfor<4 / GB>([&](auto G) {
                u32x4 bf[GB];
                sfor<GB>([&](auto U) { lds_read16<st * 4096>(bf[U.value], a_b[G.value * GB + U.value] + sb); });
                lgkm_fence(bf[0]);
                if constexpr (DROP) asm volatile("" : "+v"(kw) :: "memory");
#pragma unroll
                for (int uu = 0; uu < GB; ++uu) {
                    const int u = G.value * GB + uu;
                    if (uu) lgkm_tie(bf[uu]);
                    if constexpr (DROP) {                     // clear the dropped elements: byte u of kw = the 8 features of k-step u
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int lo = ((int)(kw << (31 - 8 * u - 2 * q))) >> 31, hi = ((int)(kw << (30 - 8 * u - 2 * q))) >> 31;
                            bf[uu][q] &= __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x07060100u);
                        }
                    }
                    acc = mfma32(wf[(hf * SSN + st) * 4 + u], as_bf(bf[uu]), acc);
                }
            });
        });
#ifdef VLPET_F2_STAMPS
        asm volatile("s_nop 0" : "+v"(acc[15]));
#endif
        F2_STAMP(1)
    };
#pragma unroll 1
    for (; t < ntiles; ++t) {
        sfor<NPT>([&](auto P) { substep(NPT * t + P.value, P); });
        // bias + gelu_new (+ derivative): register 8*sh + j of the tile <-> c = 32ct + 16sh + 8h + j
#ifdef VLPET_F2_ABL
        if (!(VLPET_F2_ABL & 2))
#endif
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
            u32x4 b0, b1;
            lds_read16<0>(b0, a_bias + sh * 64);
            lds_read16<16>(b1, a_bias + sh * 64);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1) :: "memory");
            float v[8], gd[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float bj = __uint_as_float(j < 4 ? b0[j] : b1[j - 4]);
                const float xv = (DROP ? acc[8 * sh + j] * kscale : acc[8 * sh + j]) + bj;
                if constexpr (ACT_ID) { v[j] = xv; gd[j] = 1.0f; }
                else {
                    const float x2 = xv * xv;
                    const float u = VLPET_GELU_K * (xv + 0.044715f * xv * x2);
                    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.0f * 1.4426950408889634f * u));
                    v[j] = xv * s;
                    const float du = VLPET_GELU_K * (1.0f + 3.0f * 0.044715f * x2);
                    gd[j] = s + xv * s * (1.0f - s) * 2.0f * du;
                }
            }
            if (sh == 0) { lds_write16<0>(a_stg, pack8(v)); if constexpr (!ACT_ID) lds_write16<32 * PB>(a_stg, pack8(gd)); }
            else { lds_write16<32>(a_stg, pack8(v)); if constexpr (!ACT_ID) lds_write16<32 * PB + 32>(a_stg, pack8(gd)); }
        }
        acc = zero16();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the staged pieces are in LDS at the next barrier
        F2_STAMP(2)
    }
    __builtin_amdgcn_s_barrier();
    store_out(ntiles - 1);
    F2_STAMP_PRINT("down compute", "barrier %llu  reads+mfma %llu  epilogue %llu")
}

// ------------------------------------------------------------------------------------------------ pass B
// Two rings: the x2 columns come from HBM (a few microseconds under load) and are requested NX - 1 steps ahead, the z tiles are
// hits of the L2 that pass A just wrote and travel one step ahead -- with one ring of depth 2 the kernel waited out an HBM
// latency per 32-row step (25.8 us at M = 28,000: 4.3 us per step).
template <int RT, int NX_> struct UpGeo {
    static constexpr int NW = 8;                        // waves = 32-column quarters
    static constexpr int NX = NX_;                      // x2 ring depth
    static constexpr int NZ = 2;                        // z ring depth
    static constexpr int KT = 2 * RT;
    static constexpr int PB = 64 * RT;                  // bytes of a bottleneck row
    static constexpr int PT_B = 32 * PB;                // one bottleneck tile (32 rows)
    static constexpr int X_B = (NW / 2) * 4096;         // NW / 2 pair tiles [32 rows x 128 B] of x2
    static constexpr int Z_OFF = NX * X_B;
    static constexpr int BIAS_OFF = Z_OFF + NZ * 2 * PT_B;
    static constexpr int LDS = BIAS_OFF + 2 * NW * 32 * 4;
};

// GATE = false (K2 / K3): no gate projection; out = s2 * res + sd * (bu + Wu z), res = the residual input (y / base)
template <int RT, bool ADD, int WPE, int NXD, bool GATE>
__global__ __launch_bounds__(512, WPE) void k1_up_kernel(PetFwdArgs a, int row_chunks, int64_t rows_per_chunk) {
    using GEO = UpGeo<RT, NXD>;
    constexpr int NW = GEO::NW, NX = GEO::NX, NZ = GEO::NZ, KT = GEO::KT, PB = GEO::PB, PT_B = GEO::PT_B, X_B = GEO::X_B;
    constexpr int NCG = F2_D / (32 * NW);               // column groups
    constexpr int NPZ = (GATE ? 4 : 2) * RT;            // 1 KiB pieces of the bottleneck tiles of a step
    constexpr int NB = (NPZ + NW - 1) / NW;             // bottleneck pieces per wave (some waves one fewer)
    constexpr int GRP = WPE >= 4 ? (KT % 3 == 0 ? 3 : KT) : (KT > 6 ? 6 : KT);     // B fragments per LDS batch (what the register budget allows)
    static_assert(NZ == 2, "the counted waits below assume the z tiles travel one step ahead");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int rc, cg;
    f2_decode((int)blockIdx.x, NCG, 16 * WPE, rc, cg);
    if (rc >= row_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int pt = wave >> 1, nt = wave & 1;            // pair tile of the workgroup, 32-column half of it
    const int colg = cg * (32 * NW);                    // first column of the workgroup
    const int c0 = colg + 32 * wave;
    const int64_t r_begin = (int64_t)rc * rows_per_chunk;
    int64_t r_end = r_begin + rows_per_chunk;
    if (r_end > a.M) r_end = a.M;
    const int nsteps = r_end > r_begin ? (int)((r_end - r_begin + 31) >> 5) : 0;
    const PackGeom pg = pack_geom(RT, F2_D, 1);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    F2_STAMP_DECL

    // up biases of the workgroup's columns -> LDS: [bu_A (32 NW) | bu_G (32 NW)]
    {
        float* sbias = reinterpret_cast<float*>(smem + GEO::BIAS_OFF);
        const uint8_t* pkb = tid < 32 * NW ? a.pk_a : a.pk_g;
        if (GATE || tid < 32 * NW) sbias[tid] = reinterpret_cast<const float*>(pkb + pg.bias_off)[32 * RT + colg + (tid & (32 * NW - 1))];
    }
    // resident A fragments of the wave's 32 columns from the "up" packs, gathered so that a lane (m, h) ends with the 16 CONTIGUOUS
    // columns c0 + 16h .. + 15 of row m (the gather of pet_cols.hip: MFMA row i = column c0 + 16*((i>>2)&1) + 4*(i>>3) + (i&3))
    bf16x8 wA[KT], wG[GATE ? KT : 1];
    {
        const int cb = c0 >> 7, wc = (c0 >> 5) & 3, pp = wc >> 1, ntt = wc & 1;
        const int i = m, v = (i >> 2) & 1, ip = (i & 3) | (ntt << 2) | ((i >> 3) << 3);
        const int64_t off = pg.pack_bytes + (int64_t)(2 * cb + pp) * (4 * RT * 1024) + (int64_t)(v * KT) * 1024 + (ip + 32 * h) * 16;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            wA[ks] = *reinterpret_cast<const bf16x8*>(a.pk_a + off + ks * 1024);
            if constexpr (GATE) wG[ks] = *reinterpret_cast<const bf16x8*>(a.pk_g + off + ks * 1024);
        }
    }
    // ---- stage pieces of this wave (1 KiB each): x2 pieces q = wave, wave + NW (pair tile q / 4, rows 8 (q % 4) ..); bottleneck
    // pieces q' = wave + NW j < 4 RT (tensor q' / KT, piece q' % KT of the 32 contiguous rows)
    const uint8_t* x2p = reinterpret_cast<const uint8_t*>(a.res);
    const uint8_t* zbase[2] = {reinterpret_cast<const uint8_t*>(a.save), reinterpret_cast<const uint8_t*>(a.save) + 2 * a.save_stride};
    uint32_t xoff[2], xdst[2]; int xrow[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int q = wave + NW * k, t = q >> 2;
        xrow[k] = 8 * (q & 3) + (lane >> 3);
        xoff[k] = (uint32_t)xrow[k] * (uint32_t)(F2_D * 2) + (uint32_t)((colg + 64 * t) * 2 + (((lane & 7) ^ fsw(xrow[k])) * 16));
        xdst[k] = (uint32_t)(t * 4096 + (q & 3) * 1024);
    }
    uint32_t poff[NB], pdst[NB]; int prow[NB], pten[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int q = wave + NW * j, t = q / KT, piece = q % KT;
        const int sig = piece * 64 + lane;
        prow[j] = sig / (PB / 16);
        pten[j] = t;
        poff[j] = (uint32_t)(prow[j] * PB + ((sig % (PB / 16)) ^ gsw(prow[j])) * 16);
        pdst[j] = (uint32_t)(t * PT_B + piece * 1024);
    }
    auto sbase = [](const uint8_t* p) {
        const uint64_t u = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const uint8_t*>(((uint64_t)hi << 32) | lo);
    };
    auto issue_x = [&](int s) {
        const int64_t rb = r_begin + 32 * (int64_t)s;
        uint8_t* st = smem + (size_t)(s % NX) * X_B;
        const int last = (int)(r_end - rb) - 1;                               // rows past the end re-read the last row
#pragma unroll
        for (int k = 0; k < 2; ++k)
            glds16_row(sbase(x2p + rb * (F2_D * 2)) + xoff[k] - (uint32_t)(xrow[k] > last ? xrow[k] - last : 0) * (uint32_t)(F2_D * 2), st + xdst[k]);
    };
    auto issue_z = [&](int s) {
        const int64_t rb = r_begin + 32 * (int64_t)s;
        uint8_t* st = smem + GEO::Z_OFF + (size_t)(s % NZ) * (2 * PT_B);
        const int last = (int)(r_end - rb) - 1;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (wave + NW * j < NPZ)
                glds16(sbase(zbase[pten[j] ? 1 : 0] + rb * PB) + poff[j] - (uint32_t)(prow[j] > last ? prow[j] - last : 0) * PB, st + pdst[j]);
    };
    // per-lane LDS addresses (relative to the slot bases)
    uint32_t a_xcl[2], a_pbf[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        a_xcl[k] = (uint32_t)(pt * 4096 + m * 128 + (((4 * nt + 2 * h + k) ^ fsw(m)) * 16));    // columns 8k .. 8k+7 of the lane's 16 (row m)
        a_pbf[k] = (uint32_t)(m * PB + (((2 * k + h) ^ gsw(m)) * 16));                           // B fragment of row m, k-step 2j + k (+ 64 j)
    }
    const uint32_t a_bias = lds0 + (uint32_t)(GEO::BIAS_OFF + (32 * wave + 16 * h) * 4);
    const float gs = a.gs, s2g = a.s2 * gs, sdg = a.sd * gs;
    uint32_t a_ost[2];                                  // read-back of the staged outputs: row lane / 4 (+ 16), piece 4 nt + lane % 4
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rr = (lane >> 2) + 16 * j;
        a_ost[j] = (uint32_t)(pt * 4096 + rr * 128 + (((4 * nt + (lane & 3)) ^ fsw(rr)) * 16));
    }
    uint8_t* outp = reinterpret_cast<uint8_t*>(a.out) + (size_t)c0 * 2 + (lane & 3) * 16;

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // weights in registers, biases in LDS
    // request order (it is what the counted waits below rely on): slot t = -(NX - 1) .. : z of step t + 1, then x2 of step t + NX - 1
#pragma unroll
    for (int t = -(NX - 1); t < 0; ++t) {
        if (t + 1 >= 0 && t + 1 < nsteps) issue_z(t + 1);
        if (t + NX - 1 < nsteps) issue_x(t + NX - 1);
    }

#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        // younger than the z pieces of step s (requested in slot s - 1): the x2 pieces of step s + NX - 2 and this wave's two output
        // stores of step s - 1
        vm_wait((s + NX - 2 < nsteps ? 2 : 0) + (s > 0 ? 2 : 0));
        __builtin_amdgcn_s_barrier();                                     // step s has landed for every wave; the slots of step s - 1 are free
        F2_STAMP(0)
        if (s + 1 < nsteps) issue_z(s + 1);
        if (s + NX - 1 < nsteps) issue_x(s + NX - 1);
        const uint32_t sbx = lds0 + (uint32_t)((s % NX) * X_B);
        const uint32_t sbz = lds0 + (uint32_t)(GEO::Z_OFF + (s % NZ) * (2 * PT_B));
        // up projection of one chain, starting at its bias
        auto project_up = [&](auto TC, auto OC, const bf16x8* w, f32x16& acc) {
            constexpr int T = decltype(TC)::value;
            u32x4 bb[4], bf0[GRP];
            sfor<4>([&](auto Q) { lds_read16<decltype(OC)::value + 16 * Q.value>(bb[Q.value], a_bias); });
            sfor<GRP>([&](auto K) { lds_read16<T * PT_B + 64 * (K.value >> 1)>(bf0[K.value], sbz + a_pbf[K.value & 1]); });
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
                    lds_read16<T * PT_B + 64 * (ks >> 1)>(bf[K.value], sbz + a_pbf[ks & 1]);
                });
                lgkm_fence(bf[0]);
#pragma unroll
                for (int k = 0; k < GRP; ++k) { if (k) lgkm_tie(bf[k]); acc = mfma32(w[(G.value + 1) * GRP + k], as_bf(bf[k]), acc); }
            });
        };
        f32x16 aA, aG;
#ifdef VLPET_F2_ABL
        if (VLPET_F2_ABL & 1) { aA = zero16(); aG = zero16(); } else
#endif
        {
        if constexpr (GATE) project_up(std::integral_constant<int, 1>{}, std::integral_constant<int, 32 * NW * 4>{}, wG, aG);
        project_up(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, wA, aA);
        }
        u32x4 xv[2];
        lds_read16<0>(xv[0], sbx + a_xcl[0]);
        lds_read16<0>(xv[1], sbx + a_xcl[1]);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xv[0]), "+v"(xv[1]) :: "memory");
        F2_STAMP(1)
        // outputs: staged in place over the wave's own x2 columns, read back as 4 lanes per row and stored as 64 contiguous bytes
        // of 16 rows per instruction (the accumulator layout would store 16-byte pieces of 32 rows: twice the memory transactions)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = 8 * k + j;
                const float lin = s2g * bf_at(xv[k], j) + sdg * aA[e];
                if constexpr (GATE) {
                    const float gt = sigm(aG[e]);
                    o[j] = ADD ? lin + gs * gt : lin * gt;
#ifdef VLPET_F2_ABL
                    if (VLPET_F2_ABL & 2) o[j] = aG[e] + aA[e];
#endif
                } else o[j] = lin;
            }
            lds_write16<0>(sbx + a_xcl[k], pack8(o));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {
            u32x4 ov[2];
            lds_read16<0>(ov[0], sbx + a_ost[0]);
            lds_read16<0>(ov[1], sbx + a_ost[1]);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ov[0]), "+v"(ov[1]) :: "memory");
            const int64_t row0 = r_begin + 32 * (int64_t)s + (lane >> 2);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#ifdef VLPET_F2_ABL
                if (!(VLPET_F2_ABL & 4))
#endif
                if (row0 + 16 * j < r_end) *reinterpret_cast<u32x4*>(outp + (row0 + 16 * j) * (F2_D * 2)) = ov[j];
        }
        F2_STAMP(2)
    }
    F2_STAMP_PRINT("up", "top(wait+barrier) %llu  issue+proj %llu  ew+store %llu")
}

// ------------------------------------------------------------------------------------------------ K3's keep flags
// One thread per mask dword (32 elements): byte j of dword w of a row holds group G = (p & ~7) + 2 (p & 3) + ((p >> 2) & 1),
// p = 4 w + j (the inverse of rng.h drop_pos), flags from the generator of rng.h (a function of seed and element index only:
// the same mask the in-kernel generator of pet_fwd.hip produces).  keep_out (optional): the 0/1 byte export for parity tests.
__global__ __launch_bounds__(256) void drop_bits_kernel(uint8_t* bits, uint8_t* keep_out, int64_t M, int d, uint64_t seed0, const uint64_t* seed_ctr, uint32_t thr) {
    const uint64_t seed = vlpet_eff_seed(seed0, seed_ctr);
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int wpr = d >> 5;
    if (t >= M * wpr) return;
    const int64_t row = t / wpr;
    const int w = (int)(t - row * wpr);
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = 4 * w + j, G = (p & ~7) + 2 * (p & 3) + ((p >> 2) & 1);
        const int64_t grp = row * (int64_t)(d >> 3) + G;
        const uint32_t kb = keep8(grp, seed, thr);
        if (keep_out != nullptr) drop_export8(keep_out, grp * 8, kb);
        out |= kb << (8 * j);
    }
    reinterpret_cast<uint32_t*>(bits)[t] = out;
}

template <int RT, int SSN, int NCH, bool ACT_ID, bool DROP>
hipError_t launch_down(const PetFwdArgs& a, hipStream_t stream) {
    using GEO = DownGeo<RT, SSN, DROP>;
    auto kern = k1_down_kernel<RT, SSN, NCH, ACT_ID, DROP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GEO::LDS);
    if (e != hipSuccess) return e;
    // row blocks: one round of the chip (two workgroups per CU up to r = 96, one at r = 192), any multiple of 32 rows
    const int slots_per_chain = (RT == 6 ? 256 : 512) / NCH;
    int64_t tiles = (a.M + 31) / 32;
    int64_t tpb = (tiles + slots_per_chain - 1) / slots_per_chain;
    if (tpb < 1) tpb = 1;
    const int rows_per_block = (int)(32 * tpb);
    const int nblocks = (int)((a.M + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(kern, dim3(NCH * nblocks), dim3(GEO::THREADS), GEO::LDS, stream, a, rows_per_block, ACT_ID ? 0 : 1);
    return hipGetLastError();
}

template <int RT, int WPE, int NXD, bool GATE>
hipError_t launch_up(const PetFwdArgs& a, hipStream_t stream) {
    using GEO = UpGeo<RT, NXD>;
    constexpr int NCG = F2_D / (32 * GEO::NW);
    const bool add = GATE && (a.flags & PET_GATE_ADD);
    auto kern = add ? k1_up_kernel<RT, GATE, WPE, NXD, GATE> : k1_up_kernel<RT, false, WPE, NXD, GATE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GEO::LDS);
    if (e != hipSuccess) return e;
    constexpr int S = 16 * WPE;                           // workgroup slots per XCD: 32 CUs x (WPE / 2) 512-thread workgroups
    const int max_chunks = f2_groups_max(NCG, S);
    const int64_t steps = (a.M + 31) / 32;
    int64_t spc = (steps + max_chunks - 1) / max_chunks;
    if (spc < 1) spc = 1;
    const int64_t rows_per_chunk = 32 * spc;
    const int row_chunks = (int)((a.M + rows_per_chunk - 1) / rows_per_chunk);
    const unsigned grid = f2_grid(NCG, S, row_chunks);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), GEO::LDS, stream, a, row_chunks, rows_per_chunk);
    return hipGetLastError();
}

// pass A of one shape family: the gated K1 (two chains), K2 (one chain, gelu_new), K3 (one chain, identity, with or without the mask)
template <int RT>
hipError_t launch_down_rt(const PetFwdArgs& a, hipStream_t stream) {
    if (a.flags & PET_GATE) return launch_down<RT, 3, 2, false, false>(a, stream);
    if (!(a.flags & PET_ACT_IDENTITY)) return launch_down<RT, 3, 1, false, false>(a, stream);
    return a.drop.bits != nullptr ? launch_down<RT, 3, 1, true, true>(a, stream) : launch_down<RT, 3, 1, true, false>(a, stream);
}
template <int RT, int WPE, int NXD>
hipError_t launch_up_rt(const PetFwdArgs& a, hipStream_t stream) {
    return (a.flags & PET_GATE) ? launch_up<RT, WPE, NXD, true>(a, stream) : launch_up<RT, WPE, NXD, false>(a, stream);
}

}  // namespace

// Shapes the two-pass forward takes: bf16, d = 768, training form (the z the cut goes through is an output there), square
// (not the low-rank visual projector).  Dropout (K3): the library's generator or none -- an explicit byte mask stays on pet_fwd.hip.
bool k1_fwd2p_applies(const PetFwdArgs& a, int io_fp32) {
    if (io_fp32 || a.d != F2_D || a.save == nullptr || a.d_in != 0 || a.M < 1 || a.drop.keep != nullptr) return false;
    if (a.flags & PET_GATE) return !(a.flags & PET_ACT_IDENTITY) && !drop_active(a.drop);
    if (a.flags & PET_ACT_IDENTITY) return !drop_active(a.drop) || (a.drop.thr != 0 && a.drop.bits_out != nullptr);
    return !drop_active(a.drop);
}
// Which form is faster (MI355X, bf16, profiles/r04_k1fwd_two_pass_ab.txt): at r = 192 the two-pass form at every size (51 vs 124 us at
// 18,250 rows, 15 vs 62 us at 2,100); up to r = 96 everywhere except where the one-kernel forward runs ONE well-filled round of its
// 128-row workgroups (17,000 < M <= 32,768: 45.7 vs 50.6 us at 28,000 rows) -- below that it is a latency chain of few workgroups
// (32 vs 14 us at 3,500 rows), above it two rounds (81 vs 57 us at 33,200).
bool k1_fwd2p_preferred(const PetFwdArgs& a) {
    if (a.flags & PET_GATE) {
        if (a.RT == 6) return true;
        return !(a.M > 17000 && a.M <= 32768);
    }
    // K2 / K3 (profiles/r04_k2k3_fwd_two_pass_ab.txt): below one well-filled round of the one-kernel forward's 128-row workgroups the
    // two passes win by 1.3-2.7x (K2 at 3,500 rows 26 -> 11 us, K3 r = 64 with dropout at 2,500 rows 36 -> 14 us); just above
    // 32,768 rows (the one-kernel form's second round) by 15 %; at six tiles above 32,768 rows (66 -> 56 us).  With dropout at
    // large M the one-kernel forward with its in-kernel generator stays ahead (42.7 vs 46.6 us at 28,000 rows, r = 64).
    if (a.M <= 17000) return true;
    if (a.RT == 6 && a.M > 32768) return true;
    return !drop_active(a.drop) && a.M > 32768 && a.M <= 42000;
}

hipError_t launch_k1_fwd2p(const PetFwdArgs& a0, int passes, hipStream_t stream) {
    PetFwdArgs a = a0;
    hipError_t e = hipSuccess;
    if (passes & 1) {
        if (!(a.flags & PET_GATE) && a.drop.thr != 0 && a.drop.bits == nullptr) {
            // K3: the packed mask first (where the training form leaves it for the backward), then pass A reads it back
            const int64_t words = a.M * (int64_t)(a.d >> 5);
            hipLaunchKernelGGL(drop_bits_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream,
                               a.drop.bits_out, a.drop.keep_out, a.M, a.d, a.drop.seed, a.drop.seed_ctr, a.drop.thr);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            a.drop.bits = a.drop.bits_out;
        }
        switch (a.RT) {
            case 1: e = launch_down_rt<1>(a, stream); break;
            case 3: e = launch_down_rt<3>(a, stream); break;
            case 6: e = launch_down_rt<6>(a, stream); break;
            default: return hipErrorInvalidValue;
        }
        if (e != hipSuccess) return e;
    }
    if (passes & 2) {
        switch (a.RT) {
            case 1: return launch_up_rt<1, 4, 3>(a, stream);
            case 3: return launch_up_rt<3, 4, 3>(a, stream);
            case 6: return launch_up_rt<6, 2, 4>(a, stream);
            default: return hipErrorInvalidValue;
        }
    }
    return e;
}
