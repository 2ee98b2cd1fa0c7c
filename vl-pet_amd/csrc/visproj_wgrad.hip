// K4 weight gradient as a tiled split-K GEMM (bf16):  dW[c, n] = sum_m dpre[m, c] * feats[m, n],  db[c] = sum_m dpre[m, c]
// (autograd of the Linear of VisualEmbedding.feat_embedding, src/modeling_bart.py:91-110 / :157; T5: src/modeling_t5.py:56-66).
//
// Until round 3 this gradient ran as d_out / 96 jobs of the generic weight-gradient stream (wgrad.hip): each job re-read ALL
// of feats (8 x 76.6 MB at the bench's 18,700 visual rows) for a [96 x 2048] slice of dW -- 150 us, 0.16 of the MFMA peak,
// bound by those re-reads.  Here a workgroup owns a [384 c x 256 n] tile of dW and a row chunk: feats is read d_out / 384 = 2
// times, dpre F / 256 = 8 times (out of L2: the tiles of a row chunk run on one XCD), 24 MFMAs per wave for 14 transpose reads.
//   * 8 waves = 4 c-quarters (96 c = 3 tiles) x 2 n-halves (128 n = 4 tiles): 12 accumulator tiles = 192 registers per wave,
//     two waves per SIMD (the two n-halves of a c-quarter);
//   * a stage = 32 rows: the dpre tile [32 x 768 B] and the feats tile [32 x 512 B], 40 pieces of 1 KiB by global_load_lds
//     (5 per wave) into a 3-slot ring, two stages ahead, one barrier per stage;
//   * both operands of the m-contraction come out of the row-major tiles by ds_read_b64_tr_b16; the rows of both tiles are a
//     multiple of 256 B long, so the 16-byte slots are swizzled on the source side (slot ^= (row & 3) << 2): the four rows of a
//     transpose read then fall into four different 64-byte bank windows;
//   * row-chunk partials in wgrad.hip's workspace layout (one job, PR = d_out), summed by wgrad_finalize_kernel
//     (deterministic, no atomics); the bias gradient is summed from the A operands (8 rows of a column per lane) in the n-slab-0 workgroups.
#include "cols_common.h"

struct K4WgArgs {
    const void* P; const void* X;     // dpre [M, d_out], feats [M, F]  (bf16)
    float* partial;
    int64_t M;
    int d_out, F;
    int row_chunks; int64_t rows_per_chunk;
    int xcd_map;                      // row_chunks % 8 == 0: the 16 * row_chunks / 8 workgroups of an XCD are whole row chunks
};

namespace {
constexpr int PT_B = 32 * 768, XT_B = 32 * 512, STG_B = PT_B + XT_B, NSTG = 3, NWP = 5;   // pieces per wave and stage
}

__global__ __launch_bounds__(512, 2) void k4_wgrad_kernel(K4WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cq = wave & 3, nh = wave >> 2;
    const int m = lane & 31, h = lane >> 5;
    const int NS = a.F >> 8, NCH = a.d_out / 384, TPR = NS * NCH;        // tiles per row chunk
    int rc, t;
    if (a.xcd_map) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, k = a.row_chunks >> 3;
        rc = xcd * k + j / TPR; t = j % TPR;
    } else {
        rc = blockIdx.x / TPR; t = blockIdx.x % TPR;
    }
    const int ch = t % NCH, ns = t / NCH;
    const int64_t ldp2 = (int64_t)a.d_out * 2, ldx2 = (int64_t)a.F * 2;
    const int64_t r_begin = (int64_t)rc * a.rows_per_chunk;
    int64_t r_end = r_begin + a.rows_per_chunk;
    if (r_end > a.M) r_end = a.M;
    const int nsteps = r_end > r_begin ? (int)((r_end - r_begin + 31) >> 5) : 0;
    const bool want_csp = ns == 0 && nh == 0;

    // ---- the stage pieces of this wave (5 of the 40): wave w brings rows 4w .. 4w+3 of the dpre tile as three pieces of
    // [4 rows x 16 slots] (LDS image of that tile: [8 row groups][3 slot blocks][4 rows][16 slots]) and rows 2w, 2w+1 and 2w+16,
    // 2w+17 of the feats tile (row-major, 32 slots per row).  Per lane ONE 32-bit offset per tensor; the row base is scalar.
    const int prow = 4 * wave + (lane >> 4), xrow = 2 * wave + (lane >> 5);
    const uint32_t poff = (uint32_t)prow * (uint32_t)ldp2 + (uint32_t)((((lane & 15) ^ ((prow & 3) << 2))) * 16);
    const uint32_t xoff = (uint32_t)xrow * (uint32_t)ldx2 + (uint32_t)((((lane & 31) ^ ((xrow & 3) << 2))) * 16);
    const uint8_t* Pb = reinterpret_cast<const uint8_t*>(a.P) + (int64_t)ch * 768;
    const uint8_t* Xb = reinterpret_cast<const uint8_t*>(a.X) + (int64_t)ns * 512;
    auto sbase = [](const uint8_t* p) {     // a wave-uniform pointer as a fresh scalar (keeps the per-lane part a 32-bit loop invariant)
        const uint64_t u = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const uint8_t*>(((uint64_t)hi << 32) | lo);
    };
    auto issue = [&](int s) {
        const int64_t rb = r_begin + 32 * (int64_t)s;
        uint8_t* st = smem + (size_t)(s % NSTG) * STG_B;
        const int last = (int)(r_end - rb) - 1;                          // rows past the end re-read the last row (wave-uniform test first)
        uint32_t po = poff, xo0 = xoff, xo1 = xoff + 16u * (uint32_t)ldx2;
        if (last < 31) {
            po -= (uint32_t)(prow > last ? prow - last : 0) * (uint32_t)ldp2;
            xo0 -= (uint32_t)(xrow > last ? xrow - last : 0) * (uint32_t)ldx2;
            xo1 -= (uint32_t)(xrow + 16 > last ? xrow + 16 - last : 0) * (uint32_t)ldx2;
        }
        const uint8_t* pb = sbase(Pb + rb * ldp2);
        const uint8_t* xb = sbase(Xb + rb * ldx2);
#pragma unroll
        for (int j = 0; j < 3; ++j) glds16(pb + po + 256 * j, st + (3 * wave + j) * 1024);       // (dpre rows are re-read by the 8 n-slabs: default cache policy)
        glds16_row(xb + xo0, st + PT_B + wave * 1024);
        glds16_row(xb + xo1, st + PT_B + (wave + 8) * 1024);
    };

    // ---- per-lane LDS byte addresses of the transpose reads (k-step 0, first four rows; + 4 rows and + 16 rows are offsets):
    // dpre tile: (row, slot) at ((row >> 2) * 3 + (slot >> 4)) * 1024 + ((row & 3) * 16 + (slot & 15)) * 16, slot swizzled by row & 3
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    uint32_t pa[3], xa[4];
    {
        const int g4 = lane >> 4, sl = lane & 15;
        const int row = 8 * (g4 >> 1) + (sl >> 2), r3 = sl >> 2;
        const int low = 2 * (g4 & 1) + ((sl & 3) >> 1), half = 8 * (sl & 1);
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            const int slot = (12 * cq + 4 * ct + low) ^ (r3 << 2);
            pa[ct] = (uint32_t)(((row >> 2) * 3 + (slot >> 4)) * 1024 + ((row & 3) * 16 + (slot & 15)) * 16 + half);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) xa[nt] = (uint32_t)(PT_B + row * 512 + (((16 * nh + 4 * nt + low) ^ (r3 << 2)) * 16) + half);
    }

    f32x16 acc[3][4];
    float csum[3] = {0.f, 0.f, 0.f};                 // column sums of dpre over this lane's row slots (n-slab-0 workgroups)
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[ct][nt] = zero16();

#pragma unroll
    for (int s = 0; s < NSTG - 1; ++s)
        if (s < nsteps) issue(s);
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        int ahead = nsteps - 1 - s;
        if (ahead > NSTG - 2) ahead = NSTG - 2;
        vm_wait(ahead * NWP);
        __builtin_amdgcn_s_barrier();                                     // stage s has landed for every wave; the slot of stage s - 1 is free
        if (s + NSTG - 1 < nsteps) issue(s + NSTG - 1);
        const int valid = (int)(r_end - (r_begin + 32 * (int64_t)s));
        if (valid < 32) {                                                 // last step: the dpre rows past the end must not contribute
            const u32x4 z = {0u, 0u, 0u, 0u};
            uint8_t* pt = smem + (size_t)(s % NSTG) * STG_B;
            for (int q = tid; q < 32 * 48; q += 512) {                     // 16-byte unit q of the image: row 4 (q / 192) + (q / 16) % 4
                const int row = 4 * (q / 192) + ((q >> 4) & 3);
                if (row >= valid) *reinterpret_cast<u32x4*>(pt + (size_t)q * 16) = z;
            }
            __syncthreads();
        }
        const uint32_t sb = lds0 + (uint32_t)((s % NSTG) * STG_B);
        sfor<2>([&](auto KS) {
            constexpr int ks = KS.value;
            TrOp ap[3], bx[4];
            sfor<4>([&](auto NT) { tr_read<ks * 16 * 512, (ks * 16 + 4) * 512>(bx[NT.value], sb + xa[NT.value]); });
            sfor<3>([&](auto CT) { tr_read<ks * 4 * 3072, (ks * 4 + 1) * 3072>(ap[CT.value], sb + pa[CT.value]); });
            tr_fence(bx[0]);
#pragma unroll
            for (int nt = 1; nt < 4; ++nt) tr_tie(bx[nt]);
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                tr_tie(ap[ct]);
                const bf16x8 av = tr_val(ap[ct]);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[ct][nt] = mfma32(av, tr_val(bx[nt]), acc[ct][nt]);
            }
            if (want_csp) {                                               // bias gradient: this lane's 8 rows of its column, plain adds
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) {
                    const u32x4 v = __builtin_bit_cast(u32x4, tr_val(ap[ct]));
#pragma unroll
                    for (int w = 0; w < 4; ++w) csum[ct] += bf_lo(v[w]) + bf_hi(v[w]);
                }
            }
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // ---- this row chunk's partial tile, wgrad.hip's workspace layout for ONE job with PR = d_out rows
    const int xc = a.F, PR = a.d_out;
    float* tile = a.partial + (int64_t)rc * PR * xc;
    const int c0 = 384 * ch + 96 * cq, n0 = 256 * ns + 128 * nh;
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int crow = c0 + 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h;
                tile[(int64_t)crow * xc + n0 + 32 * nt + m] = acc[ct][nt][i];
            }
    if (want_csp) {
        float* psp = a.partial + (int64_t)a.row_chunks * PR * xc + (int64_t)a.row_chunks * xc + (int64_t)rc * PR;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            const float v = csum[ct] + __shfl_xor(csum[ct], 32);          // lanes (c, 0) and (c, 1) hold the two halves of every 16 rows
            if (h == 0) psp[c0 + 32 * ct + m] = v;
        }
    }
}

bool k4_wgrad2_applies(int64_t M, int F, int d_out, int io_fp32) {
    return !io_fp32 && M > 0 && F % 256 == 0 && d_out % 384 == 0 && d_out <= 6144 && F >= 256;
}
void k4_wgrad2_plan(int64_t M, int F, int d_out, int* row_chunks, int64_t* rows_per_chunk) {
    const int tpr = (F / 256) * (d_out / 384);                      // tiles per row chunk
    int64_t rc = 256 / tpr;                                          // one workgroup per CU
    rc = rc / 8 * 8;
    if (rc < 8) rc = 8;
    const int64_t blocks32 = (M + 31) / 32;
    if (rc > blocks32) rc = blocks32;
    const int64_t per = (blocks32 + rc - 1) / rc;
    rc = (blocks32 + per - 1) / per;
    *row_chunks = (int)rc; *rows_per_chunk = per * 32;
}
size_t k4_wgrad2_workspace_bytes(int64_t M, int F, int d_out) {
    int rc; int64_t rpc;
    k4_wgrad2_plan(M, F, d_out, &rc, &rpc);
    return (size_t)rc * ((size_t)d_out * F + F + d_out) * sizeof(float);
}

hipError_t launch_k4_wgrad2(const void* dpre, const void* feats, float* dw, float* db, void* workspace, int64_t M, int F, int d_out,
                            hipStream_t stream) {
    K4WgArgs a{};
    a.P = dpre; a.X = feats; a.partial = reinterpret_cast<float*>(workspace);
    a.M = M; a.d_out = d_out; a.F = F;
    k4_wgrad2_plan(M, F, d_out, &a.row_chunks, &a.rows_per_chunk);
    a.xcd_map = a.row_chunks % 8 == 0;
    const int tpr = (F / 256) * (d_out / 384);
    const size_t lds = (size_t)NSTG * STG_B;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k4_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k4_wgrad_kernel, dim3((unsigned)(tpr * a.row_chunks)), dim3(512), lds, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // the chunk sum: wgrad.hip's finalize on a one-job description with PR = d_out (RT = d_out / 32)
    WgradArgs g{};
    g.M = M; g.RT = d_out / 32; g.row_chunks = a.row_chunks; g.rows_per_chunk = a.rows_per_chunk;
    g.partial = a.partial; g.njobs = 1;
    WgradJob& J = g.job[0];
    J.P = dpre; J.ldp = d_out; J.pcols = d_out; J.X = feats; J.ldx = F; J.xcols = F;
    J.has_drop = 0; J.scale = 1.f; J.out = dw; J.ldo = F; J.transposed = 0; J.out_rows = d_out;
    J.colsum_x = nullptr; J.colsum_p = db;
    return launch_wgrad_finalize(g, stream);
}
