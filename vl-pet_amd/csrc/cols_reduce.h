// In-launch reduce-scatter of the row-chunk partials of the column-parallel backward passes (round 6; pet_cols.hip, pet_cols_ng.hip).
//
// Before: every workgroup (column block cb, row chunk rc) stored its [4 x 32*RT x 128] fp32 accumulators as a slab of the workspace and
// a SECOND launch (wgrad_finalize_kernel) summed the RC slabs of every column block: a dependent kernel boundary behind ~47 MB of
// freshly written partials + a 12-16 us launch that reads them back (11 % of the K1 backward at 28,000 rows, 20 % at the per-rank sizes).
//
// Now the RC workgroups of a column block do that sum themselves, each a 1/RC slice of the block's slab (the workgroups are already one
// per CU -- the LDS ring sees to that):
//   1. slab out: the accumulators leave as whole 16-byte units in REGISTER order (unit = (wave, array, c-tile, register quad, lane): 1 KiB
//      per wave-instruction), write-through (sc1: cdna_hip_programming.md Guideline 16, form R1 / "publish-large"), then vmcnt(0) +
//      barrier + ONE relaxed agent-scope fetch_add on the column block's arrival counter;
//   2. wait: one lane polls that counter (relaxed, s_sleep) until all RC workgroups of the block have arrived -- BOUNDED;
//   3. sum: thread t of slice s adds unit u of the RC slabs in chunk order 0, 1, .. (sc1 loads, every chunk's load in flight before the
//      first add) -- the order and the arithmetic of wgrad_finalize_kernel, so the results are bit-identical to the two-launch form --
//      applies the job's scale, drops the rank padding and writes the parameter's layout (the flat gradient buffer).
// Nothing here presumes that the workgroups of a block run at the same time (MI355X_MICROARCH.md: dispatch order and residency are not
// promised; the GPU may be shared with a collective or another process).  A workgroup whose wait runs out marks its slice "given up" and
// leaves; the LAST arriver of a block never waits (its own arrival completes the count), and after its own slice it sums every slice
// whose owner gave up (or has not decided within a second bound: a duplicate sum writes the same bits).  So every slice is summed at
// least once whatever the schedule, the result does not depend on who summed it, and no second launch exists even as a fallback.
// State: per column block COLS_RED_STRIDE 32-bit words of the workspace -- [0] arrivals, [2 + s] slice s: 0 undecided / 1 owner sums /
// 2 owner gave up -- zeroed by PASS 1 of the same op (its first workgroup; run_bwd in api.hip falls back to a memset node where pass 1
// is a kernel that does not), so no memset node sits between the passes and nothing depends on a previous call.
#pragma once
#include "cols_common.h"

// (ColsRedJob / ColsRedArgs / COLS_RED_STRIDE: kernels.h)

#define COLS_RED_BATCH 16                    // chunk loads of a unit in flight at once
typedef __attribute__((address_space(1))) unsigned cr_gu32;

__device__ __forceinline__ void cols_red_store_f32(float* p, float v) {       // 4-byte partial, write-through
    __hip_atomic_store((cr_gu32*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float cols_red_load_f32(const float* p) {
    return __uint_as_float(__hip_atomic_load((cr_gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Geometry of a slab: NJB accumulator arrays per wave, RT c-tiles each
template <int RT, int NJB> struct ColsRedGeo {
    static constexpr int UNITS = 8 * NJB * RT * 4 * 64;      // 16-byte units of a slab
    static constexpr int SLAB_B = UNITS * 16;
    static constexpr int SLAB_F = UNITS * 4;
};

// step 1: array jb, c-tile ct of this wave -> its slab (write-through 16-byte stores, one KiB per instruction)
template <int RT, int NJB>
__device__ __forceinline__ void cols_red_put(__amdgpu_buffer_rsrc_t slab_rsrc, int wave, int lane, int jb, int ct, const f32x16& acc) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32x4 v = {__float_as_uint(acc[4 * q]), __float_as_uint(acc[4 * q + 1]), __float_as_uint(acc[4 * q + 2]), __float_as_uint(acc[4 * q + 3])};
        const int unit = (((wave * NJB + jb) * RT + ct) * 4 + q) * 64 + lane;
        __builtin_amdgcn_raw_buffer_store_b128(v, slab_rsrc, unit * 16, 0, 16);      // aux 16 = sc1
    }
}

// steps 2 + 3.  JOBMAP(role, jb) -> job index; XSMAP(x) -> job whose colsum_x the x-th column-sum vector is; PTMAP(k) -> (job, first
// element) of bottleneck tile k's column sums.  `lds` = one free 16-byte LDS location.  Every thread of the workgroup calls this after
// it has issued its slab / bias stores.
template <int RT, int NJB, int NXS, int NPT, typename JobMap, typename XsMap, typename PtMap>
__device__ __forceinline__ void cols_red_finish(const ColsRedArgs& r, int NCB, int RC, int cb, int rc, int d, volatile unsigned* lds,
                                                JobMap jobmap, XsMap xsmap, PtMap ptmap) {
    using G = ColsRedGeo<RT, NJB>;
    const int tid = threadIdx.x, lane = tid & 63;
    cr_gu32* ctrl = (cr_gu32*)(r.ctrl + (size_t)cb * COLS_RED_STRIDE);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // EVERY storing wave drains its write-through stores ...
    __syncthreads();
    if (tid == 0) {                                                   // ... then ONE lane arrives
        const unsigned ticket = __hip_atomic_fetch_add(ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned mode = 2u;                                           // 2: last arriver (nothing to wait for), 1: all arrived in time, 0: gave up
        if (ticket + 1u != (unsigned)RC) {
            mode = 0u;
            for (unsigned spins = 0; spins < r.spin_limit; ++spins) {
                if (__hip_atomic_load(ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)RC) { mode = 1u; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __hip_atomic_store(ctrl + 2 + rc, mode == 0u ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds[0] = mode;
    }
    __syncthreads();
    const unsigned mode = lds[0];
    if (mode == 0u) return;

    const __amdgpu_buffer_rsrc_t all = __builtin_amdgcn_make_buffer_rsrc(r.slab, 0, (int)((size_t)RC * NCB * G::SLAB_B), 0x00020000);
    const unsigned cstride = (unsigned)(NCB * G::SLAB_B);            // bytes between the slabs of consecutive row chunks
    // ONE copy of the summing code: the own slice first (pass 0), then -- last arriver only -- 64 slices per pass whose states one wave
    // has polled (every slice whose owner gave up, or has not said so within a second bound: a duplicate sum writes the same bits)
    unsigned long long todo = 1ull;
    int sbase_ = rc;
    for (int pass = 0;; ++pass) {
        while (todo) {
            const int s = sbase_ + __builtin_ctzll(todo);
            todo &= todo - 1;
            const int u0 = (int)(((int64_t)s * G::UNITS) / RC), u1 = (int)(((int64_t)(s + 1) * G::UNITS) / RC);
#pragma unroll 1
            for (int u = u0 + tid; u < u1; u += 512) {
                const unsigned voff = (unsigned)(cb * G::SLAB_B + u * 16);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int c0 = 0; c0 < RC; c0 += COLS_RED_BATCH) {     // a batch of chunk loads in flight before its first add; chunk order kept
                    u32x4 v[COLS_RED_BATCH];
#pragma unroll
                    for (int q = 0; q < COLS_RED_BATCH; ++q) {
                        const int c2 = c0 + q < RC ? c0 + q : RC - 1;
                        v[q] = __builtin_amdgcn_raw_buffer_load_b128(all, voff, (unsigned)c2 * cstride, 16);      // sc1: past this CU's L1
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < COLS_RED_BATCH; ++q)
                        if (c0 + q < RC) acc += __builtin_bit_cast(f32x4, v[q]);
                }
                const int ln = u & 63, q = (u >> 6) & 3;
                int rest = u >> 8;
                const int ct = rest % RT; rest /= RT;
                const int jb = rest % NJB, w = rest / NJB;
                const int m = ln & 31, h = ln >> 5;
                const int crow = 32 * ct + 8 * q + 4 * h, col = 128 * cb + 32 * (w & 3) + m;
                const ColsRedJob& J = r.job[jobmap(w >> 2, jb)];
                acc = acc * J.scale;
                if (J.transposed) {                                   // out[col][crow ..]: the lane's four rows are contiguous
                    float* o = J.out + (int64_t)col * J.ldo + crow;
                    if (crow + 3 < J.out_rows && (reinterpret_cast<uintptr_t>(o) & 15) == 0) *reinterpret_cast<f32x4*>(o) = acc;
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (crow + j < J.out_rows) o[j] = acc[j];
                    }
                } else {                                              // out[crow + j][col]: the 32 lanes of a row write 128 contiguous bytes
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (crow + j < J.out_rows) J.out[(int64_t)(crow + j) * J.ldo + col] = acc[j];
                }
            }
            if (s == 0) {                                             // the column block's bias gradients ride with slice 0
                for (int e = tid; e < NXS * 128; e += 512) {
                    const int x = e >> 7, col = 128 * cb + (e & 127);
                    const ColsRedJob& J = r.job[xsmap(x)];
                    if (J.colsum_x == nullptr) continue;
                    const float* p = r.bias_x + (int64_t)x * RC * d + col;
                    float sum = 0.f;
                    int c = 0;
                    for (; c + 8 <= RC; c += 8) {
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = cols_red_load_f32(p + (int64_t)(c + q) * d);
#pragma unroll
                        for (int q = 0; q < 8; ++q) sum += v[q];
                    }
                    for (; c < RC; ++c) sum += cols_red_load_f32(p + (int64_t)c * d);
                    J.colsum_x[col] = sum * J.scale;
                }
                for (int e = tid; e < NPT * 32; e += 512) {           // tiles whose sums THIS column block's workgroups produced
                    const int k = e >> 5;
                    if ((k % (4 * NCB)) % NCB != cb) continue;
                    int job, first;
                    ptmap(k, job, first);
                    const ColsRedJob& J = r.job[job];
                    const int c = first + (e & 31);
                    if (J.colsum_p == nullptr || c >= J.out_rows) continue;
                    const float* p = r.bias_p + e;
                    float sum = 0.f;
                    int cc = 0;
                    for (; cc + 8 <= RC; cc += 8) {
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = cols_red_load_f32(p + (int64_t)(cc + q) * (NPT * 32));
#pragma unroll
                        for (int q = 0; q < 8; ++q) sum += v[q];
                    }
                    for (; cc < RC; ++cc) sum += cols_red_load_f32(p + (int64_t)cc * (NPT * 32));
                    J.colsum_p[c] = sum;
                }
            }
        }
        if (mode != 2u || 64 * pass >= RC) return;
        sbase_ = 64 * pass;
        __syncthreads();
        if (tid < 64) {
            const int s = sbase_ + lane;
            bool redo = false;
            if (s < RC && s != rc) {
                unsigned st = 0u;
                for (unsigned spins = 0; spins < 2u * r.spin_limit + 64u; ++spins) {
                    st = __hip_atomic_load(ctrl + 2 + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (st != 0u) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                redo = st != 1u;
            }
            const unsigned long long mask = __ballot(redo);
            if (lane == 0) { lds[1] = (unsigned)mask; lds[2] = (unsigned)(mask >> 32); }
        }
        __syncthreads();
        todo = ((unsigned long long)lds[2] << 32) | lds[1];
    }
}
