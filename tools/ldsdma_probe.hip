// Loader-wave probe: how fast can NL waves of one workgroup per CU stream a stage of W weight pieces (1 KiB each, the same
// L2-resident buffer for every workgroup) + R row pieces (private, HBM) into LDS, with one s_barrier per stage?
//   mode 0: global_load_lds, wait until everything of THIS stage but the rows has landed (prefetch distance 1 for weights)
//   mode 1: global_load_lds, wait only for the PREVIOUS stage's pieces (distance 2 for everything)
//   mode 2: global_load_dwordx4 -> VGPR -> ds_write_b128, previous stage's registers written this stage (distance 2)
//   mode 3: global_load_lds, no waits at all (issue/throughput only)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_ldsdma_probe tools/ldsdma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void gmem_cv;
typedef __attribute__((address_space(3))) void lmem_v;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int NL, int W, int R>      // W, R: pieces per loader wave per stage
__global__ __launch_bounds__(768) void probe(const uint8_t* wbuf, const uint8_t* rbuf, unsigned long long* cyc,
                                             uint32_t* sink, int stages) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int P = W + R;
    if (wave >= NL) {                       // the other waves only keep the barrier count
        for (int t = 0; t < stages; ++t) __builtin_amdgcn_s_barrier();
        return;
    }
    const uint8_t* rows = rbuf + ((size_t)blockIdx.x * NL + wave) * (size_t)stages * R * 1024 + lane * 16;
    const uint8_t* wts = wbuf + (size_t)wave * W * 1024 + lane * 16;
    u32x4 regs[MODE == 2 ? P : 1];
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < stages; ++t) {
        uint8_t* dst = smem + ((size_t)(t % 3) * NL + wave) * P * 1024;
        const uint8_t* ws = wts + (size_t)(t % 24) * NL * W * 1024;
        const uint8_t* rs = rows + (size_t)t * R * 1024;
        if constexpr (MODE == 2) {
            if (t > 0) {
#pragma unroll
                for (int i = 0; i < P; ++i) *reinterpret_cast<u32x4*>(dst + i * 1024 + lane * 16) = regs[i];
            }
#pragma unroll
            for (int i = 0; i < W; ++i) regs[i] = *reinterpret_cast<const u32x4*>(ws + i * 1024);
#pragma unroll
            for (int i = 0; i < R; ++i) regs[W + i] = *reinterpret_cast<const u32x4*>(rs + i * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int i = 0; i < W; ++i)
                __builtin_amdgcn_global_load_lds((gmem_cv*)(ws + i * 1024), (lmem_v*)(dst + i * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < R; ++i)
                __builtin_amdgcn_global_load_lds((gmem_cv*)(rs + i * 1024), (lmem_v*)(dst + (W + i) * 1024), 16, 0, 0);
            if constexpr (MODE == 0) wait_vm<R>();
            if constexpr (MODE == 1) wait_vm<P>();
        }
        __builtin_amdgcn_s_barrier();
    }
    if constexpr (MODE != 2) wait_vm<0>();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if constexpr (MODE == 2) { for (int i = 0; i < P; ++i) acc += regs[i]; }
    if (lane == 0 && wave == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc[0] == 0x12345678) sink[0] = acc[1];
}

template <int MODE, int NL, int W, int R>
static void run(const char* what, int blocks, const uint8_t* wbuf, const uint8_t* rbuf, unsigned long long* cyc, uint32_t* sink) {
    const int stages = 48;
    const size_t lds = (size_t)3 * NL * (W + R) * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, NL, W, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, NL, W, R>), dim3(blocks), dim3(768), lds, 0, wbuf, rbuf, cyc, sink, stages);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
    const double per_stage = s / blocks / stages;
    const double kb = (double)NL * (W + R);
    printf("%-34s NL=%d W=%d R=%d blocks=%3d : %7.0f cycles/stage  %5.1f B/clk/CU  (%.1f us, %.0f GB/s chip)\n", what, NL, W, R, blocks,
           per_stage, kb * 1024 / per_stage, ms * 1e3, (double)blocks * stages * kb * 1024 / (ms * 1e-3) / 1e9);
}

int main() {
    uint8_t *wbuf, *rbuf; unsigned long long* cyc; uint32_t* sink;
    const size_t wsz = (size_t)24 * 64 * 1024, rsz = (size_t)256 * 8 * 48 * 8 * 1024;
    hipMalloc(&wbuf, wsz); hipMemset(wbuf, 1, wsz); hipMalloc(&rbuf, rsz); hipMemset(rbuf, 1, rsz);
    hipMalloc(&cyc, 1024 * 8); hipMalloc(&sink, 64);
    for (int blocks : {16, 219}) {
        run<0, 4, 6, 8>("glds dist1 (as the kernel)", blocks, wbuf, rbuf, cyc, sink);
        run<1, 4, 6, 8>("glds dist2", blocks, wbuf, rbuf, cyc, sink);
        run<3, 4, 6, 8>("glds no waits", blocks, wbuf, rbuf, cyc, sink);
        run<2, 4, 6, 8>("load+ds_write dist2", blocks, wbuf, rbuf, cyc, sink);
        run<0, 8, 3, 4>("glds dist1", blocks, wbuf, rbuf, cyc, sink);
        run<1, 8, 3, 4>("glds dist2", blocks, wbuf, rbuf, cyc, sink);
        run<2, 8, 3, 4>("load+ds_write dist2", blocks, wbuf, rbuf, cyc, sink);
        run<1, 12, 2, 3>("glds dist2", blocks, wbuf, rbuf, cyc, sink);
        run<0, 4, 6, 0>("glds dist1 weights only", blocks, wbuf, rbuf, cyc, sink);
        run<1, 4, 6, 0>("glds dist2 weights only", blocks, wbuf, rbuf, cyc, sink);
        run<1, 4, 0, 8>("glds dist2 rows only", blocks, wbuf, rbuf, cyc, sink);
    }
    return 0;
}
