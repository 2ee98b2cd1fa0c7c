// Issue-side cost of the two ways to stage 1 KiB per wave instruction into LDS, measured with s_memtime around the issue
// of N instructions (data L2-resident, waits outside the timed region):
//   A: global_load_lds (LDS-DMA, 16 B per lane)           B: global_load_dwordx4 -> VGPR, then ds_write_b128
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_stage_issue_probe tools/stage_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void gmem_cv;
typedef __attribute__((address_space(3))) void lmem_v;

template <int MODE, int N>
__global__ __launch_bounds__(256) void probe(const uint8_t* buf, unsigned long long* cyc, uint32_t* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* mine = smem + wave * N * 1024;
    const uint8_t* src = buf + ((size_t)blockIdx.x * 4 + wave) * N * 1024 + lane * 16;
    unsigned long long total = 0;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_barrier();
        const unsigned long long t0 = __builtin_readcyclecounter();
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i)
                __builtin_amdgcn_global_load_lds((gmem_cv*)(src + i * 1024), (lmem_v*)(mine + i * 1024), 16, 0, 0);
        } else {
            u32x4 v[N];
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = *reinterpret_cast<const u32x4*>(src + i * 1024);
            const unsigned long long t1 = __builtin_readcyclecounter();
            total += t1 - t0;                         // issue of the loads only
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t2 = __builtin_readcyclecounter();
#pragma unroll
            for (int i = 0; i < N; ++i) *reinterpret_cast<u32x4*>(mine + i * 1024 + lane * 16) = v[i];
            const unsigned long long t3 = __builtin_readcyclecounter();
            total += t3 - t2;                         // + issue of the ds_writes
        }
        if constexpr (MODE == 0) { const unsigned long long t1 = __builtin_readcyclecounter(); total += t1 - t0; }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        acc += *reinterpret_cast<const u32x4*>(mine + lane * 16);
    }
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = total;
    sink[blockIdx.x * 256 + threadIdx.x] = acc[0];
}

template <int MODE, int N>
static void run(const char* name, int blocks, const uint8_t* buf, unsigned long long* cyc, uint32_t* sink) {
    const int iters = 200;
    auto k = probe<MODE, N>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * N * 1024);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 4 * N * 1024, 0, buf, cyc, sink, iters);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 4 * N * 1024, 0, buf, cyc, sink, iters);
    unsigned long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-46s blocks=%3d N=%2d : %6.1f cycles per instruction (issue side, wave 0)\n", name, blocks, N, (double)h[0] / iters / N);
}

int main() {
    uint8_t* buf; unsigned long long* cyc; uint32_t* sink;
    hipMalloc(&buf, (size_t)256 * 4 * 16 * 1024); hipMemset(buf, 1, (size_t)256 * 4 * 16 * 1024);
    hipMalloc(&cyc, 256 * 4 * 8); hipMalloc(&sink, 256 * 256 * 4);
    for (int blocks : {16, 219}) {
        run<0, 7>("global_load_lds x7 per wave (4 waves)", blocks, buf, cyc, sink);
        run<0, 14>("global_load_lds x14 per wave (4 waves)", blocks, buf, cyc, sink);
        run<1, 7>("global_load_dwordx4 + ds_write_b128 x7", blocks, buf, cyc, sink);
        run<1, 14>("global_load_dwordx4 + ds_write_b128 x14", blocks, buf, cyc, sink);
    }
    return 0;
}
