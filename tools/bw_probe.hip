// Memory-system calibration for the PET kernels (round 4): wall-clock bandwidth of the access shapes they use, warm (a buffer that
// fits the 256 MB Infinity Cache, re-read by repeated launches) and cold (2 GiB: every launch misses).
//   read    : global_load_dwordx4, every lane 16 B, grid-stride, W waves per CU
//   copy    : the same + global_store_dwordx4
//   write   : stores only
//   dma     : global_load_lds into a per-wave LDS ring (DEPTH x 1 KiB pieces in flight per wave), default policy / nt
//   rows    : the pass-A shape: 8 rows x 128 B per instruction at a 1536-byte row stride, 6 column stages per 32-row tile
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_bw_probe tools/bw_probe.hip        Run: tools/_bw_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void gmem_cv;
typedef __attribute__((address_space(3))) void lmem_v;

__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ src, size_t n16, uint32_t* sink) {
    u32x4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc += a ^ b ^ c ^ d;
    }
    for (; i < n16; i += stride) acc += src[i];
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[0] = acc[2] + acc[3];
}
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_write(u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const u32x4 v = {1, 2, 3, (uint32_t)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = v;
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// every wave streams its share of the buffer as 1 KiB pieces into a private LDS ring, DEPTH pieces in flight
template <int DEPTH, int AUX>
__global__ __launch_bounds__(256) void k_dma(const uint8_t* __restrict__ src, size_t pieces, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t* ring = smem + (size_t)wave * DEPTH * 1024;
    const size_t nw = (size_t)gridDim.x * (blockDim.x >> 6), w = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
    int slot = 0;
    for (size_t p = w; p < pieces; p += nw) {
        __builtin_amdgcn_global_load_lds((gmem_cv*)(src + p * 1024 + lane * 16), (lmem_v*)(ring + slot * 1024), 16, 0, AUX);
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
        wait_vm<DEPTH - 1>();
    }
    wait_vm<0>();
    if (reinterpret_cast<uint32_t*>(ring)[lane] == 0x12345678u) sink[0] = 1;
}
// pass-A shape: workgroup = one loader wave + idle waves; per 32-row tile 2 sub-steps of 6 stages x 4 instructions (8 rows x 128 B,
// row stride 1536 B); ring of NS sub-step slots, NS - 1 in flight
template <int NS, int AUX>
__global__ __launch_bounds__(64) void k_rows(const uint8_t* __restrict__ src, int64_t M, int rows_per_block, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
    const int nss = 2 * (int)((r1 - r0 + 31) / 32);
    auto issue = [&](int i) {
        const uint8_t* s = src + (r0 + 32 * (int64_t)(i >> 1)) * 1536 + (i & 1) * 768;
        uint8_t* d = smem + (size_t)(i % NS) * 24576;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int st = 0; st < 6; ++st)
                __builtin_amdgcn_global_load_lds((gmem_cv*)(s + (size_t)(8 * q + (lane >> 3)) * 1536 + st * 128 + (lane & 7) * 16),
                                                 (lmem_v*)(d + st * 4096 + q * 1024), 16, 0, AUX);
    };
    for (int i = 0; i < NS - 1 && i < nss; ++i) issue(i);
    for (int i = 0; i < nss; ++i) {
        int ahead = nss - 1 - i; if (ahead > NS - 2) ahead = NS - 2;
        if (ahead <= 0) wait_vm<0>(); else if (ahead == 1) wait_vm<24>(); else if (ahead == 2) wait_vm<48>(); else wait_vm<63>();
        if (i + NS - 1 < nss) issue(i + NS - 1);
    }
    if (reinterpret_cast<uint32_t*>(smem)[lane] == 0x12345678u) sink[0] = 1;
}

template <typename F> static double time_us(F&& launch, int reps, bool flush, void* fl, size_t flbytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        if (flush) hipMemsetAsync(fl, r, flbytes, 0);
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const size_t big = (size_t)2 << 30;
    uint8_t *a, *b; uint32_t* sink; void* fl;
    hipMalloc(&a, big); hipMalloc(&b, big); hipMalloc(&sink, 64); hipMalloc(&fl, (size_t)1 << 30);
    hipMemset(a, 1, big); hipMemset(b, 2, big);
    const size_t sizes[] = {(size_t)43 << 20, (size_t)86 << 20, (size_t)1 << 30};
    for (size_t bytes : sizes) {
        for (int flush = 0; flush < 2; ++flush) {
            if (flush && bytes >= ((size_t)1 << 30)) continue;
            const size_t n16 = bytes / 16;
            const char* tag = flush ? "cold (1 GiB memset before each launch)" : (bytes >= ((size_t)1 << 30) ? "cold (1 GiB)" : "warm (repeated launches)");
            printf("---- %zu MB, %s\n", bytes >> 20, tag);
            for (int wpc : {4, 8, 16, 32}) {
                const int blocks = 256 * wpc / 4;
                double t = time_us([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, (const u32x4*)a, n16, sink); }, 9, flush, fl, (size_t)1 << 30);
                printf("read   %2d waves/CU: %7.1f us  %6.2f TB/s\n", wpc, t, bytes / t / 1e6);
                t = time_us([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n16); }, 9, flush, fl, (size_t)1 << 30);
                printf("copy   %2d waves/CU: %7.1f us  %6.2f TB/s (read + write)\n", wpc, t, 2.0 * bytes / t / 1e6);
                t = time_us([&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, (u32x4*)b, n16); }, 9, flush, fl, (size_t)1 << 30);
                printf("write  %2d waves/CU: %7.1f us  %6.2f TB/s\n", wpc, t, bytes / t / 1e6);
            }
            const size_t pieces = bytes / 1024;
#define DMA(D, AUX, WG) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_dma<D, AUX>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * D * 1024); \
            double t = time_us([&] { hipLaunchKernelGGL((k_dma<D, AUX>), dim3(256 * WG), dim3(256), 4 * D * 1024, 0, a, pieces, sink); }, 9, flush, fl, (size_t)1 << 30); \
            printf("dma    depth %2d x 4 waves x %d WG/CU (%3d KiB in flight per CU) aux %d: %7.1f us  %6.2f TB/s\n", D, WG, 4 * D * WG, AUX, t, bytes / t / 1e6); }
            DMA(8, 0, 1) DMA(8, 2, 1) DMA(16, 0, 1) DMA(16, 2, 1) DMA(32, 0, 1) DMA(32, 2, 1) DMA(8, 0, 2) DMA(16, 0, 2) DMA(16, 2, 2) DMA(8, 0, 4) DMA(8, 2, 4)
            const int64_t M = (int64_t)(bytes / 1536);
#define ROWS(NS, AUX, RPB) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_rows<NS, AUX>), hipFuncAttributeMaxDynamicSharedMemorySize, NS * 24576); \
            const int blocks = (int)((M + RPB - 1) / RPB); \
            double t = time_us([&] { hipLaunchKernelGGL((k_rows<NS, AUX>), dim3(blocks), dim3(64), NS * 24576, 0, a, M, RPB, sink); }, 9, flush, fl, (size_t)1 << 30); \
            printf("rows   ring %d x 24 KiB, %3d rows/WG (%d WGs) aux %d: %7.1f us  %6.2f TB/s\n", NS, RPB, blocks, AUX, t, bytes / t / 1e6); }
            if (bytes < ((size_t)1 << 30)) { ROWS(3, 2, 128) ROWS(3, 0, 128) ROWS(3, 2, 64) ROWS(6, 2, 128) ROWS(6, 2, 256) ROWS(2, 2, 64) }
        }
    }
    return 0;
}
