// Per-CU global-load throughput probe: every workgroup (4 waves) streams `kb` KiB with coalesced 16-byte loads
// (UNROLL loads in flight per wave) from (mode 0) ONE buffer shared by all workgroups -- the weight pack pattern, L2
// resident -- or (mode 1) its own region -- the activation-row pattern.  Reports GB/s per CU and in total.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_load_probe tools/load_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void stream(const uint8_t* buf, uint32_t* out, int kb, int mode, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t* base = buf + (mode ? (size_t)blockIdx.x * kb * 1024 : 0);
    u32x4 acc = {0, 0, 0, 0};
    const int per_wave = kb / 4;                   // KiB per wave
    for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < per_wave; k += UNROLL) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                v[u] = *reinterpret_cast<const u32x4*>(base + ((size_t)(wave * per_wave + k + u)) * 1024 + lane * 16);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int UNROLL>
static void run(int blocks, int kb, int mode, const uint8_t* buf, uint32_t* out) {
    const int iters = 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(stream<UNROLL>, dim3(blocks), dim3(256), 0, 0, buf, out, kb, mode, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(stream<UNROLL>, dim3(blocks), dim3(256), 0, 0, buf, out, kb, mode, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * kb * 1024 * iters;
    printf("blocks=%3d kb=%4d mode=%d unroll=%2d : %7.1f us  %7.1f GB/s per WG  %8.1f GB/s total\n", blocks, kb, mode, UNROLL,
           ms * 1e3, bytes / blocks / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e9);
}

int main() {
    uint8_t* buf; uint32_t* out;
    const size_t sz = (size_t)256 * 1280 * 1024;
    hipMalloc(&buf, sz); hipMemset(buf, 1, sz); hipMalloc(&out, 256 * 256 * 4);
    for (int blocks : {16, 219}) {
        for (int mode : {0, 1}) {
            run<4>(blocks, 1280, mode, buf, out);
            run<8>(blocks, 1280, mode, buf, out);
            run<16>(blocks, 1280, mode, buf, out);
        }
    }
    return 0;
}
