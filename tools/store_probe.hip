// Store-shape calibration (round 4, after the attention finding): what the OUTPUT pattern of the column-parallel K1 backward costs.
// A workgroup owns a 128-column block (256 B per row) of a row chunk of an [M, 768] bf16 tensor and writes two such tensors
// (dx1, dx2), 32 rows per step, with a read stream of the same shape beside it (dy, x1, x2 -> three tensors read by LDS-DMA-less
// plain loads, optional):
//   cols : the kernels' shape -- four waves, wave w owns columns 32 w .. 32 w + 31, lane (m, h) stores 2 x 16 B of row m
//          (32 rows x 64 B per wave and tensor: every 128-byte line is completed by four instructions of two waves)
//   full : the same bytes as whole lines -- the four waves store 8 rows x 256 B each per step (lane = 16 B of a row, 16 lanes
//          per row): every instruction writes whole 128-byte lines
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_store_probe tools/store_probe.hip      Run: tools/_store_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool FULL, bool READ>
__global__ __launch_bounds__(256) void k_cols_store(uint8_t* __restrict__ o1, uint8_t* __restrict__ o2, const uint8_t* __restrict__ i1,
                                                    const uint8_t* __restrict__ i2, const uint8_t* __restrict__ i3, int64_t M,
                                                    int64_t rows_per_chunk, uint32_t* sink) {
    const int ncb = 6, cb = blockIdx.x % ncb, rc = blockIdx.x / ncb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = rc * rows_per_chunk;
    int64_t r1 = r0 + rows_per_chunk; if (r1 > M) r1 = M;
    u32x4 acc = {1u, 2u, 3u, (uint32_t)threadIdx.x};
    for (int64_t rb = r0; rb < r1; rb += 32) {
        if (READ) {                                      // the read stream in whole lines (what the LDS-DMA ring does)
            const int64_t row = rb + 8 * wave + (lane >> 4);
            const int64_t rr = row < M ? row : M - 1;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const size_t off = (size_t)(rr + 4 * q < M ? rr + 4 * q : M - 1) * 1536 + cb * 256 + (lane & 15) * 16;
                acc += *reinterpret_cast<const u32x4*>(i1 + off);
                acc += *reinterpret_cast<const u32x4*>(i2 + off);
                acc += *reinterpret_cast<const u32x4*>(i3 + off);
            }
        }
        if (FULL) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int64_t row = rb + 8 * wave + 4 * q + (lane >> 4);
                if (row < r1) {
                    const size_t off = (size_t)row * 1536 + cb * 256 + (lane & 15) * 16;
                    *reinterpret_cast<u32x4*>(o1 + off) = acc;
                    *reinterpret_cast<u32x4*>(o2 + off) = acc;
                }
            }
        } else {
            const int m = lane & 31, h = lane >> 5;
            const int64_t row = rb + m;
            if (row < r1) {
                const size_t off = (size_t)row * 1536 + cb * 256 + (32 * wave + 16 * h) * 2;
                reinterpret_cast<u32x4*>(o1 + off)[0] = acc; reinterpret_cast<u32x4*>(o1 + off)[1] = acc;
                reinterpret_cast<u32x4*>(o2 + off)[0] = acc; reinterpret_cast<u32x4*>(o2 + off)[1] = acc;
            }
        }
    }
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdefu) sink[0] = acc[2];
}

template <class F> static double time_us(F&& launch, int reps, void* flush, size_t flush_bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> t;
    for (int i = 0; i < reps + 2; ++i) {
        if (flush) hipMemsetAsync(flush, i, flush_bytes, 0);
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (i >= 2) t.push_back(ms * 1e3);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const int64_t Ms[] = {28000, 18250};
    uint8_t* buf[5]; uint32_t* sink; void* fl;
    for (auto& p : buf) { hipMalloc(&p, (size_t)64 << 20); hipMemset(p, 1, (size_t)64 << 20); }
    hipMalloc(&sink, 64); hipMalloc(&fl, (size_t)1 << 30);
    for (int64_t M : Ms) {
        const int chunks = 42;
        const int64_t per = ((M + 31) / 32 + chunks - 1) / chunks * 32;
        const double mb = M * 1536.0 / 1e6;
        for (int flush = 0; flush < 2; ++flush) {
            printf("---- M = %lld rows (%.0f MB per tensor), 252 workgroups of 4 waves, %s\n", (long long)M, mb, flush ? "cold (1 GiB memset before each launch)" : "warm");
#define RUN(FULL, READ, name) { double t = time_us([&] { hipLaunchKernelGGL((k_cols_store<FULL, READ>), dim3(6 * chunks), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], M, per, sink); }, 9, flush ? fl : nullptr, (size_t)1 << 30); \
            printf("%-34s %7.1f us  %5.2f TB/s\n", name, t, (READ ? 5 : 2) * mb / t); }
            RUN(false, false, "2 writes, 16 B x 2 per lane (cols)") RUN(true, false, "2 writes, whole lines (full)")
            RUN(false, true, "3 reads + 2 writes (cols)") RUN(true, true, "3 reads + 2 writes (full)")
        }
    }
    return 0;
}
