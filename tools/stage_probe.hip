// Micro-probe: cost of one "down stage" instruction stream (LDS fragment reads + MFMAs + barrier) with
// all data already in LDS -- no global traffic.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/_stage_probe tools/stage_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MODE bit0: MFMAs, bit1: LDS reads, bit2: barrier per stage, bit3: sched_barrier pipelining off (plain order)
template <int WAVES, int NT, int MODE>      // NT = c-tiles per wave (6: both chains, 3: one chain)
__global__ __launch_bounds__(WAVES * 64) void probe(float* out, int stages, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 40 * 1024 / 4; i += WAVES * 64) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x16 acc[NT];
    for (int c = 0; c < NT; ++c) for (int j = 0; j < 16; ++j) acc[c][j] = 0.f;
    const uint8_t* w = smem;                               // 24 KiB of "weights"
    const uint8_t* rows = smem + 24 * 1024 + (wave & 3) * 4096;      // this wave's 32 x 128 B
    const int m = lane & 31, h = lane >> 5;
    bf16x8 bk[4], wk[4][NT];
    for (int u = 0; u < 4; ++u) { for (int j = 0; j < 8; ++j) bk[u][j] = (__bf16)(0.01f * j); for (int c = 0; c < NT; ++c) wk[u][c] = bk[u]; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < stages; ++s) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (MODE & 2) {
                bk[u] = *reinterpret_cast<const bf16x8*>(rows + (m * 8 + ((2 * u + h) ^ ((m >> 1) & 7))) * 16);
#pragma unroll
                for (int c = 0; c < NT; ++c)
                    wk[u][c] = *reinterpret_cast<const bf16x8*>(w + ((u * NT + c) * 64 + lane) * 16 + (NT == 3 ? (wave >> 2) * 12288 : 0));
            }
            if constexpr (MODE & 1) {
#pragma unroll
                for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk[u][c], bk[u], acc[c], 0, 0, 0);
            }
        }
        if constexpr (MODE & 8) {      // epilogue-like VALU: 32 sigmoids (v_mul, v_exp, v_add, v_rcp) on accumulator values
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float a0 = acc[0][j], a1 = acc[NT - 1][j];
                acc[0][j] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a0 * -1.442695f));
                acc[NT - 1][j] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a1 * -1.442695f));
            }
        }
        if constexpr (MODE & 16) {     // same count of plain VALU (4 v_fma per element) instead of the transcendental pair
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float a0 = acc[0][j], a1 = acc[NT - 1][j];
#pragma unroll
                for (int q = 0; q < 4; ++q) { a0 = __builtin_fmaf(a0, 0.999f, 0.001f); a1 = __builtin_fmaf(a1, 0.999f, 0.001f); }
                acc[0][j] = a0; acc[NT - 1][j] = a1;
            }
        }
        if constexpr (MODE & 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int c = 0; c < NT; ++c) for (int j = 0; j < 16; ++j) sum += acc[c][j];
    for (int u = 0; u < 4; ++u) sum += (float)bk[u][0] + (float)wk[u][0][1];
    out[blockIdx.x * WAVES * 64 + tid] = sum;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int WAVES, int NT, int MODE>
static void run(const char* name, int blocks) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * WAVES * 64 * 4); hipMalloc(&cyc, blocks * 8);
    const int stages = 1000;
    auto k = probe<WAVES, NT, MODE>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(WAVES * 64), 150 * 1024, 0, out, stages, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(WAVES * 64), 150 * 1024, 0, out, stages, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4]; hipMemcpy(h, cyc, 8 * (blocks < 4 ? blocks : 4), hipMemcpyDeviceToHost);
    printf("%-44s blocks=%3d  %8.1f cycles/stage (memtime) %8.3f us/1000 stages -> %6.1f ns/stage\n", name, blocks,
           (double)h[0] / stages, ms * 1e3, ms * 1e6 / stages);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {16, 219}) {
        run<4, 6, 1>("4w x 24 MFMA", blocks);
        run<4, 6, 2>("4w x 40 ds_read_b128", blocks);
        run<4, 6, 3>("4w x (40 reads + 24 MFMA)", blocks);
        run<4, 6, 7>("4w x (40 reads + 24 MFMA) + barrier", blocks);
        run<8, 3, 1>("8w x 12 MFMA", blocks);
        run<8, 3, 2>("8w x 16 ds_read_b128", blocks);
        run<8, 3, 3>("8w x (16 reads + 12 MFMA)", blocks);
        run<8, 3, 7>("8w x (16 reads + 12 MFMA) + barrier", blocks);
        run<4, 6, 8>("4w x 32 sigmoid only", blocks);
        run<4, 6, 16>("4w x 128 v_fma only", blocks);
        run<4, 6, 9>("4w x (24 MFMA + 32 sigmoid)", blocks);
        run<4, 6, 17>("4w x (24 MFMA + 128 v_fma)", blocks);
        run<8, 3, 9>("8w x (12 MFMA + 32 sigmoid)", blocks);
    }
    return 0;
}
