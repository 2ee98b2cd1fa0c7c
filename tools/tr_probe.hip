// Probe the lane/element mapping of gfx950's ds_read_b64_tr_b16 (LDS transpose read).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/_tr_probe tools/tr_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4s __attribute__((ext_vector_type(4)));

__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    int elem;       // element index of the lane's 8-byte read
    if (mode == 0) elem = 4 * l;                                  // contiguous: lane l -> elements 4l..4l+3
    else if (mode == 1) elem = (l & 15) * 64 + (l >> 4) * 4;      // row-major [16 rows][64 cols]: lane t of a group -> row t, 4 cols at 4*(l>>4)
    else elem = (l & 3) * 64 + ((l >> 2) & 3) * 4 + (l >> 4) * 16;   // 4x4 tiles: lanes 0-3 rows 0-3 of tile col 0, ...
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + elem));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)r[j];
}

int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" %5d", h[l * 4 + j]);
            printf("%s", (l % 4 == 3) ? "\n" : "  |");
        }
    }
    return 0;
}
