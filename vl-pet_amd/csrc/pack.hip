// Weight re-pack: (down [r,d] as N_h head blocks, up [d,r], biases) -> MFMA fragment order.
// Layout specification: vl-pet_amd/packing.py (pack_down / pack_up / pack_up_t / pack_down_t).
// ~0.6 MB per pair, once per optimizer step; also performs the fp32 -> bf16 cast (NS = 1) or the
// bf16 hi/lo split (NS = 2), so no separate cast pass over the parameters exists.
#include "common.h"
#include "kernels.h"

__device__ __forceinline__ int pi_d(int ct, int i) {
    int b = i >> 3, hp = (i >> 2) & 1, a = i & 3;
    return 32 * ct + 16 * (b >> 1) + 8 * hp + 4 * (b & 1) + a;
}
__device__ __forceinline__ int pi_u(int nt, int i) {
    int b = i >> 3, hp = (i >> 2) & 1, a = i & 3;
    return 64 * (nt >> 1) + 32 * hp + 16 * (nt & 1) + 4 * b + a;
}

__device__ __forceinline__ float ld_src(const void* p, int64_t idx, int bf16_src) {
    return bf16_src ? (float)reinterpret_cast<const __bf16*>(p)[idx] : reinterpret_cast<const float*>(p)[idx];
}

__device__ __forceinline__ float fetch_down(const PackArgs& a, int c, int k) {
    if (c >= a.r) return 0.f;
    int head = c / a.rows_per_head;
    int cc = c - head * a.rows_per_head;
    return ld_src(a.wd[head], (int64_t)cc * a.d + k, a.src_bf16);
}
__device__ __forceinline__ float fetch_up(const PackArgs& a, int f, int c) {
    if (c >= a.r) return 0.f;
    return ld_src(a.wu, (int64_t)f * a.r + c, a.src_bf16);
}

template <int NS>
__global__ __launch_bounds__(256) void pack_pair_kernel(PackArgs a) {
    const int RT = a.RT, d = a.d, KT = 2 * RT;
    const int NF = d / 16 * RT;
    const int64_t slots = (int64_t)4 * NF * 64;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const PackGeom g = pack_geom(RT, d, NS);
    if (gid < slots) {
        const int lane = (int)(gid & 63);
        const int frag = (int)((gid >> 6) % NF);
        const int pack = (int)((gid >> 6) / NF);
        const int i = lane & 31, hh = lane >> 5;
        float v[8];
        if (pack == 0) {
            int ct = frag % RT, u = (frag / RT) & 3, t = frag / (4 * RT);
            int c = pi_d(ct, i);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fetch_down(a, c, 64 * t + 32 * hh + 8 * u + j);
        } else if (pack == 1) {
            int ks = frag % KT, nt = frag / KT;
            int f = pi_u(nt, i);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fetch_up(a, f, 16 * ks + 8 * hh + j);
        } else if (pack == 2) {
            int ct = frag % RT, e = (frag / RT) & 1, nt = frag / (2 * RT);
            int c = pi_d(ct, i);
            int fb = 64 * (nt >> 1) + 32 * hh + 16 * (nt & 1) + 8 * e;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fetch_up(a, fb + j, c);
        } else {
            int ks = frag % KT, nt = frag / KT;
            int k = pi_u(nt, i);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fetch_down(a, 16 * ks + 8 * hh + j, k);
        }
        Frag<NS> f = frag_from_f32<NS>(v);
        uint8_t* base = a.out + (int64_t)pack * g.pack_bytes;
#pragma unroll
        for (int p = 0; p < NS; ++p)
            *reinterpret_cast<bf16x8*>(base + ((int64_t)(frag * NS + p) * 64 + lane) * 16) = f.p[p];
    }
    // biases (fp32, zero padded)
    const int64_t bid = gid - slots;
    if (bid >= 0 && bid < 32 * RT + d) {
        float* bout = reinterpret_cast<float*>(a.out + g.bias_off);
        float val = 0.f;
        if (bid < 32 * RT) {
            int c = (int)bid;
            if (c < a.r && a.bd[0] != nullptr) {
                int head = c / a.rows_per_head;
                val = ld_src(a.bd[head], c - head * a.rows_per_head, a.src_bf16);
            }
        } else {
            int f = (int)bid - 32 * RT;
            if (a.bu != nullptr) val = ld_src(a.bu, f, a.src_bf16);
        }
        bout[bid] = val;
    }
}

hipError_t launch_pack_pair(const PackArgs& a, int NS, hipStream_t stream) {
    const int NF = a.d / 16 * a.RT;
    const int64_t total = (int64_t)4 * NF * 64 + 32 * a.RT + a.d;
    const int blocks = (int)((total + 255) / 256);
    if (NS == 1) hipLaunchKernelGGL(pack_pair_kernel<1>, dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(pack_pair_kernel<2>, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}
