// Weight re-pack: (down [r,d] as N_h head blocks, up [d,r], biases) -> MFMA fragment order for the
// 32x32x16 kernels.  Layout specification: tests/packing_spec.py section v4 (pack_down4 / pack_up4 /
// pack_up_t4 / pack_down_t4), checked lane by lane in tests/test_layout_model32.py.
// ~0.6 MB per pair, once per optimizer step; also performs the fp32 -> bf16 cast (NS = 1) or the
// bf16 hi/lo split (NS = 2), so no separate cast pass over the parameters exists.
#include "common.h"
#include "kernels.h"

__device__ __forceinline__ float ld_src(const void* p, int64_t idx, int bf16_src) {
    return bf16_src ? (float)reinterpret_cast<const __bf16*>(p)[idx] : reinterpret_cast<const float*>(p)[idx];
}
__device__ __forceinline__ float fetch_down(const PackArgs& a, int c, int k) {
    if (c >= a.r || a.wd[0] == nullptr) return 0.f;      // (no down weight: the up-side pack of the low-rank visual projector)
    int head = c / a.rows_per_head;
    int cc = c - head * a.rows_per_head;
    return ld_src(a.wd[head], (int64_t)cc * a.d + k, a.src_bf16);
}
__device__ __forceinline__ float fetch_up(const PackArgs& a, int f, int c) {
    if (c >= a.r || a.wu == nullptr) return 0.f;
    return ld_src(a.wu, (int64_t)f * a.r + c, a.src_bf16);
}

template <int NS>
__device__ __forceinline__ void pack_body(const PackArgs& a, int64_t gid) {
    constexpr int FE = 64 / NS, KU = FE / 16, NV = FE / 32, LW = FE / 2, E4 = FE / 16;
    const int RT = a.RT, d = a.d, KT = 2 * RT;
    const int NF = d / 16 * RT;                 // fragments per pack
    const int64_t slots = (int64_t)a.n_packs * NF * 64;
    const PackGeom geo = pack_geom(RT, d, NS);
    if (gid < slots) {
        const int lane = (int)(gid & 63);
        const int frag = (int)((gid >> 6) % NF);
        const int pack = (int)((gid >> 6) / NF);
        const int i = lane & 31, hh = lane >> 5;
        const int b = i >> 3, hp = (i >> 2) & 1, aa = i & 3;
        const int crow = 16 * (b >> 1) + 8 * hp + 4 * (b & 1) + aa;   // + 32ct : bottleneck index of MFMA row i
        const int frow = LW * hp + 4 * b + aa;                          // + FE*stage + 16v : feature of MFMA row i
        float v[8];
        if (pack == 0) {            // down: (stage, u, ct)
            const int per = KU * RT, st = frag / per, rem = frag % per;
            const int u = rem / RT, ct = rem % RT;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fetch_down(a, 32 * ct + crow, FE * st + 16 * u + 8 * hh + j);
        } else if (pack == 1) {     // up: (stage, v, ks)
            const int per = NV * KT, st = frag / per, rem = frag % per;
            const int vv = rem / KT, ks = rem % KT;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fetch_up(a, FE * st + 16 * vv + frow, 16 * ks + 8 * hh + j);
        } else if (pack == 2) {     // up_t: (stage, e, ct)
            const int per = E4 * RT, st = frag / per, rem = frag % per;
            const int e = rem / RT, ct = rem % RT;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fetch_up(a, FE * st + LW * hh + 8 * e + j, 32 * ct + crow);
        } else {                    // down_t: (stage, v, ks)
            const int per = NV * KT, st = frag / per, rem = frag % per;
            const int vv = rem / KT, ks = rem % KT;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fetch_down(a, 16 * ks + 8 * hh + j, FE * st + 16 * vv + frow);
        }
        Frag<NS> f = frag_from_f32<NS>(v);
        uint8_t* base = a.out + (int64_t)pack * geo.pack_bytes;
#pragma unroll
        for (int p = 0; p < NS; ++p)
            *reinterpret_cast<bf16x8*>(base + ((int64_t)(frag * NS + p) * 64 + lane) * 16) = f.p[p];
    }
    // biases (fp32, zero padded)
    const int64_t bid = gid - slots;
    if (bid >= 0 && bid < 32 * RT + d) {
        float* bout = reinterpret_cast<float*>(a.out + (a.n_packs == 4 ? geo.bias_off : geo.pack_bytes));
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

template <int NS>
__global__ __launch_bounds__(256) void pack_pair_kernel(PackArgs a) {
    pack_body<NS>(a, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// several pairs of the same geometry in one launch (blockIdx.y = pair): the ~30 packs of a train step are latency-bound
// 7.6 us launches each (0.8 % of the step); batched they are a handful
template <int NS>
__global__ __launch_bounds__(256) void pack_pairs_kernel(PackBatch b) {
    pack_body<NS>(b.p[blockIdx.y], (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

hipError_t launch_pack_pairs(const PackBatch& b, int NS, hipStream_t stream) {
    const PackArgs& a = b.p[0];
    const int NF = a.d / 16 * a.RT;
    const int64_t total = (int64_t)a.n_packs * NF * 64 + 32 * a.RT + a.d;
    const int blocks = (int)((total + 255) / 256);
    if (NS == 1) hipLaunchKernelGGL(pack_pairs_kernel<1>, dim3(blocks, b.n), dim3(256), 0, stream, b);
    else hipLaunchKernelGGL(pack_pairs_kernel<2>, dim3(blocks, b.n), dim3(256), 0, stream, b);
    return hipGetLastError();
}

hipError_t launch_pack_pair(const PackArgs& a, int NS, hipStream_t stream) {
    const int NF = a.d / 16 * a.RT;
    const int64_t total = (int64_t)a.n_packs * NF * 64 + 32 * a.RT + a.d;
    const int blocks = (int)((total + 255) / 256);
    if (NS == 1) hipLaunchKernelGGL(pack_pair_kernel<1>, dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(pack_pair_kernel<2>, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}
