// Position / order branch of the visual embedding (K4):
//     R[b, n, :] = LN_p( W_p [x1, x2, y1, y2, area] + b_p ) + img_order_embedding[img_id[b, n]] + obj_order_embedding[V - 1 - obj_id[b, n]]
// src/modeling_bart.py:129-141 (get_area), :162-166 (cat + absolute_vis_pos_embedding = Linear(5 -> d) + LayerNorm), :168-183 (the two
// order lookups; the object ids index the shared token table from its END); T5: src/modeling_t5.py:109-122, :143-147, :149-165 with
// T5LayerNorm (no mean, no bias).  R is what the feature-projection kernels (visproj*.hip) add behind their own LayerNorm.
//
// Rounds 1-5 left this branch to library ops: a K = 5 fp32 GEMM, an fp32 LayerNorm, two gathers, two adds, a cast -- and in the
// backward an fp32 copy of dout, LayerNorm's two backward kernels, batch sums for the broadcast order tables, a K = M GEMM for the
// 5-wide weight: ~15 launches and ~285 us of a 17 ms step for rows that are functions of FIVE numbers each.  Here:
//   forward  one wave per row, a lane owns 4 NG columns (c = 256 j + 4 lane + e) and keeps W_p, b_p, gamma, beta of them in registers;
//            the row's five inputs are wave-uniform loads, the statistics two wave reductions, the table rows L2 hits; R leaves in
//            the IO dtype (one rounding of the fp32 value, as the library chain's final cast) as whole 128-byte lines.
//   backward dR = the visual embedding's dout (R is added behind the feature branch's norm).  Same ownership: a wave re-derives
//            x_hat of its row from the five inputs, reduces sum(g) and sum(g x_hat) across the wave, and accumulates per COLUMN in
//            registers what every parameter needs: dbeta += dR, dgamma += dR x_hat, db_p += dpre, dW_p[:, k] += dpre p_k, and
//            dimg[i] += dR where the row's image id is i (n_images <= 4).  A workgroup's four waves meet in LDS, each workgroup
//            leaves (8 + NI) x d fp32 partials, vispos_finalize_kernel sums them in workgroup order (deterministic) into the
//            caller's gradient tensors.  No gradient for the object-order table here (the shared token table is frozen in every
//            launch script; a caller that trains it adds that one index_add itself).
#include "rowops.h"
#include "kernels.h"

namespace {

constexpr int VP_WAVES = 4;

__device__ __forceinline__ void vp_tab4(bool bf, const void* tab, int64_t row, int d, int c, float* v) {
    if (bf) {                   // (wave-uniform)
        const u32x2 r = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(tab) + row * d + c);
        v[0] = __builtin_bit_cast(float, r[0] << 16); v[1] = __builtin_bit_cast(float, r[0] & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, r[1] << 16); v[3] = __builtin_bit_cast(float, r[1] & 0xffff0000u);
    } else {
        const f32x4 r = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(tab) + row * d + c);
        v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3];
    }
}
__device__ __forceinline__ int64_t vp_id(const int64_t* ids, int64_t bstride, int64_t r, int N, int64_t dflt, int64_t hi) {
    int64_t id = dflt;
    if (ids != nullptr) { const uint32_t rr = (uint32_t)r, bb = rr / (uint32_t)N; id = ids[(int64_t)bb * bstride + (rr - bb * (uint32_t)N)]; }   // (M < 2^31: the entry points check)
    return id < 0 ? 0 : (id > hi ? hi : id);
}

// the row's five inputs (wave-uniform) and the pre-norm values / statistics of the lane's columns
struct VpRow { float p[5]; };
__device__ __forceinline__ VpRow vp_row(const float* pos, int64_t r) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(pos + r * 4);            // (x1, x2, y1, y2)
    VpRow o;
    o.p[0] = q[0]; o.p[1] = q[1]; o.p[2] = q[2]; o.p[3] = q[3];
    o.p[4] = (q[3] - q[2]) * (q[1] - q[0]);                                   // area = height * width (src/modeling_bart.py:137-140)
    return o;
}
// RB sums at once: the shuffle chains of the rows interleave (a wave that works on one row at a time waits out every step of a chain)
template <int RB> __device__ __forceinline__ void wave_sum_n(float (&v)[RB]) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
        for (int i = 0; i < RB; ++i) v[i] += __shfl_xor(v[i], o, 64);
}
// x_hat of the lane's columns for RB rows, and the rows' 1 / std (two-pass statistics, as the library kernel's)
template <int RB, int E>
__device__ __forceinline__ void vp_stats(const VpRow (&x)[RB], const float (&W)[E][5], const float (&Bc)[E], int d, float eps, bool rms,
                                         float (&xh)[RB][E], float (&rstd)[RB]) {
    float s[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        s[i] = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            float acc = Bc[e];
#pragma unroll
            for (int k = 0; k < 5; ++k) acc = fmaf(x[i].p[k], W[e][k], acc);
            xh[i][e] = acc;
            s[i] += acc;
        }
    }
    const float inv_d = 1.0f / (float)d;
    if (!rms) wave_sum_n<RB>(s);
    float q[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const float mean = rms ? 0.f : s[i] * inv_d;
        q[i] = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) { xh[i][e] -= mean; q[i] = fmaf(xh[i][e], xh[i][e], q[i]); }
    }
    wave_sum_n<RB>(q);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        rstd[i] = 1.0f / sqrtf(q[i] * inv_d + eps);
#pragma unroll
        for (int e = 0; e < E; ++e) xh[i][e] *= rstd[i];
    }
}

template <int NG, typename IO>
__global__ __launch_bounds__(VP_WAVES * 64) void vispos_fwd_kernel(VisPosArgs a) {
    constexpr int E = 4 * NG;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int d = a.d;
    float W[E][5], Bc[E], G[E], Be[E];
#pragma unroll
    for (int j = 0; j < NG; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 256 * j + 4 * lane + e, i = 4 * j + e;
#pragma unroll
            for (int k = 0; k < 5; ++k) W[i][k] = a.w[c * 5 + k];
            Bc[i] = a.b[c]; G[i] = a.gamma[c]; Be[i] = a.beta ? a.beta[c] : 0.f;
        }
    const bool rms = a.rms != 0;
    // RB rows per trip, every row's loads and reduction chains independent of the others': one row at a time made the forward a chain of
    // (scalar load -> statistics -> id -> table row) latencies per row, 31 us for 18,700 rows (profiles/r06_vispos.txt).  Rows past the
    // end are worked on as copies of the last row and not stored (no branch around the wave reductions).
    constexpr int RB = 4;
    const int64_t stride = (int64_t)gridDim.x * VP_WAVES;
    for (int64_t r0 = (int64_t)blockIdx.x * VP_WAVES + wave; r0 < a.M; r0 += RB * stride) {
        int64_t rr[RB];
        VpRow x[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) { rr[i] = r0 + i * stride < a.M ? r0 + i * stride : a.M - 1; x[i] = vp_row(a.pos, rr[i]); }
        int64_t ii[RB], oi[RB];
        if (a.img_tab != nullptr) {
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                ii[i] = vp_id(a.img_ids, a.img_bstride, rr[i], a.N, 0, a.n_img - 1);
                oi[i] = a.obj_rows - 1 - vp_id(a.obj_ids, a.obj_bstride, rr[i], a.N, (int64_t)((uint32_t)rr[i] % (uint32_t)a.N), a.obj_rows - 1);
            }
        }
        float v[RB][E], rstd[RB];
        vp_stats<RB, E>(x, W, Bc, d, a.eps, rms, v, rstd);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
#pragma unroll
            for (int e = 0; e < E; ++e) v[i][e] = fmaf(v[i][e], G[e], Be[e]);
            if (a.img_tab != nullptr) {
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    float t0[4], t1[4];
                    vp_tab4(a.img_tab_bf16 != 0, a.img_tab, ii[i], d, 256 * j + 4 * lane, t0);
                    vp_tab4(a.obj_tab_bf16 != 0, a.obj_tab, oi[i], d, 256 * j + 4 * lane, t1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[i][4 * j + e] = (v[i][4 * j + e] + t0[e]) + t1[e];      // (pos + img) + obj: the reference's order of the adds
                }
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (r0 + i * stride < a.M) {
                IO* dst = reinterpret_cast<IO*>(a.out) + rr[i] * d + 4 * lane;
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    if constexpr (sizeof(IO) == 2) Piece<__bf16, 8>::store(dst + 256 * j, v[i] + 4 * j);
                    else Piece<float, 16>::store(dst + 256 * j, v[i] + 4 * j);
                }
            }
        }
    }
}

// accumulator kinds of a column: 0 dbeta, 1 dgamma, 2 db_p, 3..7 dW_p[:, k], 8.. dimg[i]
template <int NG, typename IO, int NI>
__global__ __launch_bounds__(VP_WAVES * 64) void vispos_bwd_kernel(VisPosArgs a) {
    constexpr int E = 4 * NG, NACC = 8 + NI;
    __shared__ float red[VP_WAVES][256 * NG];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int d = a.d;
    float W[E][5], Bc[E], G[E];
#pragma unroll
    for (int j = 0; j < NG; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 256 * j + 4 * lane + e, i = 4 * j + e;
#pragma unroll
            for (int k = 0; k < 5; ++k) W[i][k] = a.w[c * 5 + k];
            Bc[i] = a.b[c]; G[i] = a.gamma[c];
        }
    float acc[NACC][E];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int e = 0; e < E; ++e) acc[k][e] = 0.f;
    const bool rms = a.rms != 0;
    const float inv_d = 1.0f / (float)d;
    using P = Piece<IO, sizeof(IO) == 2 ? 8 : 16>;
    // RB rows per trip (independent loads and reduction chains, as in the forward); a row past the end is a copy of the last row with a
    // zero gradient (it adds nothing).  The next trip's gradient rows are in flight while this trip is worked on.
    constexpr int RB = NG == 4 ? 1 : 2;          // (d = 1024: sixteen columns per lane -- 12 accumulator kinds of them leave room for one row at a time)
    const int64_t stride = (int64_t)gridDim.x * VP_WAVES;
    const int64_t first = (int64_t)blockIdx.x * VP_WAVES + wave;
    typename P::Raw nxt[RB][NG];
    auto fetch = [&](int64_t r0) {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int64_t r = r0 + i * stride < a.M ? r0 + i * stride : a.M - 1;
#pragma unroll
            for (int j = 0; j < NG; ++j) nxt[i][j] = P::load_raw_nt(reinterpret_cast<const IO*>(a.dout) + r * d + 256 * j + 4 * lane);
        }
    };
    if (first < a.M) fetch(first);
    for (int64_t r0 = first; r0 < a.M; r0 += RB * stride) {
        float dr[RB][E];
        int64_t rr[RB];
        VpRow x[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const bool live = r0 + i * stride < a.M;
            rr[i] = live ? r0 + i * stride : a.M - 1;
#pragma unroll
            for (int j = 0; j < NG; ++j) P::from_raw(nxt[i][j], dr[i] + 4 * j);
            if (!live) {
#pragma unroll
                for (int e = 0; e < E; ++e) dr[i][e] = 0.f;
            }
            x[i] = vp_row(a.pos, rr[i]);
        }
        if (r0 + RB * stride < a.M) fetch(r0 + RB * stride);
        float xh[RB][E], rstd[RB];
        vp_stats<RB, E>(x, W, Bc, d, a.eps, rms, xh, rstd);
        float ss[2 * RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            ss[2 * i] = ss[2 * i + 1] = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) { const float g = dr[i][e] * G[e]; ss[2 * i] += g; ss[2 * i + 1] = fmaf(g, xh[i][e], ss[2 * i + 1]); }
        }
        wave_sum_n<2 * RB>(ss);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const float s1 = rms ? 0.f : ss[2 * i] * inv_d, s2 = ss[2 * i + 1] * inv_d;
            int ii = 0;
            if constexpr (NI > 0) ii = (int)vp_id(a.img_ids, a.img_bstride, rr[i], a.N, 0, a.n_img - 1);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float dpre = rstd[i] * (dr[i][e] * G[e] - s1 - xh[i][e] * s2);
                acc[0][e] += dr[i][e];
                acc[1][e] = fmaf(dr[i][e], xh[i][e], acc[1][e]);
                acc[2][e] += dpre;
#pragma unroll
                for (int k = 0; k < 5; ++k) acc[3 + k][e] = fmaf(dpre, x[i].p[k], acc[3 + k][e]);
#pragma unroll
                for (int q = 0; q < NI; ++q) acc[8 + q][e] += ii == q ? dr[i][e] : 0.f;
            }
        }
    }
    // the four waves of the workgroup meet in LDS, one accumulator kind at a time; thread t sums column slots t, t + 256, ...
    float* part = a.partial + (size_t)blockIdx.x * NACC * d;
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const f32x4 v = {acc[k][4 * j], acc[k][4 * j + 1], acc[k][4 * j + 2], acc[k][4 * j + 3]};
            *reinterpret_cast<f32x4*>(&red[wave][256 * j + 4 * lane]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int c = 256 * j + (int)threadIdx.x;
            part[(size_t)k * d + c] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
        }
        __syncthreads();
    }
}

// out[kind][c] = sum over the workgroups' partials in a fixed order; scattered to the parameter layouts.  A workgroup = one accumulator
// kind x 16 columns; thread (c = t & 15, part = t >> 4) sums the workgroups b = part (mod 16) with every load in flight at once, the
// sixteen parts meet in LDS.  (The first version gave a thread one whole column: 7,680 threads walking 256 partials each, four loads at
// a time -- the finalize launch took longer than the pass that produced the partials.)
constexpr int VP_FIN_MAXB = 16;         // partials per part: vispos_grid(., bwd) <= 256 workgroups
__global__ __launch_bounds__(256) void vispos_finalize_kernel(VisPosArgs a, int nblocks, int nacc) {
    __shared__ float red[16][17];
    const int cg = (int)blockIdx.x, k = (int)blockIdx.y;
    const int cl = (int)threadIdx.x & 15, part = (int)threadIdx.x >> 4;
    const int c = 16 * cg + cl;
    float v[VP_FIN_MAXB];
#pragma unroll
    for (int i = 0; i < VP_FIN_MAXB; ++i) {
        const int b = part + 16 * i;
        v[i] = b < nblocks ? a.partial[((size_t)b * nacc + k) * a.d + c] : 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VP_FIN_MAXB; ++i) s += v[i];
    red[part][cl] = s;
    __syncthreads();
    if (part != 0) return;
    s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += red[q][cl];
    if (k == 0) { if (a.dbeta) a.dbeta[c] = s; }
    else if (k == 1) a.dgamma[c] = s;
    else if (k == 2) a.db[c] = s;
    else if (k < 8) a.dw[c * 5 + (k - 3)] = s;
    else if (k - 8 < a.n_img) a.dimg[(size_t)(k - 8) * a.d + c] = s;
}

template <int NG, typename IO>
hipError_t fwd_t(const VisPosArgs& a, unsigned grid, hipStream_t s) {
    hipLaunchKernelGGL((vispos_fwd_kernel<NG, IO>), dim3(grid), dim3(VP_WAVES * 64), 0, s, a);
    return hipGetLastError();
}
template <int NG, typename IO>
hipError_t bwd_t(const VisPosArgs& a, unsigned grid, int ni, hipStream_t s) {
    if (ni == 0) hipLaunchKernelGGL((vispos_bwd_kernel<NG, IO, 0>), dim3(grid), dim3(VP_WAVES * 64), 0, s, a);
    else if (ni == 2) hipLaunchKernelGGL((vispos_bwd_kernel<NG, IO, 2>), dim3(grid), dim3(VP_WAVES * 64), 0, s, a);
    else hipLaunchKernelGGL((vispos_bwd_kernel<NG, IO, 4>), dim3(grid), dim3(VP_WAVES * 64), 0, s, a);
    return hipGetLastError();
}

}  // namespace

bool vispos_applies(int d, int n_img) { return d > 0 && d % 256 == 0 && d <= 1024 && n_img >= 0 && n_img <= 4; }
int vispos_img_slots(int n_img) { return n_img == 0 ? 0 : (n_img <= 2 ? 2 : 4); }
// workgroups: every wave gets a few rows (the per-lane weights are loaded once per wave); forward at most two workgroups per CU's worth,
// backward one (its accumulators leave room for one wave per SIMD: more workgroups would only queue)
static unsigned vispos_grid(int64_t M, bool bwd) {
    int64_t g = (M + VP_WAVES * 8 - 1) / (VP_WAVES * 8);
    const int64_t cap = bwd ? 256 : 512;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}
size_t vispos_bwd_workspace_bytes(int64_t M, int d, int n_img) {
    return (size_t)vispos_grid(M, true) * (size_t)(8 + vispos_img_slots(n_img)) * (size_t)d * sizeof(float);
}

hipError_t launch_vispos_fwd(const VisPosArgs& a, int io_fp32, hipStream_t stream) {
    const unsigned grid = vispos_grid(a.M, false);
    switch (a.d / 256) {
        case 1: return io_fp32 ? fwd_t<1, float>(a, grid, stream) : fwd_t<1, __bf16>(a, grid, stream);
        case 2: return io_fp32 ? fwd_t<2, float>(a, grid, stream) : fwd_t<2, __bf16>(a, grid, stream);
        case 3: return io_fp32 ? fwd_t<3, float>(a, grid, stream) : fwd_t<3, __bf16>(a, grid, stream);
        default: return io_fp32 ? fwd_t<4, float>(a, grid, stream) : fwd_t<4, __bf16>(a, grid, stream);
    }
}

hipError_t launch_vispos_bwd(const VisPosArgs& a, int io_fp32, hipStream_t stream) {
    const unsigned grid = vispos_grid(a.M, true);
    const int ni = vispos_img_slots(a.n_img);
    hipError_t e;
    switch (a.d / 256) {
        case 1: e = io_fp32 ? bwd_t<1, float>(a, grid, ni, stream) : bwd_t<1, __bf16>(a, grid, ni, stream); break;
        case 2: e = io_fp32 ? bwd_t<2, float>(a, grid, ni, stream) : bwd_t<2, __bf16>(a, grid, ni, stream); break;
        case 3: e = io_fp32 ? bwd_t<3, float>(a, grid, ni, stream) : bwd_t<3, __bf16>(a, grid, ni, stream); break;
        default: e = io_fp32 ? bwd_t<4, float>(a, grid, ni, stream) : bwd_t<4, __bf16>(a, grid, ni, stream); break;
    }
    if (e != hipSuccess) return e;
    const int nacc = 8 + ni;
    hipLaunchKernelGGL(vispos_finalize_kernel, dim3((unsigned)(a.d / 16), (unsigned)nacc), dim3(256), 0, stream, a, (int)grid, nacc);
    return hipGetLastError();
}
