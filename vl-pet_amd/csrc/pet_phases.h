// Shared phases of the fused PET kernels: the weight-fragment stage pipeline and the
// down-projection phase (x rows as MFMA B operand, weights streamed through LDS).
#pragma once
#include "common.h"

struct StageDesc {
    const uint8_t* p0; int u0;   // first segment: pointer, 16-byte units
    const uint8_t* p1; int u1;   // second segment (gate chain) or empty
};

// Load one 64-wide super-step of a row as four B fragments: lane (m,h) reads
// x[row][64t + 32h .. +32) (64 B bf16 / 128 B fp32, contiguous), piece u = elements [8u, 8u+8).
template <typename IO, bool DROP>
__device__ __forceinline__ void load_x_step(const IO* p, const uint8_t* keep, float keep_scale,
                                            Frag<IoTraits<IO>::NS> (&xf)[4]) {
    constexpr int NS = IoTraits<IO>::NS;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if constexpr (!DROP) {
            xf[u] = load_frag8(p + 8 * u);
        } else {
            float v[8];
            load8_f32(p + 8 * u, v);
            const uint64_t k = *reinterpret_cast<const uint64_t*>(keep + 8 * u);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ((k >> (8 * j)) & 0xff) ? v[j] * keep_scale : 0.f;
            xf[u] = frag_from_f32<NS>(v);
        }
    }
}

// Down projection of one chain for the wave's 32 rows:
//   pre[c] = sum_k W[c,k] x[m,k] + b[c];  z = act(pre)  ->  B fragments of the up projection.
// `Ctx` supplies the stage stream: stage(s) -> StageDesc, buf(i) -> LDS stage buffer, T, tid.
// On return s has advanced by T stages and the next stage is resident in buf(s & 1).
template <typename IO, int RT, bool ACT_ID, bool WANT_GRAD, bool DROP, class Ctx>
__device__ __forceinline__ void down_phase(Ctx& c, int& s, const IO* xrow, const uint8_t* keeprow,
                                           float keep_scale, const float* sbias_h, int lane,
                                           Frag<IoTraits<IO>::NS> (&z)[2 * RT], f32x16 (&gp)[RT]) {
    constexpr int NS = IoTraits<IO>::NS;
    constexpr int MAXU = RT * NS;
    f32x16 acc[RT];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) acc[ct] = zero16();
    Frag<NS> xf[4], xn[4];
    load_x_step<IO, DROP>(xrow, keeprow, keep_scale, xf);
    for (int t = 0; t < c.T; ++t) {
        StageRegs<MAXU> sr;
        const StageDesc nx = c.stage(s + 1);
        stage_load<MAXU>(sr, nx.p0, nx.u0, nx.p1, nx.u1, c.tid);
        if (t + 1 < c.T)
            load_x_step<IO, DROP>(xrow + 64 * (t + 1), DROP ? keeprow + 64 * (t + 1) : keeprow, keep_scale, xn);
        const uint8_t* b = c.buf(s & 1);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int ct = 0; ct < RT; ++ct)
                acc[ct] = mfma_ns<NS>(lds_frag<NS>(b, u * RT + ct, lane), xf[u], acc[ct]);
        }
        stage_store<MAXU>(sr, c.buf((s + 1) & 1), nx.u0 + nx.u1, c.tid);
        __syncthreads();
        ++s;
        if (t + 1 < c.T) {
#pragma unroll
            for (int u = 0; u < 4; ++u) xf[u] = xn[u];
        }
    }
    // bias + activation; register 8*sh + j of c-tile ct is bottleneck index 32ct + 16sh + 8h + j
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) {
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float pre = acc[ct][8 * sh + j] + sbias_h[32 * ct + 16 * sh + j];
                if constexpr (ACT_ID) {
                    if constexpr (WANT_GRAD) gp[ct][8 * sh + j] = 1.0f;
                    v[j] = pre;
                } else {
                    if constexpr (WANT_GRAD) gp[ct][8 * sh + j] = gelu_new_grad_f(pre);
                    v[j] = gelu_new_f(pre);
                }
            }
            z[2 * ct + sh] = frag_from_f32<NS>(v);
        }
    }
}

// value of a fragment as fp32 (hi + lo)
template <int NS>
__device__ __forceinline__ void frag_to_f32(const Frag<NS>& f, float* v) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = (float)f.p[0][j];
        if constexpr (NS == 2) v[j] += (float)f.p[1][j];
    }
}
