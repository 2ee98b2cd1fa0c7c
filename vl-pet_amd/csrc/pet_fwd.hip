// Fused PET forward (K1 encoder adapter+gate, K2 decoder value-parallel adapter, K3 LoRA delta):
//
//   out = ( s2*res + sd*( up_A( act( down_A(xa) ) ) ) )  [ (*|+) sigmoid( up_G( gelu_new( down_G(xg) ) ) ) ] * gs
//
// Reference op chains replaced: my_transformers/modeling_bart.py:1147-1155,1195-1209,1256-1257
// (K1), adapters/adapter_modeling.py:55-61 + adapter_controller.py:149-162 (K2),
// lora/controller.py:56-70 (K3, on top of the PyTorch base GEMM).
//
// One workgroup = 4 waves = 128 rows; each wave carries its 32 rows through the whole chain in
// registers (down-projection accumulators -> bias/gelu -> bf16 B fragments -> up-projection ->
// residual/gate epilogue), the weights arrive as pre-packed MFMA A fragments through a
// double-buffered LDS stage stream shared by the 4 waves.  HBM traffic = read xa (=res), read xg,
// write out: the algorithmic 3*d*M elements.
#include "common.h"
#include "kernels.h"
#include "pet_phases.h"

template <int NS, int RT, bool GATE>
struct FwdCtx {
    static constexpr int FB = NS * 1024;
    static constexpr int STAGE_B = 4 * RT * FB;
    const uint8_t* pk_a;
    const uint8_t* pk_g;
    uint8_t* smem;
    int64_t pack_bytes;
    int tid, T, NT;
    __device__ __forceinline__ uint8_t* buf(int i) const { return smem + i * STAGE_B; }
    __device__ __forceinline__ StageDesc stage(int s) const {
        StageDesc r{pk_a, 0, pk_a, 0};   // empty stages keep a valid (never stored) address
        if (s < T) { r.p0 = pk_a + (int64_t)s * STAGE_B; r.u0 = STAGE_B / 16; return r; }
        s -= T;
        if constexpr (GATE) {
            if (s < T) { r.p0 = pk_g + (int64_t)s * STAGE_B; r.u0 = STAGE_B / 16; return r; }
            s -= T;
        }
        if (s < NT) {
            r.p0 = pk_a + pack_bytes + (int64_t)s * 2 * RT * FB; r.u0 = 2 * RT * FB / 16;
            if constexpr (GATE) { r.p1 = pk_g + pack_bytes + (int64_t)s * 2 * RT * FB; r.u1 = r.u0; }
        }
        return r;
    }
};

template <typename IO, int RT, bool GATE, bool ACT_ID, bool DROP>
__global__ __launch_bounds__(VLPET_THREADS) void pet_fwd_kernel(PetFwdArgs a) {
    constexpr int NS = IoTraits<IO>::NS;
    constexpr int KT = 2 * RT;
    constexpr int MAXU = RT * NS;
    using Ctx = FwdCtx<NS, RT, GATE>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int d = a.d;
    const int64_t row_raw = (int64_t)blockIdx.x * VLPET_ROWS_PER_WG + wave * 32 + m;
    const bool row_ok = row_raw < a.M;
    const int64_t row = row_ok ? row_raw : a.M - 1;

    const PackGeom g = pack_geom(RT, d, NS);
    Ctx c;
    c.pk_a = a.pk_a; c.pk_g = a.pk_g; c.smem = smem; c.pack_bytes = g.pack_bytes;
    c.tid = tid; c.T = d / 64; c.NT = d / 32;

    // biases -> LDS: [bdA(32RT) | buA(d) | bdG(32RT) | buG(d)]
    float* sb = reinterpret_cast<float*>(smem + 2 * Ctx::STAGE_B);
    const int nb = 32 * RT + d;
    {
        const float* ba = reinterpret_cast<const float*>(a.pk_a + g.bias_off);
        for (int i = tid; i < nb; i += VLPET_THREADS) sb[i] = ba[i];
        if constexpr (GATE) {
            const float* bg = reinterpret_cast<const float*>(a.pk_g + g.bias_off);
            for (int i = tid; i < nb; i += VLPET_THREADS) sb[nb + i] = bg[i];
        }
    }
    // stage 0
    {
        StageRegs<MAXU> sr;
        const StageDesc s0 = c.stage(0);
        stage_load<MAXU>(sr, s0.p0, s0.u0, s0.p1, s0.u1, tid);
        stage_store<MAXU>(sr, c.buf(0), s0.u0 + s0.u1, tid);
    }
    __syncthreads();

    int s = 0;
    const IO* xa = reinterpret_cast<const IO*>(a.xa) + row * d + 32 * h;
    const uint8_t* keeprow = DROP ? a.keep + row * d + 32 * h : nullptr;
    Frag<NS> zA[KT];
    f32x16 gp_unused[RT];
    down_phase<IO, RT, ACT_ID, false, DROP>(c, s, xa, keeprow, a.keep_scale, sb + 8 * h, lane, zA, gp_unused);
    Frag<NS> zG[GATE ? KT : 1];
    if constexpr (GATE) {
        const IO* xg = reinterpret_cast<const IO*>(a.xg) + row * d + 32 * h;
        down_phase<IO, RT, false, false, false>(c, s, xg, nullptr, 1.f, sb + nb + 8 * h, lane, zG, gp_unused);
    }

    const IO* res = reinterpret_cast<const IO*>(a.res) + row * d;
    IO* out = reinterpret_cast<IO*>(a.out) + row * d;
    const float* sbu = sb + 32 * RT;
    const float* sbgu = sb + nb + 32 * RT;
    const float s2 = a.s2, sd = a.sd, gs = a.gs;
    const bool gate_add = (a.flags & PET_GATE_ADD) != 0;

    for (int nt = 0; nt < c.NT; ++nt) {
        StageRegs<MAXU> sr;
        const StageDesc nx = c.stage(s + 1);
        stage_load<MAXU>(sr, nx.p0, nx.u0, nx.p1, nx.u1, tid);
        const int f0 = 64 * (nt >> 1) + 32 * h + 16 * (nt & 1);
        float r[16];
        load8_f32(res + f0, r);
        load8_f32(res + f0 + 8, r + 8);
        const uint8_t* b = c.buf(s & 1);
        f32x16 aA = zero16(), aG = zero16();
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) aA = mfma_ns<NS>(lds_frag<NS>(b, ks, lane), zA[ks], aA);
        if constexpr (GATE) {
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) aG = mfma_ns<NS>(lds_frag<NS>(b, KT + ks, lane), zG[ks], aG);
        }
        float o[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float hv = s2 * r[i] + sd * (aA[i] + sbu[f0 + i]);
            if constexpr (GATE) {
                float gt = sigmoid_f(aG[i] + sbgu[f0 + i]);
                hv = gate_add ? hv + gt : hv * gt;
                hv *= gs;
            }
            o[i] = hv;
        }
        if (row_ok) {
            store8_f32(out + f0, o);
            store8_f32(out + f0 + 8, o + 8);
        }
        stage_store<MAXU>(sr, c.buf((s + 1) & 1), nx.u0 + nx.u1, tid);
        __syncthreads();
        ++s;
    }
}

template <typename IO, int RT, bool GATE, bool ACT_ID, bool DROP>
static hipError_t launch_one(const PetFwdArgs& a, hipStream_t stream) {
    constexpr int NS = IoTraits<IO>::NS;
    const size_t lds = 2 * (size_t)FwdCtx<NS, RT, GATE>::STAGE_B + (size_t)2 * (32 * RT + a.d) * 4;
    auto kern = pet_fwd_kernel<IO, RT, GATE, ACT_ID, DROP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int blocks = (int)((a.M + VLPET_ROWS_PER_WG - 1) / VLPET_ROWS_PER_WG);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(VLPET_THREADS), lds, stream, a);
    return hipGetLastError();
}

template <typename IO, int RT>
static hipError_t launch_rt(const PetFwdArgs& a, hipStream_t stream) {
    const bool gate = a.flags & PET_GATE, act_id = a.flags & PET_ACT_IDENTITY, drop = a.keep != nullptr;
    if (gate) return launch_one<IO, RT, true, false, false>(a, stream);
    if (act_id) return drop ? launch_one<IO, RT, false, true, true>(a, stream)
                            : launch_one<IO, RT, false, true, false>(a, stream);
    return launch_one<IO, RT, false, false, false>(a, stream);
}

template <typename IO>
static hipError_t launch_io(const PetFwdArgs& a, hipStream_t stream) {
    switch (a.RT) {
        case 1: return launch_rt<IO, 1>(a, stream);
        case 3: return launch_rt<IO, 3>(a, stream);
        case 6: return launch_rt<IO, 6>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_pet_fwd(const PetFwdArgs& a, int io_fp32, hipStream_t stream) {
    return io_fp32 ? launch_io<float>(a, stream) : launch_io<__bf16>(a, stream);
}
