// Fused PET forward (K1 encoder adapter+gate, K2 decoder value-parallel adapter, K3 LoRA delta):
//
//   out = ( s2*res + sd*( up_A( act( down_A(xa) ) ) ) )  [ (*|+) sigmoid( up_G( gelu_new( down_G(xg) ) ) ) ] * gs
//
// Reference op chains replaced: my_transformers/modeling_bart.py:1147-1155,1195-1209,1256-1257
// (K1), adapters/adapter_modeling.py:55-61 + adapter_controller.py:149-162 (K2),
// lora/controller.py:56-70 (K3, on top of the PyTorch base GEMM).
//
// Structure (pet32.h): workgroup = WAVES x 32 rows; each wave carries its 32 rows through the whole
// chain in registers on v_mfma_f32_32x32x16_bf16 (down-projection accumulators -> bias/gelu -> bf16 B
// fragments -> up-projection -> residual/gate epilogue).  Everything that comes from memory arrives by
// global_load_lds (no staging registers, whole 128-byte lines):
//   * weight fragments (shared by the waves, L2-resident): 2-slot ring, one stage ahead;
//   * each wave's own 32 x 128-byte row pieces (HBM): 3-slot ring, two stages ahead, source-swizzled so
//     the fragment reads are bank-conflict free.
// A stage ends with a counted s_waitcnt vmcnt(N) (only what the NEXT stage needs must have landed; the
// rows of the stage after next and this stage's output stores stay in flight) and a raw s_barrier.
// Outputs are staged in place in the residual tile and stored as whole lines.
// HBM traffic = read xa, read xg, re-read res (= xa for K1), write out.
#include "common.h"
#include "kernels.h"
#include "pet32.h"

template <typename IO, int RT, bool GATE, int WAVES>
struct FwdLds {
    static constexpr int NS = Geo4<IO>::NS;
    static constexpr int SEG_KB = 4 * RT;                         // one chain's weights per stage (KiB)
    static constexpr int SEG_FR = SEG_KB / NS;                    // ... in fragments
    static constexpr int W_B = SEG_KB * 1024 * (GATE ? 2 : 1);
    static constexpr int TILE_B = WAVES * 32 * 128;
    static constexpr int ROW_B = TILE_B * (GATE ? 2 : 1);
    static constexpr int NR = 3;                                  // row-ring slots
    static constexpr int ROW_OFF = 2 * W_B;
    static constexpr int BIAS_OFF = ROW_OFF + NR * ROW_B;
    static size_t bytes(int d) { return (size_t)BIAS_OFF + (size_t)2 * (32 * RT + d) * 4; }
};

// SKEEP > 0 (K1, bf16, d == 64*SKEEP): the residual is the chain input x2 itself, so each lane keeps its 64
// bytes of every x2 stage in registers while the down phase streams by (SKEEP x 16 VGPRs) and the up phase
// does not read x2 a second time (one d*M*b of fabric traffic less); the stage loops are fully unrolled.
template <typename IO, int RT, bool GATE, bool GATE_ADD, bool ACT_ID, bool DROP, int WAVES, int SKEEP>
__global__ __launch_bounds__(WAVES * 64) void pet_fwd_kernel(PetFwdArgs a) {
    using G = Geo4<IO>;
    using L = FwdLds<IO, RT, GATE, WAVES>;
    constexpr int NS = G::NS;
    constexpr int KT = 2 * RT;                   // k-steps (16) of the up projection
    constexpr int NTEN = GATE ? 2 : 1;           // row tensors in the down phase
    constexpr bool KEEP = SKEEP > 0;
    static_assert(!KEEP || NS == 1, "register-resident residual is the bf16 path");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar registers and scalar branches for everything derived from it
    const int m = lane & 31, h = lane >> 5;
    const int trow = 32 * wave + m;
    const int d = a.d;
    const int64_t row0_wave = (int64_t)blockIdx.x * (WAVES * 32) + wave * 32;
    const int S = KEEP ? SKEEP : d / G::FE;      // stages per phase (compile-time when KEEP)
    const PackGeom pg = pack_geom(RT, d, NS);
    const uint8_t* pkA = a.pk_a;
    const uint8_t* pkG = GATE ? a.pk_g : a.pk_a;
    const uint8_t* xa = reinterpret_cast<const uint8_t*>(a.xa);
    const uint8_t* xg = reinterpret_cast<const uint8_t*>(a.xg);
    const uint8_t* res = reinterpret_cast<const uint8_t*>(a.res);
    uint8_t* out = reinterpret_cast<uint8_t*>(a.out);

    auto slot_w = [&](int j) { return smem + (size_t)j * L::W_B; };
    auto slot_ta = [&](int j) { return smem + L::ROW_OFF + (size_t)j * L::ROW_B; };
    auto slot_tg = [&](int j) { return smem + L::ROW_OFF + (size_t)j * L::ROW_B + L::TILE_B; };
    float* sb = reinterpret_cast<float*>(smem + L::BIAS_OFF);
    const int nb = 32 * RT + d;

    const RowLanes rl = row_lanes<IO>(row0_wave, a.M, d, wave, lane);
    const DropSpec drop_r = DROP ? drop_resolved(a.drop) : a.drop;      // (the step counter folded into the seed: rng.h)
    const int lane16 = lane * 16;

    auto rows_count = [&](int s2) { return s2 < S ? 4 * NTEN : ((s2 < 2 * S && !KEEP) ? 4 : 0); };
    auto issue_rows = [&](int s2) {
        if (s2 >= 2 * S) return;
        if ((a.dbg & 2) && s2 >= 3) return;
        const int j = s2 % L::NR;
        const bool up = s2 >= S;
        if (up && KEEP) return;                           // residual is register resident
        const int so = (up ? s2 - S : s2) * 128;          // 128 bytes of every row per stage, both dtypes
        glds_rows4(up ? res : xa, rl, so, slot_ta(j), wave);
        if constexpr (GATE) {
            if (!up) glds_rows4(xg, rl, so, slot_tg(j), wave);
        }
    };
    auto issue_w = [&](int s1) {
        if (s1 >= 2 * S) return;
        if ((a.dbg & 1) && s1 >= 2) return;
        const bool up = s1 >= S;
        const int64_t woff = (up ? pg.pack_bytes : 0) + (int64_t)(up ? s1 - S : s1) * L::SEG_KB * 1024;
        uint8_t* dst = slot_w(s1 & 1);
        constexpr int KB = L::SEG_KB * NTEN;
        for (int k = wave; k < KB; k += WAVES) {
            const uint8_t* src = (k < L::SEG_KB ? pkA + woff + (size_t)k * 1024
                                                : pkG + woff + (size_t)(k - L::SEG_KB) * 1024) + lane16;
            glds16(src, dst + (size_t)k * 1024);
        }
    };

    auto stamp = [&](int k) {      // debug timestamps (thread 0 of each block), see api.hip VLPET_DBG & 16
        if ((a.dbg & 16) && tid == 0 && blockIdx.x < 4096) a.dbg_ts[blockIdx.x * 8 + k] = __builtin_readcyclecounter();
    };
    // packed keep flags of the lane's KU 8-element groups of down stage s (byte u <-> features s*FE + 16u + 8h ..)
    auto gen_stage = [&](int s) -> uint32_t {
        uint32_t w = 0;
        if constexpr (DROP) {
            const bool live = row0_wave + m < a.M;
            const int64_t grow = live ? row0_wave + m : a.M - 1;
#pragma unroll
            for (int u = 0; u < G::KU; ++u) {
                const int f0 = s * G::FE + 16 * u + 8 * h;
                const uint32_t kb = drop_bits8(drop_r, grow, f0, d);
                if (a.drop.keep_out != nullptr && live) drop_export8(a.drop.keep_out, grow * d + f0, kb);
                w |= kb << (8 * u);
            }
        }
        return w;
    };
    uint32_t kbw_next = 0;
    stamp(0);
    issue_w(0);
    issue_rows(0);
    issue_rows(1);
    if constexpr (DROP) kbw_next = gen_stage(0);
    {   // biases -> LDS: [bdA(32RT) | buA(d) | bdG(32RT) | buG(d)]
        const float* ba = reinterpret_cast<const float*>(a.pk_a + pg.bias_off);
        for (int i = tid; i < nb; i += WAVES * 64) sb[i] = ba[i];
        if constexpr (GATE) {
            const float* bg = reinterpret_cast<const float*>(a.pk_g + pg.bias_off);
            for (int i = tid; i < nb; i += WAVES * 64) sb[nb + i] = bg[i];
        }
    }
    __syncthreads();            // drains everything issued so far (stage 0 weights, rows of stages 0 and 1)
    stamp(1);

    // ---- down projections: pre[c], register 8*sh + j of c-tile ct <-> c = 32ct + 16sh + 8h + j
    f32x16 accA[RT];
    f32x16 accG[GATE ? RT : 1];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) accA[ct] = zero16();
    if constexpr (GATE) {
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) accG[ct] = zero16();
    }
    bf16x8 xkeep[KEEP ? SKEEP : 1][4];            // the lane's 64 bytes of every x2 stage (KEEP only)
    uint32_t kbw = 0;
    auto down_stage = [&](int s) {
        if (s == 5) stamp(5);
        issue_w(s + 1);
        issue_rows(s + 2);
        const uint8_t* w = slot_w(s & 1);
        const uint8_t* ta = slot_ta(s % L::NR);
        const uint8_t* tg = slot_tg(s % L::NR);
        if constexpr (KEEP) {
#pragma unroll
            for (int p = 0; p < 4; ++p) xkeep[s][p] = *reinterpret_cast<const bf16x8*>(tile_piece(ta, trow, 4 * h + p));
        }
        // software pipeline over the k-steps: the LDS reads of k-step u+1 are issued before the MFMAs of
        // k-step u (counted lgkmcnt waits; <= 8 fragment reads in flight), so LDS and the matrix pipe overlap.
        // (One burst of all 40 reads made hipcc emit lgkmcnt(0) before the first MFMA: reads, then MFMAs.)
        Frag<NS> bA[G::KU], bG[GATE ? G::KU : 1], wa[G::KU][RT], wg[GATE ? G::KU : 1][RT];
        if constexpr (DROP) kbw = kbw_next;
        auto load_u = [&](int u) {
            bA[u] = tile_bfrag4<IO>(ta, trow, h, u);
            if constexpr (DROP) {                        // clear the dropped elements (1 / (1 - p) is applied to the sums)
                const uint32_t kb = kbw >> (8 * u);
#pragma unroll
                for (int ns = 0; ns < NS; ++ns) {
                    u32x4 t = __builtin_bit_cast(u32x4, bA[u].p[ns]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int lo = ((int)(kb << (31 - 2 * q))) >> 31, hi = ((int)(kb << (30 - 2 * q))) >> 31;
                        t[q] &= __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x07060100u);
                    }
                    bA[u].p[ns] = __builtin_bit_cast(bf16x8, t);
                }
            }
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) wa[u][ct] = wfrag<NS>(w, u * RT + ct, lane);
            if constexpr (GATE) {
                bG[u] = tile_bfrag4<IO>(tg, trow, h, u);
#pragma unroll
                for (int ct = 0; ct < RT; ++ct) wg[u][ct] = wfrag<NS>(w, L::SEG_FR + u * RT + ct, lane);
            }
        };
        auto mfma_u = [&](int u) {
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) {
                accA[ct] = mfma_ns<NS>(wa[u][ct], bA[u], accA[ct]);
                if constexpr (GATE) accG[ct] = mfma_ns<NS>(wg[u][ct], bG[u], accG[ct]);
            }
        };
        load_u(0);
#pragma unroll
        for (int u = 1; u < G::KU; ++u) {
            load_u(u);
            __builtin_amdgcn_sched_barrier(0);
            mfma_u(u - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        mfma_u(G::KU - 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DROP) {
            // the next stage's keep flags are generated here, behind this stage's MFMAs and in front of the wait for its rows:
            // the generator depends on (seed, element index) only, so its ~100 VALU instructions per group fill the time the
            // wave would spend waiting (round 3; before, they sat between the LDS read and the MFMA of every k-step)
            if (s + 1 < S) kbw_next = gen_stage(s + 1);
            // the lane's KU mask bytes of this stage are contiguous in the packed layout (rng.h drop_pos): one store
            if (a.drop.bits_out != nullptr && row0_wave + m < a.M) {
                uint8_t* bp = a.drop.bits_out + (row0_wave + m) * (int64_t)(d >> 3) + drop_pos((s * G::FE + 8 * h) >> 3);
                if constexpr (G::KU == 4) *reinterpret_cast<uint32_t*>(bp) = kbw;
                else *reinterpret_cast<uint16_t*>(bp) = (uint16_t)kbw;
            }
        }
        if (s == 5) stamp(6);
        // next stage needs: its weights (issued first in this stage) and its rows (issued a stage earlier)
        if (a.dbg & 3) wait_vm(0); else wait_vm(rows_count(s + 2));
        __builtin_amdgcn_s_barrier();
        if (s == 5) stamp(7);
    };
    int s = 0;
    if constexpr (KEEP) {
#pragma unroll
        for (int ss = 0; ss < SKEEP; ++ss) down_stage(ss);
        s = SKEEP;
    } else {
        for (; s < S; ++s) down_stage(s);
    }
    if constexpr (DROP) {                        // dropout's 1 / (1 - p): once on the sums instead of on every kept element
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) accA[ct] *= a.drop.keep_scale;
    }

    stamp(2);
    // ---- bias + activation -> B fragments of the up projection (k-step 2ct+sh holds c = 32ct+16sh+8h+j)
    Frag<NS> zA[KT];
    Frag<NS> zG[GATE ? KT : 1];
    {
        const float* bdA = sb + 8 * h;
        const float* bdG = sb + nb + 8 * h;
        // training form (kernels.h PetFwdArgs::save): z and gelu'(pre) of the adapter chain for the backward
        const bool save = !GATE && a.save != nullptr && row0_wave + m < a.M;      // (K3: z only)
        IO* sv_z = reinterpret_cast<IO*>(a.save) + (row0_wave + m) * (int64_t)(32 * RT) + 8 * h;
        IO* sv_g = reinterpret_cast<IO*>(reinterpret_cast<uint8_t*>(a.save) + a.save_stride) + (row0_wave + m) * (int64_t)(32 * RT) + 8 * h;
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) {
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = accA[ct][8 * sh + j] + bdA[32 * ct + 16 * sh + j];
                if (save && !ACT_ID) {
                    float g[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] = gelu_new_grad_f(v[j]);
                    store8_f32(sv_g + 32 * ct + 16 * sh, g);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ACT_ID ? v[j] : gelu_new_f(v[j]);
                if (save) store8_f32(sv_z + 32 * ct + 16 * sh, v);
                zA[2 * ct + sh] = frag_from_f32<NS>(v);
                if constexpr (GATE) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = gelu_new_f(accG[ct][8 * sh + j] + bdG[32 * ct + 16 * sh + j]);
                    zG[2 * ct + sh] = frag_from_f32<NS>(v);
                }
            }
        }
    }

    stamp(3);
    // ---- up projections + epilogue: FE features per stage, LW contiguous per lane
    const float* buA = sb + 32 * RT + G::LW * h;
    const float* buG = sb + nb + 32 * RT + G::LW * h;
    const float gs = GATE ? a.gs : 1.0f;
    const float s2g = a.s2 * gs, sdg = a.sd * gs;      // gate scale folded into the linear part
    auto up_stage = [&](int s, int su) {
        issue_w(s + 1);
        issue_rows(s + 2);
        const uint8_t* w = slot_w(s & 1);
        uint8_t* tr = slot_ta(s % L::NR);
        float o[G::LW];
        // accumulators start at the up-projection bias (the MFMA C operand), so the epilogue has no bias add
        f32x16 aA[G::NV], aG[G::NV];
#pragma unroll
        for (int v = 0; v < G::NV; ++v) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(buA + su * G::FE + 16 * v + 4 * q);
                aA[v][4 * q] = t[0]; aA[v][4 * q + 1] = t[1]; aA[v][4 * q + 2] = t[2]; aA[v][4 * q + 3] = t[3];
                if constexpr (GATE) {
                    const f32x4 tg = *reinterpret_cast<const f32x4*>(buG + su * G::FE + 16 * v + 4 * q);
                    aG[v][4 * q] = tg[0]; aG[v][4 * q + 1] = tg[1]; aG[v][4 * q + 2] = tg[2]; aG[v][4 * q + 3] = tg[3];
                }
            }
        }
        float r[G::LW];
        if constexpr (KEEP) {
#pragma unroll
            for (int i = 0; i < G::LW; ++i) r[i] = (float)xkeep[su][i >> 3][i & 7];
        } else {
            tile_lane_vals4<IO>(tr, trow, h, r);
        }
        // per 32-feature n-tile v: fragment reads of tile v+1 are issued before the MFMAs of tile v, and the
        // epilogue of tile v sits in the same scheduling region as the MFMAs of tile v+1 (VALU in the MFMA shadow)
        Frag<NS> wa[G::NV][KT], wg[GATE ? G::NV : 1][KT];
        auto load_v = [&](int v) {
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
                wa[v][ks] = wfrag<NS>(w, v * KT + ks, lane);
                if constexpr (GATE) wg[v][ks] = wfrag<NS>(w, L::SEG_FR + v * KT + ks, lane);
            }
        };
        auto mfma_v = [&](int v) {
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
                aA[v] = mfma_ns<NS>(wa[v][ks], zA[ks], aA[v]);
                if constexpr (GATE) aG[v] = mfma_ns<NS>(wg[v][ks], zG[ks], aG[v]);
            }
        };
        auto epi_v = [&](int v) {
#pragma unroll
            for (int i = 16 * v; i < 16 * v + 16; ++i) {
                float hv = s2g * r[i] + sdg * aA[v][i & 15];
                if constexpr (GATE) {
                    const float gt = sigmoid_f(aG[v][i & 15]);
                    hv = GATE_ADD ? hv + gs * gt : hv * gt;
                }
                o[i] = hv;
            }
        };
        load_v(0);
#pragma unroll
        for (int v = 1; v < G::NV; ++v) {
            load_v(v);
            __builtin_amdgcn_sched_barrier(0);
            mfma_v(v - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        mfma_v(G::NV - 1);
#pragma unroll
        for (int v = 0; v < G::NV; ++v) epi_v(v);
        // stage the outputs in place (the wave's own rows of the residual tile), then whole-line stores
        stage_lane_vals4<IO>(tr, trow, h, o);
        store_rows4(out, rl, su * 128, tr, wave, lane);
        if (a.dbg & 3) wait_vm(0); else wait_vm(rows_count(s + 2) + rl.n_inst);
        __builtin_amdgcn_s_barrier();
    };
    if constexpr (KEEP) {
#pragma unroll
        for (int su = 0; su < SKEEP; ++su) up_stage(SKEEP + su, su);
    } else {
        for (; s < 2 * S; ++s) up_stage(s, s - S);
    }
    stamp(4);
}

template <typename IO, int RT, bool GATE, bool GATE_ADD, bool ACT_ID, bool DROP, int WAVES, int SKEEP = 0>
static hipError_t launch_one(const PetFwdArgs& a, hipStream_t stream) {
    using L = FwdLds<IO, RT, GATE, WAVES>;
    const size_t lds = L::bytes(a.d);
    auto kern = pet_fwd_kernel<IO, RT, GATE, GATE_ADD, ACT_ID, DROP, WAVES, SKEEP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int rows = WAVES * 32;
    const int blocks = (int)((a.M + rows - 1) / rows);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

template <typename IO, int RT, bool GATE, bool GATE_ADD, bool ACT_ID, bool DROP>
static hipError_t launch_waves(const PetFwdArgs& a, hipStream_t stream) {
    // 4 waves (128 rows) per workgroup unless the rings would not fit the 160 KiB LDS
    if constexpr (FwdLds<IO, RT, GATE, 4>::BIAS_OFF + 8 * 1024 <= 160 * 1024) {
        if constexpr (GATE && IoTraits<IO>::NS == 1 && RT <= 3) {
            // K1, bf16, d = 768: residual == chain input -> keep it in registers (12 stages x 16 VGPRs)
            if (a.res == a.xa && a.d == 768 && !(a.dbg & 32))
                return launch_one<IO, RT, GATE, GATE_ADD, ACT_ID, DROP, 4, 12>(a, stream);
        }
        return launch_one<IO, RT, GATE, GATE_ADD, ACT_ID, DROP, 4>(a, stream);
    }
    else
        return launch_one<IO, RT, GATE, GATE_ADD, ACT_ID, DROP, 2>(a, stream);
}

template <typename IO, int RT>
static hipError_t launch_rt(const PetFwdArgs& a, hipStream_t stream) {
    const bool gate = a.flags & PET_GATE, add = a.flags & PET_GATE_ADD, act_id = a.flags & PET_ACT_IDENTITY;
    const bool drop = drop_active(a.drop);
    if (gate) return add ? launch_waves<IO, RT, true, true, false, false>(a, stream)
                         : launch_waves<IO, RT, true, false, false, false>(a, stream);
    if (act_id) return drop ? launch_waves<IO, RT, false, false, true, true>(a, stream)
                            : launch_waves<IO, RT, false, false, true, false>(a, stream);
    return launch_waves<IO, RT, false, false, false, false>(a, stream);
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
