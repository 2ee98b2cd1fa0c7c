// C ABI of libvlpet_hip.so (declared in include/vlpet_hip.h): argument checking + launch plumbing.
#include <atomic>
#include "../../include/vlpet_hip.h"
#include "common.h"
#include "kernels.h"
#include <cmath>
#include <stdlib.h>
#include <stdio.h>
#include <vector>

#define VLPET_VERSION 600      // 600 (round 6): in-launch reduce-scatter of the K1 backward (cols_reduce.h; vlpet_set_in_launch_reduce, vlpet_adapter_gate_bwd_finalize_launch, phases bit 5), K4 give-up repair inside the call (vlpet_visproj_gemm_exchange_bytes; larger workspace), vlpet_test_hold_cus; 500 (round 5): vlpet_visproj_fwd_gemm (K4 as a tiled GEMM + exchanged statistics), vlpet_sublayer_tail_rms_fwd / vlpet_rmsnorm_tail_bwd, vlpet_adapter_gate_bwd_saved_y (backward from the forward's output), vlpet_finalize_defer / _flush; 420: vlpet_lora_delta_fwd_r8 (K3 at rank <= 8 as a streaming kernel, lora8.hip); 410: vlpet_set_seed_counter (dropout seeds under graph replay), two-pass K2 / K3 forward; 400: two-pass K1 forward (pet_fwd2p.hip), vlpet_sublayer_tail_bwd_out;
#define VLPET_VERSION_R3 300      // 300: column-parallel K1 backward pass (pet_cols.hip), phases bits 3 / 4, vlpet_adapter_gate_bwd_form;
#define VLPET_VERSION_R2 221      // 221: vlpet_sublayer_tail_reduce, vlpet_layernorm_bwd_xhat, vlpet_rmsnorm_{fwd,bwd}, vlpet_colsum;  round 2: LoRA dropout generator ABI, sliced AdamW, K3 training form; 210: strided attention entry points, streaming weight gradients; 220: low-rank visual projector

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }
static inline int herr(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

extern "C" int vlpet_version(void) { return VLPET_VERSION; }
extern "C" int vlpet_debug_build(void) { return VLPET_IS_DEBUG_BUILD; }

// Test instrument (tests/test_gpu_k4.py, tests/test_gpu_cols.py: the in-launch hand-offs under a GPU that is NOT the launch's alone):
// `workgroups` workgroups of 64 threads, each holding `lds_bytes` of LDS (so one sits on a CU and a workgroup needing more than
// 160 KiB - lds_bytes does not fit beside it), poll *release until it is nonzero or `max_ms` milliseconds have passed, whichever is
// first -- the bound (capped at 10 s) makes a forgotten release harmless.
__global__ __launch_bounds__(64) void hold_cus_kernel(unsigned* release, unsigned long long ticks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t hold_smem[];
    typedef __attribute__((address_space(1))) unsigned gu32;
    if (threadIdx.x == 0) {
        hold_smem[0] = 1;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load((gu32*)release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && wall_clock64() - t0 < ticks)
            __builtin_amdgcn_s_sleep(32);
    }
}
extern "C" int vlpet_test_hold_cus(int workgroups, int lds_bytes, void* release_flag, int max_ms, vlpet_stream_t stream) {
    if (!release_flag) return VLPET_E_NULL;
    if (workgroups <= 0 || workgroups > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || max_ms <= 0) return VLPET_E_SHAPE;
    if (max_ms > 10000) max_ms = 10000;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hold_cus_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return herr(e);
    hipLaunchKernelGGL(hold_cus_kernel, dim3((unsigned)workgroups), dim3(64), (size_t)lds_bytes, (hipStream_t)stream,
                       reinterpret_cast<unsigned*>(release_flag), (unsigned long long)max_ms * 100000ull);     // wall_clock64: 100 MHz
    return herr(hipGetLastError());
}

extern "C" const char* vlpet_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case VLPET_E_SHAPE: return "bad shape (need M > 0, d % 64 == 0, r > 0)";
        case VLPET_E_RANK: return "unsupported bottleneck rank (max 192) or tile count";
        case VLPET_E_ALIGN: return "pointer not 16-byte aligned";
        case VLPET_E_WORKSPACE: return "workspace too small";
        case VLPET_E_NULL: return "required pointer is NULL";
        case VLPET_E_DTYPE: return "unsupported dtype";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

extern "C" int vlpet_rank_tiles(int r) {
    if (r <= 0) return VLPET_E_SHAPE;
    if (r <= 32) return 1;
    if (r <= 96) return 3;
    if (r <= 192) return 6;
    return VLPET_E_RANK;
}

static inline bool tiles_ok(int t) { return t == 1 || t == 3 || t == 6; }
static inline bool dtype_ok(int t) { return t == VLPET_F32 || t == VLPET_BF16; }

extern "C" size_t vlpet_packed_bytes(int tiles, int d, int io_dtype) {
    return (size_t)pack_geom(tiles, d, io_dtype == VLPET_F32 ? 2 : 1).total_bytes;
}

extern "C" int vlpet_pack_pair(const void* const* wd_heads, const void* const* bd_heads, int n_heads,
                               const void* wu, const void* bu, int r, int d, int tiles,
                               int param_dtype, int io_dtype, void* packed, vlpet_stream_t stream) {
    if (!wd_heads || !wu || !packed) return VLPET_E_NULL;
    if (r <= 0 || d <= 0 || d % 64 != 0 || n_heads <= 0 || n_heads > VLPET_MAX_HEADS || r % n_heads != 0)
        return VLPET_E_SHAPE;
    if (!tiles_ok(tiles) || r > 32 * tiles) return VLPET_E_RANK;
    if (!dtype_ok(param_dtype) || !dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    if (!aligned16(packed)) return VLPET_E_ALIGN;
    PackArgs a;
    for (int i = 0; i < VLPET_MAX_HEADS; ++i) {
        a.wd[i] = i < n_heads ? wd_heads[i] : nullptr;
        a.bd[i] = (bd_heads && i < n_heads) ? bd_heads[i] : nullptr;
        if (i < n_heads && a.wd[i] == nullptr) return VLPET_E_NULL;
    }
    a.wu = wu; a.bu = bu;
    a.n_heads = n_heads; a.rows_per_head = r / n_heads;
    a.r = r; a.d = d; a.RT = tiles; a.src_bf16 = param_dtype == VLPET_BF16;
    a.n_packs = 4;
    a.out = reinterpret_cast<uint8_t*>(packed);
    return herr(launch_pack_pair(a, io_dtype == VLPET_F32 ? 2 : 1, (hipStream_t)stream));
}

extern "C" int vlpet_pack_pairs(int n, const void* const* wd_heads_flat, const void* const* bd_heads_flat, int n_heads,
                                const void* const* wu, const void* const* bu, int r, int d, int tiles,
                                int param_dtype, int io_dtype, void* const* packed, vlpet_stream_t stream) {
    if (n <= 0) return 0;
    if (!wd_heads_flat || !wu || !packed) return VLPET_E_NULL;
    if (r <= 0 || d <= 0 || d % 64 != 0 || n_heads <= 0 || n_heads > VLPET_MAX_HEADS || r % n_heads != 0) return VLPET_E_SHAPE;
    if (!tiles_ok(tiles) || r > 32 * tiles) return VLPET_E_RANK;
    if (!dtype_ok(param_dtype) || !dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    for (int i0 = 0; i0 < n; i0 += VLPET_PACK_BATCH) {
        PackBatch b;
        b.n = n - i0 < VLPET_PACK_BATCH ? n - i0 : VLPET_PACK_BATCH;
        for (int k = 0; k < b.n; ++k) {
            PackArgs& a = b.p[k];
            const int idx = i0 + k;
            for (int i = 0; i < VLPET_MAX_HEADS; ++i) {
                a.wd[i] = i < n_heads ? wd_heads_flat[(size_t)idx * n_heads + i] : nullptr;
                a.bd[i] = (bd_heads_flat && i < n_heads) ? bd_heads_flat[(size_t)idx * n_heads + i] : nullptr;
                if (i < n_heads && a.wd[i] == nullptr) return VLPET_E_NULL;
            }
            a.wu = wu[idx]; a.bu = bu ? bu[idx] : nullptr;
            if (!a.wu || !packed[idx]) return VLPET_E_NULL;
            if (!aligned16(packed[idx])) return VLPET_E_ALIGN;
            a.n_heads = n_heads; a.rows_per_head = r / n_heads;
            a.r = r; a.d = d; a.RT = tiles; a.src_bf16 = param_dtype == VLPET_BF16;
            a.n_packs = 4;
            a.out = reinterpret_cast<uint8_t*>(packed[idx]);
        }
        hipError_t e = launch_pack_pairs(b, io_dtype == VLPET_F32 ? 2 : 1, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

static int check_common(int64_t M, int d, int tiles, int io_dtype) {
    if (M <= 0 || d <= 0 || d % 64 != 0) return VLPET_E_SHAPE;
    if (!tiles_ok(tiles)) return VLPET_E_RANK;
    if (!dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    return 0;
}

static int gate_flags(int gate_mode, int* flags);

// the forward's saved bottleneck activations: four [M, 32*tiles] IO-dtype tensors (z_a, gelu'_a, z_g, gelu'_g)
static size_t saved_stride(int64_t M, int tiles, int io_dtype) {
    return align256((size_t)M * 32 * tiles * (io_dtype == VLPET_F32 ? 4 : 2));
}
extern "C" size_t vlpet_saved_bytes(int64_t M, int tiles, int io_dtype) {
    if (M <= 0 || !tiles_ok(tiles)) return 0;
    return 4 * saved_stride(M, tiles, io_dtype);
}

static const DropSpec NO_DROP = {nullptr, nullptr, nullptr, nullptr, 0, 0, 1.f, nullptr};

// Device step counter mixed into every dropout seed (rng.h vlpet_eff_seed): process-wide, set by the trainer that replays captured
// steps (train.Trainer(graph=True)); nullptr = seeds are used as passed.  The only state the library keeps between calls.
static std::atomic<const uint64_t*> g_seed_ctr{nullptr};
extern "C" int vlpet_set_seed_counter(const uint64_t* device_counter) {
    g_seed_ctr.store(device_counter);
    return 0;
}

// Round 6: the column-parallel backward passes (gated K1 at r <= 96, K2 / K3) sum their row-chunk partials inside the launch
// (cols_reduce.h) -- unless the caller has said that OTHER kernels may run beside the backward (gradient collectives on their own
// stream, a second process on the device): a workgroup that waits for partners which cannot start holds its CU for as long as the
// foreign kernel lasts, where the two-launch form simply runs in two rounds.  Process-wide, like the seed counter; default on.
static std::atomic<int> g_in_launch_reduce{1};
extern "C" int vlpet_set_in_launch_reduce(int on) { return g_in_launch_reduce.exchange(on != 0 ? 1 : 0); }
// ... and only where it was measured ahead (profiles/r06_in_launch_reduce_ab.txt, ABBA on one box): the configs[1] step at the full
// batch (15,272-46,648 rows per call) 27.92 / 27.94 k samples/s with it against 27.73 / 27.45 k with the finalize launch, the K1
// backward op 137.5 vs 138.9 us; at the per-rank sizes of an 8-GPU run (1,900-5,800 rows, replayed graph) 5.45 / 5.47 ms per step with it
// against 5.43 / 5.43 without -- so small launches keep the finalize launch
static bool k1_in_launch_reduce(int64_t M) {
    return g_in_launch_reduce.load() != 0 && vlpet_tuning().cols_red != 0 && (vlpet_tuning().cols_red == 2 || M >= 8192);
}

// p in [0, 1): explicit mask (keep_mask != NULL) or the in-kernel generator keyed by `seed`; p == 0: no dropout
static int make_drop(const uint8_t* keep_mask, float p, uint64_t seed, uint8_t* keep_out, DropSpec* ds) {
    if (!(p >= 0.f && p < 1.f)) return VLPET_E_SHAPE;
    *ds = NO_DROP;
    if (p == 0.f) return 0;
    ds->keep = keep_mask;
    ds->keep_out = keep_out;
    ds->seed = seed;
    ds->seed_ctr = g_seed_ctr.load();
    double t = (double)p * 65536.0 + 0.5;
    if (t > 65535.0) t = 65535.0;
    ds->thr = (uint32_t)t;
    if (ds->thr == 0 && !keep_mask) { *ds = NO_DROP; return 0; }      // p < 2^-17: the generator never drops
    ds->keep_scale = 1.0f / (1.0f - p);
    return 0;
}

static int run_fwd(const void* xa, const void* res, const void* xg, const void* pk_a, const void* pk_g,
                   const DropSpec& drop, void* out, int64_t M, int d, int tiles,
                   float s2, float sd, float gs, int flags, int io_dtype, vlpet_stream_t stream,
                   void* saved = nullptr) {
    int rc = check_common(M, d, tiles, io_dtype);
    if (rc) return rc;
    if (!xa || !res || !pk_a || !out) return VLPET_E_NULL;
    if (saved && !aligned16(saved)) return VLPET_E_ALIGN;
    if ((flags & PET_GATE) && (!xg || !pk_g)) return VLPET_E_NULL;
    if (!aligned16(xa) || !aligned16(res) || !aligned16(out) || !aligned16(pk_a) ||
        ((flags & PET_GATE) && (!aligned16(xg) || !aligned16(pk_g))) || (drop.keep && !aligned16(drop.keep)) ||
        (drop.keep_out && !aligned16(drop.keep_out)))
        return VLPET_E_ALIGN;
    PetFwdArgs a;
    a.xa = xa; a.res = res; a.xg = xg; a.out = out;
    a.pk_a = reinterpret_cast<const uint8_t*>(pk_a);
    a.pk_g = reinterpret_cast<const uint8_t*>(pk_g);
    a.drop = drop;
    // K3 training form: the packed dropout mask goes into the saved block, right after z
    if (saved && (flags & PET_ACT_IDENTITY) && drop_active(drop))
        a.drop.bits_out = reinterpret_cast<uint8_t*>(saved) + saved_stride(M, tiles, io_dtype);
    a.M = M; a.d = d; a.RT = tiles;
    a.s2 = s2; a.sd = sd; a.gs = gs; a.flags = flags;
    a.save = saved; a.save_stride = (int64_t)saved_stride(M, tiles, io_dtype);
    a.dbg = 0;
    a.dbg_ts = nullptr;
    a.d_in = 0; a.pk_a_dn = nullptr; a.pk_g_dn = nullptr; a.gm = 1.f; a.go = 0.f;
#ifdef VLPET_DEBUG      // ablation bits / cycle stamps: debug builds only (function-static device buffer, synchronises, prints)
    a.dbg = vlpet_tuning().dbg;
    if (a.dbg & 16) {     // debug only: per-phase timestamps of wave 0 of every block, printed at the next call
        static unsigned long long* dev = nullptr;
        static int nblk = 0;
        if (dev) {
            (void)hipDeviceSynchronize();
            std::vector<unsigned long long> h((size_t)nblk * 8);
            (void)hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost);
            double acc[8] = {0}; int n = 0;
            for (int b = 0; b < nblk; b += 7) {
                const unsigned long long* t = &h[(size_t)b * 8];
                acc[0] += (double)(t[1] - t[0]); acc[1] += (double)(t[2] - t[1]); acc[2] += (double)(t[3] - t[2]);
                acc[3] += (double)(t[4] - t[3]); acc[4] += (double)(t[6] - t[5]); acc[5] += (double)(t[7] - t[6]);
                ++n;
            }
            fprintf(stderr, "[vlpet ts] cycles: prologue=%.0f down=%.0f act=%.0f up=%.0f | down-stage 5: issue+reads+mfma=%.0f wait+barrier=%.0f (n=%d)\n",
                    acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, n);
        } else {
            (void)hipMalloc(&dev, 4096 * 8 * 8);
        }
        nblk = (int)((M + 127) / 128); if (nblk > 4096) nblk = 4096;
        a.dbg_ts = dev;
    }
#endif
    // training form, bf16: two passes with the weights resident in registers where that form is the faster one
    if (const int f2 = vlpet_tuning().fwd2p; f2 != 0 && k1_fwd2p_applies(a, io_dtype == VLPET_F32) && (f2 > 0 || k1_fwd2p_preferred(a)))
        return herr(launch_k1_fwd2p(a, f2 == 2 ? 1 : f2 == 3 ? 2 : 3, (hipStream_t)stream));
    if ((flags & PET_GATE) && !(a.dbg & 64))      // two-chain gate forward: one wave per chain (VLPET_DBG=64: single-wave form)
        return herr(launch_pet_gate_fwd(a, io_dtype == VLPET_F32, (hipStream_t)stream));
    return herr(launch_pet_fwd(a, io_dtype == VLPET_F32, (hipStream_t)stream));
}

extern "C" int vlpet_adapter_gate_fwd(const void* x1, const void* x2, const void* packed_a,
                                      const void* packed_g, void* out, int64_t M, int d, int tiles,
                                      int gate_mode, float delta_scale, float x2_scale, float gate_scale,
                                      int io_dtype, vlpet_stream_t stream) {
    int flags = 0;
    if (gate_mode == VLPET_GATE_MUL) flags = PET_GATE;
    else if (gate_mode == VLPET_GATE_ADD) flags = PET_GATE | PET_GATE_ADD;
    else if (gate_mode != VLPET_GATE_NONE) return VLPET_E_SHAPE;
    return run_fwd(x2, x2, x1, packed_a, packed_g, NO_DROP, out, M, d, tiles, x2_scale, delta_scale,
                   flags ? gate_scale : 1.f, flags, io_dtype, stream);
}

extern "C" int vlpet_adapter_gate_fwd_save(const void* x1, const void* x2, const void* packed_a,
                                           const void* packed_g, void* out, void* saved, int64_t M, int d, int tiles,
                                           int gate_mode, float delta_scale, float x2_scale, float gate_scale,
                                           int io_dtype, vlpet_stream_t stream) {
    int flags;
    if (gate_flags(gate_mode, &flags)) return VLPET_E_SHAPE;
    if (!saved) return VLPET_E_NULL;
    return run_fwd(x2, x2, x1, packed_a, packed_g, NO_DROP, out, M, d, tiles, x2_scale, delta_scale,
                   flags ? gate_scale : 1.f, flags, io_dtype, stream, saved);
}

extern "C" int vlpet_parallel_adapter_fwd(const void* x, const void* y, const void* packed, void* out,
                                          int64_t M, int d, int tiles, float scale, int io_dtype,
                                          vlpet_stream_t stream) {
    return run_fwd(x, y, nullptr, packed, nullptr, NO_DROP, out, M, d, tiles, 1.f, scale, 1.f, 0, io_dtype, stream);
}

extern "C" int vlpet_parallel_adapter_fwd_save(const void* x, const void* y, const void* packed, void* out, void* saved,
                                               int64_t M, int d, int tiles, float scale, int io_dtype,
                                               vlpet_stream_t stream) {
    if (!saved) return VLPET_E_NULL;
    return run_fwd(x, y, nullptr, packed, nullptr, NO_DROP, out, M, d, tiles, 1.f, scale, 1.f, 0, io_dtype, stream, saved);
}

extern "C" int vlpet_lora_delta_fwd(const void* x, const void* base, const void* packed,
                                    const uint8_t* keep_mask, float p, uint64_t seed, uint8_t* keep_out, void* out,
                                    int64_t M, int d, int tiles, float scaling, int io_dtype, vlpet_stream_t stream) {
    DropSpec ds;
    if (int rc = make_drop(keep_mask, p, seed, keep_out, &ds)) return rc;
    return run_fwd(x, base, nullptr, packed, nullptr, ds, out, M, d, tiles, 1.f, scaling, 1.f,
                   PET_ACT_IDENTITY, io_dtype, stream);
}

extern "C" size_t vlpet_lora_saved_bytes(int64_t M, int d, int tiles, int io_dtype) {
    if (M <= 0 || d <= 0 || !tiles_ok(tiles)) return 0;
    return saved_stride(M, tiles, io_dtype) + align256((size_t)M * (d / 8));      // z, then 1 bit per element of the mask
}

extern "C" int vlpet_lora_delta_fwd_save(const void* x, const void* base, const void* packed,
                                         const uint8_t* keep_mask, float p, uint64_t seed, uint8_t* keep_out, void* out,
                                         void* saved, int64_t M, int d, int tiles, float scaling, int io_dtype,
                                         vlpet_stream_t stream) {
    if (!saved) return VLPET_E_NULL;
    DropSpec ds;
    if (int rc = make_drop(keep_mask, p, seed, keep_out, &ds)) return rc;
    return run_fwd(x, base, nullptr, packed, nullptr, ds, out, M, d, tiles, 1.f, scaling, 1.f,
                   PET_ACT_IDENTITY, io_dtype, stream, saved);
}

// K3 at rank <= 8: the streaming form (lora8.hip).  `saved` NULL = inference form; else the training form's saved block
// (vlpet_lora_saved_bytes(M, d, 1, io): z, then the packed mask) exactly as vlpet_lora_delta_fwd_save leaves it.
extern "C" int vlpet_lora_delta_fwd_r8(const void* x, const void* base, const void* packed, const uint8_t* keep_mask, float p,
                                       uint64_t seed, uint8_t* keep_out, void* out, void* saved, int64_t M, int d, int r,
                                       float scaling, int io_dtype, vlpet_stream_t stream) {
    int rc = check_common(M, d, 1, io_dtype);
    if (rc) return rc;
    if (!x || !base || !packed || !out) return VLPET_E_NULL;
    if (!lora8_applies(M, d, r, io_dtype == VLPET_F32)) return VLPET_E_SHAPE;
    if (!aligned16(x) || !aligned16(base) || !aligned16(out) || !aligned16(packed) || (saved && !aligned16(saved)) ||
        (keep_mask && !aligned16(keep_mask)) || (keep_out && !aligned16(keep_out)))
        return VLPET_E_ALIGN;
    Lora8Args a{};
    if (int rd = make_drop(keep_mask, p, seed, keep_out, &a.drop)) return rd;
    if (saved && drop_active(a.drop)) a.drop.bits_out = reinterpret_cast<uint8_t*>(saved) + saved_stride(M, 1, io_dtype);
    a.x = x; a.base = base; a.out = out; a.pk = reinterpret_cast<const uint8_t*>(packed);
    a.save = saved; a.M = M; a.d = d; a.scaling = scaling;
    return herr(launch_lora8_fwd(a, (hipStream_t)stream));
}
extern "C" int vlpet_lora_r8_applies(int64_t M, int d, int r, int io_dtype) {
    return lora8_applies(M, d, r, io_dtype == VLPET_F32) ? 1 : 0;
}

// ------------------------------------------------------------------ backward workspace
struct BwdWs {
    size_t z_a, dp_a, z_g, dp_g, dh, dq, partial, red_ctrl, total;     // red_ctrl: control words of the in-launch reduce-scatter (cols_reduce.h)
    int row_chunks;
    int64_t rows_per_chunk;
};
static BwdWs bwd_ws(int64_t M, int d, int tiles, bool gate, int io_dtype) {
    BwdWs w{};
    const size_t esz = io_dtype == VLPET_F32 ? 4 : 2;
    const size_t side = align256((size_t)M * 32 * tiles * esz);
    size_t wide = align256((size_t)M * d * esz);
    // the feature-split pass 1 parks its fp32 partial dz in the dh / dq area (unused by the two-pass form): four blocks at six tiles
    // need twice the room of the two wide tensors
    if (gate && tiles == 6 && io_dtype != VLPET_F32 && k1_dz6_feature_blocks(M, d) > 1) {
        const size_t need = align256((size_t)k1_dz6_feature_blocks(M, d) * (size_t)M * 32 * tiles * 4);
        if (need > wide) wide = need;
    }
    size_t o = 0;
    w.z_a = o; o += side;
    w.dp_a = o; o += side;
    if (gate) {
        w.z_g = o; o += side;
        w.dp_g = o; o += side;
        w.dh = o; o += wide;
        w.dq = o; o += wide;
    }
    const int njobs = gate ? 4 : 2;
    wgrad_plan(M, njobs, d, &w.row_chunks, &w.rows_per_chunk);
    int chunks = w.row_chunks;
    if (!gate && d % 128 == 0) {                        // the two-pass form without a gate (pet_cols_ng.hip) cuts the rows its own way
        int rcn; int64_t rpcn;
        ng_cols_plan(M, d, &rcn, &rpcn);
        if (rcn > chunks) chunks = rcn;
    }
    if (gate) {     // the two-pass gated backward (pet_gate_bwd3.hip) cuts the rows differently: room for either plan
        int rc3, gs3, ng3; int64_t rpc3;
        gate_bwd3_plan(M, d, io_dtype == VLPET_F32, &rc3, &rpc3, &gs3, &ng3);
        if (rc3 > chunks) chunks = rc3;
        int rc4; int64_t rpc4;                          // ... and the column-parallel pass of pet_cols.hip
        k1_cols_plan(M, d, &rc4, &rpc4);
        if (rc4 > chunks) chunks = rc4;
        k1_cols6_plan(M, d, &rc4, &rpc4);
        if (rc4 > chunks) chunks = rc4;
    }
    w.partial = o;
    o += align256(wgrad_workspace_bytes(njobs, tiles, d, chunks));
    w.red_ctrl = o;
    o += align256((size_t)(d >= 128 ? d / 128 : 1) * COLS_RED_STRIDE * 4);
    w.total = o;
    return w;
}

extern "C" size_t vlpet_bwd_workspace_bytes(int64_t M, int d, int tiles, int has_gate, int io_dtype) {
    if (M <= 0 || d <= 0 || !tiles_ok(tiles)) return 0;
    return bwd_ws(M, d, tiles, has_gate != 0, io_dtype).total;
}

static int run_bwd(const void* dy, const void* xa, const void* res, const void* xg,
                   const void* pk_a, const void* pk_g, const DropSpec& drop,
                   void* dxa, void* dxg,
                   float* dwd, float* dbd, float* dwu, float* dbu,
                   float* dwgd, float* dbgd, float* dwgu, float* dbgu, int r, int rg,
                   void* workspace, size_t workspace_bytes, int64_t M, int d, int tiles,
                   float s2, float sd, float gs, int flags, int io_dtype, vlpet_stream_t stream,
                   int phases = 3 /* bit0: row-parallel kernel, bit1: weight gradients; bit3: skip the finalize of bit1, bit4: finalize only */,
                   const void* saved = nullptr /* vlpet_adapter_gate_fwd_save's block */,
                   const void* dx1_in = nullptr /* gated K1: added to dxg (must not alias it) */,
                   const void* yout = nullptr /* gated K1: the forward's output (PetBwdArgs::y) */) {
    int rc = check_common(M, d, tiles, io_dtype);
    if (rc) return rc;
    const bool gate = flags & PET_GATE;
    if (!dy || !xa || !pk_a || !dxa || !dwd || !dwu || !workspace) return VLPET_E_NULL;
    if (gate && (!res || !xg || !pk_g || !dxg || !dwgd || !dwgu || !dbgd || !dbgu)) return VLPET_E_NULL;
    if (r <= 0 || r > 32 * tiles || (gate && (rg <= 0 || rg > 32 * tiles))) return VLPET_E_RANK;
    if (!aligned16(dy) || !aligned16(xa) || !aligned16(dxa) || !aligned16(workspace) || !aligned16(pk_a) ||
        (gate && (!aligned16(xg) || !aligned16(res) || !aligned16(dxg) || !aligned16(pk_g))) ||
        (drop.keep && !aligned16(drop.keep)))
        return VLPET_E_ALIGN;
    const BwdWs w = bwd_ws(M, d, tiles, gate, io_dtype);
    if (workspace_bytes < w.total) return VLPET_E_WORKSPACE;
    uint8_t* ws = reinterpret_cast<uint8_t*>(workspace);

    PetBwdArgs b;
    b.dy = dy; b.xa = xa; b.res = res; b.xg = xg;
    b.dxa = dxa; b.dxg = dxg; b.dxg_in = nullptr;
    if (dx1_in && (!gate || !aligned16(dx1_in) || dx1_in == dxg)) return VLPET_E_ALIGN;
    b.z_a = ws + w.z_a; b.dp_a = ws + w.dp_a;
    b.z_g = gate ? ws + w.z_g : nullptr; b.dp_g = gate ? ws + w.dp_g : nullptr;
    b.dh = gate ? ws + w.dh : nullptr; b.dq = gate ? ws + w.dq : nullptr;
    b.pk_a = reinterpret_cast<const uint8_t*>(pk_a);
    b.pk_g = reinterpret_cast<const uint8_t*>(pk_g);
    b.drop = drop; b.drop.keep_out = nullptr; b.drop.bits_out = nullptr;
    if (saved && (flags & PET_ACT_IDENTITY) && drop_active(drop)) {        // the forward's packed mask instead of regenerating it
        b.drop.bits = reinterpret_cast<const uint8_t*>(saved) + saved_stride(M, tiles, io_dtype);
        b.drop.keep = nullptr;
    }
    b.M = M; b.d = d; b.RT = tiles;
    b.s2 = s2; b.sd = sd; b.gs = gs; b.flags = flags;
    b.gm = 1.f; b.go = 0.f; b.fsplit = 0; b.dz_part = nullptr;
    if (yout && (!gate || !saved || !aligned16(yout))) return VLPET_E_ALIGN;
    b.y = (flags & PET_GATE_ADD) ? nullptr : yout;
    b.red_ctrl = nullptr; b.red_words = 0;
    if (saved && !aligned16(saved)) return VLPET_E_ALIGN;
    b.saved = saved; b.saved_stride = (int64_t)saved_stride(M, tiles, io_dtype);
    if (saved) {        // z comes from the forward; the rows kernel does not write it
        b.z_a = const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(saved));
        if (gate) b.z_g = const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(saved)) + 2 * b.saved_stride;
    }
    // two-pass form (pass 1: dpre only; pass 2: input gradients + weight gradients from recomputed dh / dq) unless the
    // caller needs the input gradients right after phase 1 (phases bit 2: weight gradients on a side stream)
    // bf16, r <= 96: pass 1 (dpre) + the column-parallel pass of pet_cols.hip (input gradients + the four weight gradients from
    // one read of dy, x1, x2) -- unless the caller needs the input gradients right after phase 1
    const bool cols6 = !(phases & 4) && k1_cols6_applies(b, io_dtype == VLPET_F32);
    const bool cols4 = cols6 || (!(phases & 4) && k1_cols_applies(b, io_dtype == VLPET_F32));
    const bool two_pass = cols4 || (!(phases & 4) && pet_gate_bwd3_applies(b));
    const bool rows2 = !two_pass && pet_gate_bwd2_applies(b);
    // without a gate (K2, adapter-only K1, K3 without dropout), bf16, saved activations, whole call: pass 1 (dpre) + the
    // column-parallel pass of pet_cols_ng.hip (dx and both weight gradients from one read of dy and x)
    const bool ng2 = !gate && phases == 3 && vlpet_tuning().ng2 != 0 && ng_two_pass_applies(b, io_dtype == VLPET_F32);
    // round 6: pass 2 sums its row-chunk partials inside the launch (cols_reduce.h) -- no finalize launch; `phases` bit 5 keeps the
    // round-3 two-launch form (same-box A/Bs; the results are bit-identical)
    const bool red4 = cols4 && !cols6 && !(phases & 32) && k1_in_launch_reduce(M);
    if (red4) { b.red_ctrl = reinterpret_cast<unsigned*>(ws + w.red_ctrl); b.red_words = (d / 128) * COLS_RED_STRIDE; }
    int gs3 = 0, ng3 = 0;
    if (ng2) {
        WgradArgs g{};
        g.M = M; g.RT = tiles; g.njobs = 2;
        ng_cols_plan(M, d, &g.row_chunks, &g.rows_per_chunk);
        g.partial = reinterpret_cast<float*>(ws + w.partial);
        const int ldp = 32 * tiles;
        WgradJob& J0 = g.job[0];                        // dWd[c,k] = sum_m dpre[m,c] x[m,k];  dbd = column sums of dpre
        J0.P = b.dp_a; J0.ldp = ldp; J0.pcols = ldp; J0.X = xa; J0.ldx = d; J0.xcols = d; J0.drop = NO_DROP; J0.has_drop = 0;
        J0.scale = drop_active(b.drop) ? b.drop.keep_scale : 1.f;      // (dropout: the kernel clears the dropped x, 1 / (1 - p) here)
        J0.out = dwd; J0.ldo = d; J0.transposed = 0; J0.out_rows = r; J0.colsum_x = nullptr; J0.colsum_p = dbd;
        WgradJob& J1 = g.job[1];                        // dWu[f,c] = sd * sum_m dy[m,f] z[m,c];  dbu = sd * column sums of dy
        J1.P = b.z_a; J1.ldp = ldp; J1.pcols = ldp; J1.X = dy; J1.ldx = d; J1.xcols = d; J1.drop = NO_DROP; J1.has_drop = 0;
        J1.scale = sd; J1.out = dwu; J1.ldo = r; J1.transposed = 1; J1.out_rows = r; J1.colsum_x = dbu; J1.colsum_p = nullptr;
        // (the in-launch reduce-scatter of cols_reduce.h was built for this pass 2 as well and measured no gain: K2 backward 67.9 / 66.7 us
        //  with it against 66.4 / 67.1 us with the finalize launch in the configs[1] step, profiles/r06_in_launch_reduce_ab.txt -- not kept)
        hipError_t e = launch_ng_two_pass(b, g, 3, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
        return herr(launch_wgrad_finalize(g, (hipStream_t)stream));
    }
    if (phases & 1) {
        if (rows2) b.dxg_in = dx1_in;                   // the chain-split row kernel adds it in its epilogue
        const bool dz2 = two_pass && vlpet_tuning().dz2 != 0 && k1_dz2_applies(b, io_dtype == VLPET_F32);
        if (dz2 && gate) {      // small M: feature blocks (their fp32 partial dz in the dh / dq area, which the two-pass form does not use)
            const int nfb = k1_dz2_feature_blocks(M, d);
            const size_t need = (size_t)nfb * (size_t)M * 2 * 32 * tiles * 4;
            if (nfb > 1 && need <= 2 * align256((size_t)M * d * (io_dtype == VLPET_F32 ? 4 : 2))) {
                b.fsplit = nfb;
                b.dz_part = reinterpret_cast<float*>(ws + w.dh);
            }
        }
        const bool dz6 = two_pass && !dz2 && vlpet_tuning().dz6 != 0 && k1_dz6_applies(b, io_dtype == VLPET_F32);
        if (dz6 && gate) {
            const int nfb = k1_dz6_feature_blocks(M, d);
            const size_t need = (size_t)nfb * (size_t)M * 2 * 32 * tiles * 4;
            if (nfb > 1 && w.dq > w.dh && need <= 2 * (w.dq - w.dh)) {
                b.fsplit = nfb;
                b.dz_part = reinterpret_cast<float*>(ws + w.dh);
            }
        }
        if (b.red_ctrl && !dz2) {       // a pass 1 that does not zero the reduce-scatter state itself (non-default forms): a memset node
            hipError_t em = hipMemsetAsync(b.red_ctrl, 0, (size_t)b.red_words * 4, (hipStream_t)stream);
            if (em != hipSuccess) return (int)em;
        }
        hipError_t e = dz2 ? launch_k1_dz2(b, (hipStream_t)stream)
                     : dz6 ? launch_k1_dz6(b, (hipStream_t)stream)
                     : two_pass ? launch_pet_gate_dz(b, io_dtype == VLPET_F32, (hipStream_t)stream)
                     : rows2 ? launch_pet_gate_bwd2(b, io_dtype == VLPET_F32, (hipStream_t)stream)
                             : launch_pet_bwd(b, io_dtype == VLPET_F32, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
        if (dx1_in && !two_pass && !rows2) {            // other row kernels: one more pass over dx1
            e = launch_add_inplace(dxg, dx1_in, M * (int64_t)d, io_dtype == VLPET_F32, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
        }
    }
    if (!(phases & (2 | 16))) return 0;

    WgradArgs g{};
    g.M = M; g.RT = tiles; g.row_chunks = w.row_chunks; g.rows_per_chunk = w.rows_per_chunk;
    if (cols6) k1_cols6_plan(M, d, &g.row_chunks, &g.rows_per_chunk);
    else if (cols4) k1_cols_plan(M, d, &g.row_chunks, &g.rows_per_chunk);
    else if (two_pass) gate_bwd3_plan(M, d, io_dtype == VLPET_F32, &g.row_chunks, &g.rows_per_chunk, &gs3, &ng3);
    g.partial = reinterpret_cast<float*>(ws + w.partial);
    const int ldp = 32 * tiles;
    auto job = [&](int i, const void* P, const void* X, bool dropped, float scale, float* out, int ldo,
                   int transposed, int out_rows, float* csx, float* csp) {
        WgradJob& J = g.job[i];
        J.P = P; J.ldp = ldp; J.pcols = ldp;
        J.X = X; J.ldx = d; J.xcols = d;
        J.drop = dropped ? b.drop : NO_DROP; J.has_drop = dropped && drop_active(b.drop);
        J.scale = scale; J.out = out; J.ldo = ldo; J.transposed = transposed; J.out_rows = out_rows;
        J.colsum_x = csx; J.colsum_p = csp;
    };
    // down weight:  dWd[c,k] = sum_m dpre[m,c] * xa[m,k];  bias = column sums of dpre
    job(0, b.dp_a, xa, true, 1.f, dwd, d, 0, r, nullptr, dbd);
    // up weight:    dWu[f,c] = sd * sum_m dh[m,f] * z[m,c]  (X = dh, or dy itself without a gate)
    job(1, b.z_a, gate ? b.dh : dy, false, sd, dwu, r, 1, r, dbu, nullptr);
    g.njobs = 2;
    if (gate) {
        job(2, b.dp_g, xg, false, 1.f, dwgd, d, 0, rg, nullptr, dbgd);
        job(3, b.z_g, b.dq, false, 1.f, dwgu, rg, 1, rg, dbgu, nullptr);
        g.njobs = 4;
    }
    if (cols4) {
        ColzArgs c{};
        c.dy = dy; c.x1 = xg; c.x2 = res; c.dxin = dx1_in; c.y = b.y;
        c.z_a = b.z_a; c.z_g = b.z_g; c.dp_a = b.dp_a; c.dp_g = b.dp_g;
        c.dx1 = dxg; c.dx2 = dxa;
        c.pk_a = b.pk_a; c.pk_g = b.pk_g;
        c.M = M; c.d = d; c.s2 = s2; c.sd = sd; c.gs = gs; c.flags = flags;
        c.row_chunks = g.row_chunks; c.rows_per_chunk = g.rows_per_chunk;
        const WgradLayout L = wgrad_layout(g);
        for (int j = 0; j < 4; ++j) c.part[j] = g.partial + L.off[j];
        if (red4) {
            // the same workspace bytes, cut differently: [RC][NCB] slabs (= the four [RC][PR][d] blocks), then the column-sum partials
            const int64_t PRl = 32 * tiles;
            c.red.slab = g.partial;
            c.red.bias_x = g.partial + (int64_t)4 * g.row_chunks * PRl * d;
            c.red.bias_p = c.red.bias_x + (int64_t)2 * g.row_chunks * d;
            c.red.ctrl = reinterpret_cast<unsigned*>(ws + w.red_ctrl);
            c.red.spin_limit = COLS_RED_SPIN_DEFAULT;
            for (int j = 0; j < 4; ++j) {
                const WgradJob& J = g.job[j];
                c.red.job[j] = ColsRedJob{J.out, J.ldo, J.transposed, J.out_rows, J.scale, J.colsum_x, J.colsum_p};
            }
            if (!(phases & 2)) return 0;                    // ("finalize only": there is none)
            return herr(launch_k1_cols(c, tiles, (hipStream_t)stream));
        }
        if (phases & 2) {
            hipError_t e = cols6 ? launch_k1_cols6y(c, (hipStream_t)stream)
                                 : launch_k1_cols(c, tiles, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
        }
        if ((phases & 8) && !(phases & 16)) return 0;       // (the partial sums stay in the workspace)
        return herr(launch_wgrad_finalize(g, (hipStream_t)stream));
    }
    if (!(phases & 2)) return 0;                            // (bits 3 / 4 split the column-parallel form only)
    if (two_pass) {                                     // (dx1 is an output of pass 2 there)
        hipError_t e = launch_pet_gate_cols(b, g, gs3, ng3, io_dtype == VLPET_F32, (hipStream_t)stream);
        if (e == hipSuccess && dx1_in) e = launch_add_inplace(dxg, dx1_in, M * (int64_t)d, io_dtype == VLPET_F32, (hipStream_t)stream);
        return herr(e);
    }
    return herr(launch_wgrad(g, io_dtype == VLPET_F32, (hipStream_t)stream));
}

extern "C" int vlpet_adapter_gate_bwd_form(int64_t M, int d, int tiles, int io_dtype) {
    if (check_common(M, d, tiles, io_dtype)) return -1;
    PetBwdArgs b{};
    b.M = M; b.d = d; b.RT = tiles; b.flags = PET_GATE; b.saved = &b; b.drop = NO_DROP; b.y = &b;      // (the default path hands the forward's output over)
    if (k1_cols_applies(b, io_dtype == VLPET_F32) || k1_cols6_applies(b, io_dtype == VLPET_F32)) return 2;
    if (pet_gate_bwd3_applies(b)) return 1;
    return 0;
}

// 1: the two-pass form at this shape ends in a separate finalize launch (`phases` bit 4 runs it); 0: pass 2 sums its row chunks itself
// (round 6, cols_reduce.h) or the form is not the two-pass one
extern "C" int vlpet_adapter_gate_bwd_finalize_launch(int64_t M, int d, int tiles, int io_dtype) {
    if (check_common(M, d, tiles, io_dtype)) return -1;
    PetBwdArgs b{};
    b.M = M; b.d = d; b.RT = tiles; b.flags = PET_GATE; b.saved = &b; b.drop = NO_DROP; b.y = &b;
    if (k1_cols6_applies(b, io_dtype == VLPET_F32)) return 1;
    if (k1_cols_applies(b, io_dtype == VLPET_F32)) return k1_in_launch_reduce(M) ? 0 : 1;
    return 0;
}

extern "C" int vlpet_adapter_gate_bwd(const void* dy, const void* x1, const void* x2,
                                      const void* packed_a, const void* packed_g, void* dx1, void* dx2,
                                      float* dwd, float* dbd, float* dwu, float* dbu,
                                      float* dwgd, float* dbgd, float* dwgu, float* dbgu, int r, int rg,
                                      void* workspace, size_t workspace_bytes, int64_t M, int d, int tiles,
                                      int gate_mode, float delta_scale, float x2_scale, float gate_scale,
                                      int io_dtype, vlpet_stream_t stream) {
    int flags = 0;
    if (gate_mode == VLPET_GATE_MUL) flags = PET_GATE;
    else if (gate_mode == VLPET_GATE_ADD) flags = PET_GATE | PET_GATE_ADD;
    else if (gate_mode != VLPET_GATE_NONE) return VLPET_E_SHAPE;
    if (!dbd || !dbu) return VLPET_E_NULL;
    return run_bwd(dy, x2, x2, x1, packed_a, packed_g, NO_DROP, dx2, dx1, dwd, dbd, dwu, dbu,
                   dwgd, dbgd, dwgu, dbgu, r, rg, workspace, workspace_bytes, M, d, tiles,
                   x2_scale, delta_scale, flags ? gate_scale : 1.f, flags, io_dtype, stream);
}

static int gate_flags(int gate_mode, int* flags) {
    *flags = 0;
    if (gate_mode == VLPET_GATE_MUL) *flags = PET_GATE;
    else if (gate_mode == VLPET_GATE_ADD) *flags = PET_GATE | PET_GATE_ADD;
    else if (gate_mode != VLPET_GATE_NONE) return VLPET_E_SHAPE;
    return 0;
}

extern "C" int vlpet_adapter_gate_bwd_phase(int phases, const void* dy, const void* x1, const void* x2,
                                            const void* packed_a, const void* packed_g, void* dx1, void* dx2,
                                            float* dwd, float* dbd, float* dwu, float* dbu,
                                            float* dwgd, float* dbgd, float* dwgu, float* dbgu, int r, int rg,
                                            void* workspace, size_t workspace_bytes, int64_t M, int d, int tiles,
                                            int gate_mode, float delta_scale, float x2_scale, float gate_scale,
                                            int io_dtype, vlpet_stream_t stream) {
    int flags;
    if (gate_flags(gate_mode, &flags)) return VLPET_E_SHAPE;
    if (!dbd || !dbu || (phases & 3) == 0) return VLPET_E_NULL;
    return run_bwd(dy, x2, x2, x1, packed_a, packed_g, NO_DROP, dx2, dx1, dwd, dbd, dwu, dbu,
                   dwgd, dbgd, dwgu, dbgu, r, rg, workspace, workspace_bytes, M, d, tiles,
                   x2_scale, delta_scale, flags ? gate_scale : 1.f, flags, io_dtype, stream, phases & 7);
}

extern "C" int vlpet_adapter_gate_bwd_saved(int phases, const void* dy, const void* x1, const void* x2, const void* saved,
                                            const void* packed_a, const void* packed_g, void* dx1, void* dx2,
                                            float* dwd, float* dbd, float* dwu, float* dbu,
                                            float* dwgd, float* dbgd, float* dwgu, float* dbgu, int r, int rg,
                                            void* workspace, size_t workspace_bytes, int64_t M, int d, int tiles,
                                            int gate_mode, float delta_scale, float x2_scale, float gate_scale,
                                            int io_dtype, vlpet_stream_t stream) {
    int flags;
    if (gate_flags(gate_mode, &flags)) return VLPET_E_SHAPE;
    if (!dbd || !dbu || !saved || (phases & 19) == 0) return VLPET_E_NULL;
    return run_bwd(dy, x2, x2, x1, packed_a, packed_g, NO_DROP, dx2, dx1, dwd, dbd, dwu, dbu,
                   dwgd, dbgd, dwgu, dbgu, r, rg, workspace, workspace_bytes, M, d, tiles,
                   x2_scale, delta_scale, flags ? gate_scale : 1.f, flags, io_dtype, stream, phases & 31, saved);
}

extern "C" int vlpet_adapter_gate_bwd_saved_acc(int phases, const void* dy, const void* x1, const void* x2, const void* saved,
                                                const void* packed_a, const void* packed_g, const void* dx1_in, void* dx1, void* dx2,
                                                float* dwd, float* dbd, float* dwu, float* dbu,
                                                float* dwgd, float* dbgd, float* dwgu, float* dbgu, int r, int rg,
                                                void* workspace, size_t workspace_bytes, int64_t M, int d, int tiles,
                                                int gate_mode, float delta_scale, float x2_scale, float gate_scale,
                                                int io_dtype, vlpet_stream_t stream) {
    int flags;
    if (gate_flags(gate_mode, &flags)) return VLPET_E_SHAPE;
    if (!dbd || !dbu || !saved || !dx1_in || !flags || (phases & 19) == 0) return VLPET_E_NULL;
    return run_bwd(dy, x2, x2, x1, packed_a, packed_g, NO_DROP, dx2, dx1, dwd, dbd, dwu, dbu,
                   dwgd, dbgd, dwgu, dbgu, r, rg, workspace, workspace_bytes, M, d, tiles,
                   x2_scale, delta_scale, gate_scale, flags, io_dtype, stream, phases & 31, saved, dx1_in);
}

// The same with the forward's output y at hand (round 5): pass 1 of the two-pass forms then needs the gate chain's up projection
// only (dq = dy * y * (1 - g)); dx1_in is optional here.  y == NULL: exactly vlpet_adapter_gate_bwd_saved / _acc.
extern "C" int vlpet_adapter_gate_bwd_saved_y(int phases, const void* dy, const void* x1, const void* x2, const void* y, const void* saved,
                                              const void* packed_a, const void* packed_g, const void* dx1_in, void* dx1, void* dx2,
                                              float* dwd, float* dbd, float* dwu, float* dbu,
                                              float* dwgd, float* dbgd, float* dwgu, float* dbgu, int r, int rg,
                                              void* workspace, size_t workspace_bytes, int64_t M, int d, int tiles,
                                              int gate_mode, float delta_scale, float x2_scale, float gate_scale,
                                              int io_dtype, vlpet_stream_t stream) {
    int flags;
    if (gate_flags(gate_mode, &flags)) return VLPET_E_SHAPE;
    if (!dbd || !dbu || !saved || !flags || (phases & 19) == 0) return VLPET_E_NULL;
    return run_bwd(dy, x2, x2, x1, packed_a, packed_g, NO_DROP, dx2, dx1, dwd, dbd, dwu, dbu,
                   dwgd, dbgd, dwgu, dbgu, r, rg, workspace, workspace_bytes, M, d, tiles,
                   x2_scale, delta_scale, gate_scale, flags, io_dtype, stream, phases & 31, saved, dx1_in, y);
}

extern "C" int vlpet_parallel_adapter_bwd(const void* dy, const void* x, const void* packed, void* dx,
                                          float* dwd, float* dbd, float* dwu, float* dbu, int r,
                                          void* workspace, size_t workspace_bytes, int64_t M, int d,
                                          int tiles, float scale, int io_dtype, vlpet_stream_t stream) {
    if (!dbd || !dbu) return VLPET_E_NULL;
    return run_bwd(dy, x, nullptr, nullptr, packed, nullptr, NO_DROP, dx, nullptr, dwd, dbd, dwu, dbu,
                   nullptr, nullptr, nullptr, nullptr, r, 0, workspace, workspace_bytes, M, d, tiles,
                   1.f, scale, 1.f, 0, io_dtype, stream);
}

extern "C" int vlpet_parallel_adapter_bwd_saved(const void* dy, const void* x, const void* saved, const void* packed, void* dx,
                                                float* dwd, float* dbd, float* dwu, float* dbu, int r,
                                                void* workspace, size_t workspace_bytes, int64_t M, int d,
                                                int tiles, float scale, int io_dtype, vlpet_stream_t stream) {
    if (!dbd || !dbu || !saved) return VLPET_E_NULL;
    return run_bwd(dy, x, nullptr, nullptr, packed, nullptr, NO_DROP, dx, nullptr, dwd, dbd, dwu, dbu,
                   nullptr, nullptr, nullptr, nullptr, r, 0, workspace, workspace_bytes, M, d, tiles,
                   1.f, scale, 1.f, 0, io_dtype, stream, 3, saved);
}

extern "C" int vlpet_lora_delta_bwd(const void* dy, const void* x, const void* packed,
                                    const uint8_t* keep_mask, float p, uint64_t seed, void* dx,
                                    float* da, float* db, int r, void* workspace, size_t workspace_bytes,
                                    int64_t M, int d, int tiles, float scaling, int io_dtype,
                                    vlpet_stream_t stream) {
    DropSpec ds;
    if (int rc = make_drop(keep_mask, p, seed, nullptr, &ds)) return rc;
    return run_bwd(dy, x, nullptr, nullptr, packed, nullptr, ds, dx, nullptr,
                   da, nullptr, db, nullptr, nullptr, nullptr, nullptr, nullptr, r, 0,
                   workspace, workspace_bytes, M, d, tiles, 1.f, scaling, 1.f, PET_ACT_IDENTITY,
                   io_dtype, stream);
}

extern "C" int vlpet_lora_delta_bwd_saved(const void* dy, const void* x, const void* saved, const void* packed,
                                          const uint8_t* keep_mask, float p, uint64_t seed, void* dx,
                                          float* da, float* db, int r, void* workspace, size_t workspace_bytes,
                                          int64_t M, int d, int tiles, float scaling, int io_dtype,
                                          vlpet_stream_t stream) {
    if (!saved) return VLPET_E_NULL;
    DropSpec ds;
    if (int rc = make_drop(keep_mask, p, seed, nullptr, &ds)) return rc;
    return run_bwd(dy, x, nullptr, nullptr, packed, nullptr, ds, dx, nullptr,
                   da, nullptr, db, nullptr, nullptr, nullptr, nullptr, nullptr, r, 0,
                   workspace, workspace_bytes, M, d, tiles, 1.f, scaling, 1.f, PET_ACT_IDENTITY,
                   io_dtype, stream, 3, saved);
}


// ------------------------------------------------------------------ K4: visual projection
static inline bool dout_ok(int d) { return d == 64 || d == 128 || d == 768; }

extern "C" size_t vlpet_visproj_packed_bytes(int d_out, int feat_dim, int io_dtype) {
    const int NS = io_dtype == VLPET_F32 ? 2 : 1;
    return align256((size_t)(feat_dim / 16) * (d_out / 32) * NS * 1024 + (size_t)(d_out + feat_dim) * 4);
}

extern "C" int vlpet_visproj_pack(const void* w, const void* b, int d_out, int feat_dim, int param_dtype,
                                  int io_dtype, void* packed, vlpet_stream_t stream) {
    if (!w || !packed) return VLPET_E_NULL;
    if (!dout_ok(d_out) || feat_dim <= 0 || feat_dim % 64 != 0) return VLPET_E_SHAPE;
    if (!dtype_ok(param_dtype) || !dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    if (!aligned16(packed)) return VLPET_E_ALIGN;
    PackArgs a;
    for (int i = 0; i < VLPET_MAX_HEADS; ++i) { a.wd[i] = nullptr; a.bd[i] = nullptr; }
    a.wd[0] = w; a.bd[0] = b;
    a.wu = nullptr; a.bu = nullptr;
    a.n_heads = 1; a.rows_per_head = d_out;
    a.r = d_out; a.d = feat_dim; a.RT = d_out / 32; a.src_bf16 = param_dtype == VLPET_BF16;
    a.n_packs = 1;
    a.out = reinterpret_cast<uint8_t*>(packed);
    return herr(launch_pack_pair(a, io_dtype == VLPET_F32 ? 2 : 1, (hipStream_t)stream));
}

extern "C" int vlpet_visproj_fwd(const void* feats, const void* packed, const float* gamma, const float* beta,
                                 const void* r, void* out, void* xhat, float* rstd, int64_t M, int feat_dim,
                                 int d_out, float eps, int rms, int io_dtype, vlpet_stream_t stream) {
    if (!feats || !packed || !gamma || !out) return VLPET_E_NULL;
    if (M <= 0 || !dout_ok(d_out) || feat_dim <= 0 || feat_dim % 64 != 0) return VLPET_E_SHAPE;
    if (!dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    if (!aligned16(feats) || !aligned16(packed) || !aligned16(out) || (r && !aligned16(r)) || (xhat && !aligned16(xhat)))
        return VLPET_E_ALIGN;
    VisprojArgs a;
    a.feats = feats; a.pk = reinterpret_cast<const uint8_t*>(packed); a.gamma = gamma; a.beta = beta;
    a.R = r; a.out = out; a.xhat = xhat; a.rstd = rstd;
    a.M = M; a.F = feat_dim; a.d_out = d_out; a.eps = eps; a.rms = rms;
    return herr(launch_visproj_fwd(a, io_dtype == VLPET_F32, (hipStream_t)stream));
}

// K4 forward as a tiled GEMM (visproj_gemm.hip): bf16, d_out a multiple of 256 (<= 1024), feat_dim a multiple of 64.
extern "C" size_t vlpet_visproj_gemm_workspace_bytes(int64_t M, int feat_dim, int d_out) {
    return visproj_gemm_workspace_bytes(M, feat_dim, d_out);
}
// out = the sum of n tensors of `len` IO-dtype elements (n >= 1 sources as an array of device pointers on the HOST; out may be srcs[0]):
// fp32 accumulation, one rounding per launch of up to eight sources
extern "C" int vlpet_sum_n(const void* const* srcs, int n, void* out, int64_t len, int io_dtype, vlpet_stream_t stream) {
    if (!srcs || !out) return VLPET_E_NULL;
    if (n < 1 || len <= 0 || len % 8 != 0) return VLPET_E_SHAPE;
    if (!dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    if (!aligned16(out)) return VLPET_E_ALIGN;
    for (int i = 0; i < n; ++i) {
        if (!srcs[i]) return VLPET_E_NULL;
        if (!aligned16(srcs[i])) return VLPET_E_ALIGN;
    }
    const size_t esz = io_dtype == VLPET_F32 ? 4 : 2;
    if (n == 1) {
        if (srcs[0] == out) return 0;
        return herr(hipMemcpyAsync(out, srcs[0], (size_t)len * esz, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    int done = 0;
    while (done < n) {                      // eight sources per launch; later launches start from the running sum
        SumNArgs a{};
        int k = 0;
        if (done > 0) a.src[k++] = out;
        while (k < 8 && done < n) a.src[k++] = srcs[done++];
        a.out = out;
        if (k == 1) break;
        hipError_t e = launch_sum_n(a, k, len, io_dtype == VLPET_F32, (hipStream_t)stream);
        if (e != hipSuccess) return herr(e);
    }
    return 0;
}

// K4's position / order branch (vispos.hip).  table dtypes: VLPET_F32 / VLPET_BF16 each; ids int64 [B or 1, N] with a batch stride of N or 0
// (nullptr: image 0 / object n, the reference's defaults, src/modeling_bart.py:169-177)
extern "C" int vlpet_vispos_applies(int d, int n_img) { return vispos_applies(d, n_img) ? 1 : 0; }
extern "C" size_t vlpet_vispos_bwd_workspace_bytes(int64_t M, int d, int n_img) {
    return (M > 0 && vispos_applies(d, n_img)) ? vispos_bwd_workspace_bytes(M, d, n_img) : 0;
}
static int vispos_common(const float* pos, const float* w, const float* b, const float* gamma, int64_t M, int N, int d, int n_img, int io_dtype) {
    if (!pos || !w || !b || !gamma) return VLPET_E_NULL;
    if (M <= 0 || M >= ((int64_t)1 << 31) || N <= 0 || M % N != 0 || !vispos_applies(d, n_img)) return VLPET_E_SHAPE;
    if (!dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    if (!aligned16(pos) || !aligned16(w) || !aligned16(b) || !aligned16(gamma)) return VLPET_E_ALIGN;
    return 0;
}
extern "C" int vlpet_vispos_fwd(const float* pos, const float* w, const float* b, const float* gamma, const float* beta,
                                const void* img_table, int img_table_dtype, int n_img, const int64_t* img_ids, int64_t img_ids_bstride,
                                const void* obj_table, int obj_table_dtype, int64_t obj_rows, const int64_t* obj_ids, int64_t obj_ids_bstride,
                                void* out, int64_t M, int N, int d, float eps, int rms, int io_dtype, vlpet_stream_t stream) {
    const bool tabs = img_table != nullptr || obj_table != nullptr;
    if (int rc = vispos_common(pos, w, b, gamma, M, N, d, tabs ? n_img : 0, io_dtype)) return rc;
    if (!out || (!rms && !beta)) return VLPET_E_NULL;
    if (tabs) {
        if (!img_table || !obj_table) return VLPET_E_NULL;              // the reference adds both or neither (use_vis_order_embedding)
        if (n_img < 1 || obj_rows < 1) return VLPET_E_SHAPE;
        if (!dtype_ok(img_table_dtype) || !dtype_ok(obj_table_dtype)) return VLPET_E_DTYPE;
        if (!aligned16(img_table) || !aligned16(obj_table)) return VLPET_E_ALIGN;
    }
    if (!aligned16(out) || (beta && !aligned16(beta))) return VLPET_E_ALIGN;
    VisPosArgs a{};
    a.pos = pos; a.w = w; a.b = b; a.gamma = gamma; a.beta = rms ? nullptr : beta;
    a.img_tab = img_table; a.obj_tab = obj_table; a.img_tab_bf16 = img_table_dtype == VLPET_BF16; a.obj_tab_bf16 = obj_table_dtype == VLPET_BF16;
    a.img_ids = img_ids; a.obj_ids = obj_ids; a.img_bstride = img_ids_bstride; a.obj_bstride = obj_ids_bstride;
    a.n_img = n_img; a.obj_rows = obj_rows; a.out = out; a.M = M; a.N = N; a.d = d; a.eps = eps; a.rms = rms;
    return herr(launch_vispos_fwd(a, io_dtype == VLPET_F32, (hipStream_t)stream));
}
// dout = the gradient of the visual embedding's output (= dR); results are WRITTEN: dw [d, 5], db [d], dgamma [d], dbeta [d] (nullptr with
// rms), dimg [n_img, d] (nullptr / n_img = 0: no image-order table)
extern "C" int vlpet_vispos_bwd(const void* dout, const float* pos, const float* w, const float* b, const float* gamma,
                                int n_img, const int64_t* img_ids, int64_t img_ids_bstride,
                                float* dw, float* db, float* dgamma, float* dbeta, float* dimg,
                                void* workspace, size_t workspace_bytes, int64_t M, int N, int d, float eps, int rms, int io_dtype,
                                vlpet_stream_t stream) {
    if (!dimg) n_img = 0;
    if (int rc = vispos_common(pos, w, b, gamma, M, N, d, n_img, io_dtype)) return rc;
    if (!dout || !dw || !db || !dgamma || !workspace || (!rms && !dbeta)) return VLPET_E_NULL;
    if (!aligned16(dout) || !aligned16(workspace)) return VLPET_E_ALIGN;
    if (workspace_bytes < vispos_bwd_workspace_bytes(M, d, n_img)) return VLPET_E_WORKSPACE;
    VisPosArgs a{};
    a.pos = pos; a.w = w; a.b = b; a.gamma = gamma;
    a.img_ids = img_ids; a.img_bstride = img_ids_bstride; a.n_img = n_img;
    a.dout = dout; a.partial = reinterpret_cast<float*>(workspace);
    a.dw = dw; a.db = db; a.dgamma = dgamma; a.dbeta = rms ? nullptr : dbeta; a.dimg = dimg;
    a.M = M; a.N = N; a.d = d; a.eps = eps; a.rms = rms;
    return herr(launch_vispos_bwd(a, io_dtype == VLPET_F32, (hipStream_t)stream));
}

extern "C" size_t vlpet_visproj_gemm_exchange_bytes(int d_out) {
    return (d_out > 0 && d_out % 256 == 0 && d_out / 256 <= 4) ? visproj_gemm_exchange_bytes(d_out) : 0;
}
static int visproj_gemm_call(const void* feats, const void* w_io, const float* bias, const float* gamma, const float* beta,
                             const void* r, void* out, void* xhat, float* rstd, float* mean, void* workspace, size_t workspace_bytes,
                             int64_t M, int feat_dim, int d_out, float eps, int rms, int io_dtype, int form, int bm, vlpet_stream_t stream) {
    if (!feats || !w_io || !gamma || !out || !workspace) return VLPET_E_NULL;
    if (!xhat && d_out > 256) return VLPET_E_NULL;       // (a workgroup that gives up on its partners parks its pre-norm tile there)
    if (!dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    if (!visproj_gemm_applies(M, feat_dim, d_out, io_dtype == VLPET_F32)) return VLPET_E_SHAPE;
    if (workspace_bytes < visproj_gemm_workspace_bytes(M, feat_dim, d_out)) return VLPET_E_WORKSPACE;
    if (!aligned16(feats) || !aligned16(w_io) || !aligned16(out) || (r && !aligned16(r)) || (xhat && !aligned16(xhat)) || !aligned16(workspace))
        return VLPET_E_ALIGN;
    VisGemmArgs a{};
    a.feats = feats; a.w = w_io; a.bias = bias; a.gamma = gamma; a.beta = beta; a.R = r; a.out = out; a.xhat = xhat; a.rstd = rstd;
    a.mean = mean; a.M = M; a.F = feat_dim; a.d_out = d_out; a.eps = eps; a.rms = rms;
    return herr(launch_visproj_gemm(a, workspace, form, bm, (hipStream_t)stream));
}
extern "C" int vlpet_visproj_fwd_gemm(const void* feats, const void* w_io, const float* bias, const float* gamma, const float* beta,
                                      const void* r, void* out, void* xhat, float* rstd, float* mean, void* workspace,
                                      size_t workspace_bytes, int64_t M, int feat_dim, int d_out, float eps, int rms, int io_dtype,
                                      vlpet_stream_t stream) {
    return visproj_gemm_call(feats, w_io, bias, gamma, beta, r, out, xhat, rstd, mean, workspace, workspace_bytes, M, feat_dim, d_out,
                             eps, rms, io_dtype, 0, 0, stream);
}
// the same with the ring form (1: BK 64 / 2 slots, 2: BK 32 / 4 slots, 3: BK 32 / 3 slots; 0: default) and the rows per workgroup
// (128 / 256; 0: by shape) forced -- tools/k4bench.py and the parity tests of the non-default forms
extern "C" int vlpet_visproj_fwd_gemm_cfg(const void* feats, const void* w_io, const float* bias, const float* gamma, const float* beta,
                                          const void* r, void* out, void* xhat, float* rstd, float* mean, void* workspace,
                                          size_t workspace_bytes, int64_t M, int feat_dim, int d_out, float eps, int rms, int io_dtype,
                                          int form, int rows_per_workgroup, vlpet_stream_t stream) {
    if (form < 0 || (form & 255) > 6 || (form >> 8) > 31 || (rows_per_workgroup != 0 && rows_per_workgroup != 128 && rows_per_workgroup != 192 && rows_per_workgroup != 256)) return VLPET_E_SHAPE;
    return visproj_gemm_call(feats, w_io, bias, gamma, beta, r, out, xhat, rstd, mean, workspace, workspace_bytes, M, feat_dim, d_out,
                             eps, rms, io_dtype, form, rows_per_workgroup, stream);
}

static void visproj_wgrad_plan(int64_t M, int feat_dim, int d_out, int* RT, int* pcols, int* rc, int64_t* rpc) {
    *RT = (d_out % 96 == 0) ? 3 : 1;
    *pcols = 32 * *RT;
    wgrad_plan(M, 4, feat_dim, rc, rpc);
}

// The tiled split-K form (bf16, visproj_wgrad.hip) keeps one fp32 [d_out, feat_dim] partial per row chunk: 16 chunks x 6.3 MB = 100 MB at
// feat_dim 2048 -> 768, written once and read once by the finalize launch (16.5 of the kernel's 87 us); the job-stream form of the
// fp32 path needs a few MB.  The size is asked for per IO dtype; the dtype-less entry point returns the larger of the two.
extern "C" size_t vlpet_visproj_wgrad_workspace_bytes_io(int64_t M, int feat_dim, int d_out, int io_dtype) {
    if (M <= 0 || feat_dim <= 0 || d_out <= 0 || !dtype_ok(io_dtype)) return 0;
    if (vlpet_tuning().k4_wgrad2 != 0 && k4_wgrad2_applies(M, feat_dim, d_out, io_dtype == VLPET_F32))
        return align256(k4_wgrad2_workspace_bytes(M, feat_dim, d_out));
    int RT, pcols, rc; int64_t rpc;
    visproj_wgrad_plan(M, feat_dim, d_out, &RT, &pcols, &rc, &rpc);
    return align256(wgrad_workspace_bytes(4, RT, feat_dim, rc));
}
extern "C" size_t vlpet_visproj_wgrad_workspace_bytes(int64_t M, int feat_dim, int d_out) {
    const size_t a = vlpet_visproj_wgrad_workspace_bytes_io(M, feat_dim, d_out, VLPET_BF16);
    const size_t b = vlpet_visproj_wgrad_workspace_bytes_io(M, feat_dim, d_out, VLPET_F32);
    return a > b ? a : b;
}

extern "C" int vlpet_visproj_wgrad(const void* dpre, const void* feats, float* dw, float* db, void* workspace,
                                   size_t workspace_bytes, int64_t M, int feat_dim, int d_out, int io_dtype,
                                   vlpet_stream_t stream) {
    if (!dpre || !feats || !dw || !db || !workspace) return VLPET_E_NULL;
    if (M <= 0 || d_out % 32 != 0 || feat_dim <= 0 || feat_dim % 64 != 0) return VLPET_E_SHAPE;
    if (!dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    if (!aligned16(dpre) || !aligned16(feats) || !aligned16(workspace)) return VLPET_E_ALIGN;
    if (vlpet_tuning().k4_wgrad2 != 0 && k4_wgrad2_applies(M, feat_dim, d_out, io_dtype == VLPET_F32)) {   // tiled split-K GEMM (round 3)
        if (workspace_bytes < k4_wgrad2_workspace_bytes(M, feat_dim, d_out)) return VLPET_E_WORKSPACE;
        return herr(launch_k4_wgrad2(dpre, feats, dw, db, workspace, M, feat_dim, d_out, (hipStream_t)stream));
    }
    int RT, pcols, rc; int64_t rpc;
    visproj_wgrad_plan(M, feat_dim, d_out, &RT, &pcols, &rc, &rpc);
    if (workspace_bytes < wgrad_workspace_bytes(4, RT, feat_dim, rc)) return VLPET_E_WORKSPACE;
    const size_t esz = io_dtype == VLPET_F32 ? 4 : 2;
    const int njobs_total = d_out / pcols;
    for (int j0 = 0; j0 < njobs_total; j0 += 4) {
        WgradArgs g{};
        g.M = M; g.RT = RT; g.row_chunks = rc; g.rows_per_chunk = rpc;
        g.partial = reinterpret_cast<float*>(workspace);
        g.njobs = njobs_total - j0 < 4 ? njobs_total - j0 : 4;
        for (int j = 0; j < g.njobs; ++j) {
            WgradJob& J = g.job[j];
            const int c0 = (j0 + j) * pcols;
            J.P = reinterpret_cast<const uint8_t*>(dpre) + (size_t)c0 * esz; J.ldp = d_out; J.pcols = pcols;
            J.X = feats; J.ldx = feat_dim; J.xcols = feat_dim;
            J.drop = NO_DROP; J.has_drop = 0; J.scale = 1.f;
            J.out = dw + (size_t)c0 * feat_dim; J.ldo = feat_dim; J.transposed = 0; J.out_rows = pcols;
            J.colsum_x = nullptr; J.colsum_p = db + c0;
        }
        hipError_t e = launch_wgrad(g, io_dtype == VLPET_F32, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// ------------------------------------------------------------------ f4: low-rank visual projector
// LowRankVisualEmbedding (src/modeling_bart.py:195-334):  fe = up(gelu_new(cat_i down_i(feats))) [* (sigmoid(gup(gelu_new(gdown(feats)))) (+ 1))]
// One packed buffer per projection pair: [square pair pack at d_out: up, up_t, biases (down slots zero)] [down-only pack at feat_dim].
static inline bool lowrank_dims_ok(int feat_dim, int d_out) {
    return feat_dim > 0 && feat_dim % 64 == 0 && d_out > 0 && d_out % 64 == 0;
}
static inline size_t lowrank_sq_bytes(int tiles, int d_out, int NS) { return align256((size_t)pack_geom(tiles, d_out, NS).total_bytes); }
static inline size_t lowrank_dn_bytes(int tiles, int feat_dim, int NS) {
    return align256((size_t)(feat_dim / 16) * tiles * NS * 1024 + (size_t)(32 * tiles + feat_dim) * 4);
}
extern "C" size_t vlpet_lowrank_packed_bytes(int tiles, int feat_dim, int d_out, int io_dtype) {
    if (!(tiles == 1 || tiles == 3) || !lowrank_dims_ok(feat_dim, d_out)) return 0;
    const int NS = io_dtype == VLPET_F32 ? 2 : 1;
    return lowrank_sq_bytes(tiles, d_out, NS) + lowrank_dn_bytes(tiles, feat_dim, NS);
}

extern "C" int vlpet_lowrank_pack(const void* const* wd_heads, const void* const* bd_heads, int n_heads,
                                  const void* wu, const void* bu, int r, int feat_dim, int d_out, int tiles,
                                  int param_dtype, int io_dtype, void* packed, vlpet_stream_t stream) {
    if (!wd_heads || !wu || !packed) return VLPET_E_NULL;
    if (!lowrank_dims_ok(feat_dim, d_out) || r <= 0 || n_heads <= 0 || n_heads > VLPET_MAX_HEADS || r % n_heads != 0) return VLPET_E_SHAPE;
    if (!(tiles == 1 || tiles == 3) || r > 32 * tiles) return VLPET_E_RANK;
    if (!dtype_ok(param_dtype) || !dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    if (!aligned16(packed)) return VLPET_E_ALIGN;
    const int NS = io_dtype == VLPET_F32 ? 2 : 1;
    PackArgs a;
    for (int i = 0; i < VLPET_MAX_HEADS; ++i) {
        a.wd[i] = nullptr;
        a.bd[i] = (bd_heads && i < n_heads) ? bd_heads[i] : nullptr;
        if (i < n_heads && wd_heads[i] == nullptr) return VLPET_E_NULL;
    }
    a.n_heads = n_heads; a.rows_per_head = r / n_heads;
    a.r = r; a.RT = tiles; a.src_bf16 = param_dtype == VLPET_BF16;
    // (i) the up side at d_out: up / up_t fragments and both biases; no down weight (its slots come out zero)
    a.wu = wu; a.bu = bu; a.d = d_out; a.n_packs = 4;
    a.out = reinterpret_cast<uint8_t*>(packed);
    hipError_t e = launch_pack_pair(a, NS, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    // (ii) the down side at feat_dim: down fragments only
    for (int i = 0; i < n_heads; ++i) a.wd[i] = wd_heads[i];
    a.wu = nullptr; a.bu = nullptr; a.d = feat_dim; a.n_packs = 1;
    a.out = reinterpret_cast<uint8_t*>(packed) + lowrank_sq_bytes(tiles, d_out, NS);
    return herr(launch_pack_pair(a, NS, (hipStream_t)stream));
}

static int lowrank_common(int64_t M, int feat_dim, int d_out, int tiles, int io_dtype) {
    if (M <= 0 || !lowrank_dims_ok(feat_dim, d_out)) return VLPET_E_SHAPE;
    if (io_dtype == VLPET_F32 && (feat_dim % 32 != 0)) return VLPET_E_SHAPE;
    if (!(tiles == 1 || tiles == 3)) return VLPET_E_RANK;
    if (!dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    return 0;
}

// packed_g == NULL: ungated projector (the gate chain runs on packed_a with multiplier 0, offset 1).
// gate_residual: 1 = fe + fe*gate (use_visual_projector_residual_connection), 0 = fe*gate.
extern "C" int vlpet_lowrank_gate_fwd(const void* feats, const void* packed_a, const void* packed_g, void* fe,
                                      void* saved, int64_t M, int feat_dim, int d_out, int tiles, int gate_residual,
                                      int io_dtype, vlpet_stream_t stream) {
    int rc = lowrank_common(M, feat_dim, d_out, tiles, io_dtype);
    if (rc) return rc;
    if (!feats || !packed_a || !fe) return VLPET_E_NULL;
    if (!aligned16(feats) || !aligned16(packed_a) || !aligned16(fe) || (packed_g && !aligned16(packed_g)) ||
        (saved && !aligned16(saved))) return VLPET_E_ALIGN;
    const int NS = io_dtype == VLPET_F32 ? 2 : 1;
    const size_t sq = lowrank_sq_bytes(tiles, d_out, NS);
    const bool gated = packed_g != nullptr;
    const uint8_t* pa = reinterpret_cast<const uint8_t*>(packed_a);
    const uint8_t* pgt = gated ? reinterpret_cast<const uint8_t*>(packed_g) : pa;
    PetFwdArgs a;
    a.xa = feats; a.res = nullptr; a.xg = feats; a.out = fe;
    a.pk_a = pa; a.pk_g = pgt;
    a.pk_a_dn = pa + sq; a.pk_g_dn = pgt + sq;
    a.drop = NO_DROP;
    a.M = M; a.d = d_out; a.d_in = feat_dim; a.RT = tiles;
    a.s2 = 0.f; a.sd = 1.f; a.gs = 1.f; a.flags = PET_GATE;
    a.gm = gated ? 1.f : 0.f; a.go = gated ? (gate_residual ? 1.f : 0.f) : 1.f;
    a.save = saved; a.save_stride = (int64_t)saved_stride(M, tiles, io_dtype);
    a.dbg = 0; a.dbg_ts = nullptr;
    return herr(launch_pet_lowrank_fwd(a, io_dtype == VLPET_F32, (hipStream_t)stream));
}

struct LowrankWs { size_t dp_a, dp_g, dh, dq, partial, total; int row_chunks; int64_t rows_per_chunk; };
static LowrankWs lowrank_ws(int64_t M, int feat_dim, int d_out, int tiles, int io_dtype) {
    LowrankWs w{};
    const size_t esz = io_dtype == VLPET_F32 ? 4 : 2;
    const size_t side = align256((size_t)M * 32 * tiles * esz), wide = align256((size_t)M * d_out * esz);
    size_t o = 0;
    w.dp_a = o; o += side; w.dp_g = o; o += side; w.dh = o; o += wide; w.dq = o; o += wide;
    const int xmax = feat_dim > d_out ? feat_dim : d_out;
    wgrad_plan(M, 4, xmax, &w.row_chunks, &w.rows_per_chunk);
    w.partial = o; o += align256(wgrad_workspace_bytes(4, tiles, xmax, w.row_chunks));
    w.total = o;
    return w;
}
extern "C" size_t vlpet_lowrank_bwd_workspace_bytes(int64_t M, int feat_dim, int d_out, int tiles, int io_dtype) {
    if (lowrank_common(M, feat_dim, d_out, tiles, io_dtype)) return 0;
    return lowrank_ws(M, feat_dim, d_out, tiles, io_dtype).total;
}

// dfe = d loss / d fe ([M, d_out]); `saved` = the forward's block.  Weight gradients (fp32, overwritten): dwd [r, feat_dim],
// dbd [r], dwu [d_out, r], dbu [d_out] and the gate's four (ignored and may be NULL when packed_g == NULL).
extern "C" int vlpet_lowrank_gate_bwd(const void* dfe, const void* feats, const void* saved, const void* packed_a,
                                      const void* packed_g, float* dwd, float* dbd, float* dwu, float* dbu,
                                      float* dwgd, float* dbgd, float* dwgu, float* dbgu, int r, int rg,
                                      void* workspace, size_t workspace_bytes, int64_t M, int feat_dim, int d_out,
                                      int tiles, int gate_residual, int io_dtype, vlpet_stream_t stream) {
    int rc = lowrank_common(M, feat_dim, d_out, tiles, io_dtype);
    if (rc) return rc;
    const bool gated = packed_g != nullptr;
    if (!dfe || !feats || !saved || !packed_a || !dwd || !dbd || !dwu || !dbu || !workspace) return VLPET_E_NULL;
    if (gated && (!dwgd || !dbgd || !dwgu || !dbgu)) return VLPET_E_NULL;
    if (r <= 0 || r > 32 * tiles || (gated && (rg <= 0 || rg > 32 * tiles))) return VLPET_E_RANK;
    if (!aligned16(dfe) || !aligned16(feats) || !aligned16(saved) || !aligned16(packed_a) || !aligned16(workspace) ||
        (gated && !aligned16(packed_g))) return VLPET_E_ALIGN;
    const LowrankWs w = lowrank_ws(M, feat_dim, d_out, tiles, io_dtype);
    if (workspace_bytes < w.total) return VLPET_E_WORKSPACE;
    uint8_t* ws = reinterpret_cast<uint8_t*>(workspace);
    const uint8_t* sv = reinterpret_cast<const uint8_t*>(saved);
    PetBwdArgs b{};
    b.dy = dfe; b.xa = feats; b.res = nullptr; b.xg = feats;
    b.dxa = nullptr; b.dxg = nullptr; b.dxg_in = nullptr;
    b.saved = saved; b.saved_stride = (int64_t)saved_stride(M, tiles, io_dtype);
    b.z_a = const_cast<uint8_t*>(sv); b.z_g = const_cast<uint8_t*>(sv) + 2 * b.saved_stride;
    b.dp_a = ws + w.dp_a; b.dp_g = ws + w.dp_g; b.dh = ws + w.dh; b.dq = ws + w.dq;
    b.pk_a = reinterpret_cast<const uint8_t*>(packed_a);
    b.pk_g = gated ? reinterpret_cast<const uint8_t*>(packed_g) : b.pk_a;
    b.drop = NO_DROP;
    b.M = M; b.d = d_out; b.RT = tiles;
    b.s2 = 0.f; b.sd = 1.f; b.gs = 1.f; b.flags = PET_GATE;
    b.gm = gated ? 1.f : 0.f; b.go = gated ? (gate_residual ? 1.f : 0.f) : 1.f;
    hipError_t e = launch_pet_lowrank_bwd(b, io_dtype == VLPET_F32, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;

    WgradArgs g{};
    g.M = M; g.RT = tiles; g.row_chunks = w.row_chunks; g.rows_per_chunk = w.rows_per_chunk;
    g.partial = reinterpret_cast<float*>(ws + w.partial);
    const int ldp = 32 * tiles;
    auto job = [&](int i, const void* P, const void* X, int xw, float* out, int ldo, int transposed, int out_rows,
                   float* csx, float* csp) {
        WgradJob& J = g.job[i];
        J.P = P; J.ldp = ldp; J.pcols = ldp;
        J.X = X; J.ldx = xw; J.xcols = xw;
        J.drop = NO_DROP; J.has_drop = 0;
        J.scale = 1.f; J.out = out; J.ldo = ldo; J.transposed = transposed; J.out_rows = out_rows;
        J.colsum_x = csx; J.colsum_p = csp;
    };
    job(0, b.dp_a, feats, feat_dim, dwd, feat_dim, 0, r, nullptr, dbd);     // dWd[c,k] = sum_m dpre[m,c] feats[m,k]
    job(1, b.z_a, b.dh, d_out, dwu, r, 1, r, dbu, nullptr);                  // dWu[f,c] = sum_m dh[m,f] z[m,c]
    g.njobs = 2;
    if (gated) {
        job(2, b.dp_g, feats, feat_dim, dwgd, feat_dim, 0, rg, nullptr, dbgd);
        job(3, b.z_g, b.dq, d_out, dwgu, rg, 1, rg, dbgu, nullptr);
        g.njobs = 4;
    }
    return herr(launch_wgrad(g, io_dtype == VLPET_F32, (hipStream_t)stream));
}

// out = LayerNorm(y) * gamma + beta + r   (visual_projector_layer_norm, then the position / order-embedding term:
// src/modeling_bart.py:298-299, 324-325).  Backward: vlpet_sublayer_tail_bwd with h_save = y, p = 0 (its dx1 is d/dy;
// d/dr is dout itself).
extern "C" int vlpet_norm_residual_fwd(const void* y, const void* r, const float* gamma, const float* beta, void* out,
                                       float* mean, float* rstd, int64_t M, int d, float eps, int io_dtype,
                                       vlpet_stream_t stream);

// ---- K5 sublayer tail -------------------------------------------------------------------------------------
static int tail_common(int64_t M, int d, float p, int io_dtype) {
    if (M <= 0 || d <= 0 || d % 8 != 0) return VLPET_E_SHAPE;
    if (!(p >= 0.f && p < 1.f)) return VLPET_E_SHAPE;
    if (!dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    const int pieces = d / (io_dtype == VLPET_F32 ? 4 : 8);
    if (pieces > 8 * 64) return VLPET_E_SHAPE;
    return 0;
}
static uint32_t tail_thr(float p) {
    double t = (double)p * 65536.0 + 0.5;
    if (t > 65535.0) t = 65535.0;
    return (uint32_t)t;
}

extern "C" int vlpet_sublayer_tail_partials(int64_t M) { return M > 0 ? tail_blocks(M) : 0; }

// LayerNorm backward from the normalised rows (K4 saves xhat and rstd, not the pre-norm rows): the tail's backward kernel
// with `h` = xhat.  dx [M, d] = d/d(pre-norm); dgb_partials as in vlpet_sublayer_tail_bwd (or NULL).
extern "C" int vlpet_layernorm_bwd_xhat(const void* dout, const void* xhat, const float* rstd, const float* gamma, void* dx,
                                        float* dgb_partials, int64_t M, int d, int io_dtype, vlpet_stream_t stream) {
    int rc = tail_common(M, d, 0.f, io_dtype);
    if (rc) return rc;
    if (!dout || !xhat || !rstd || !gamma || !dx) return VLPET_E_NULL;
    if (!aligned16(dout) || !aligned16(xhat) || !aligned16(dx)) return VLPET_E_ALIGN;
    TailArgs a{};
    a.out = const_cast<void*>(dout); a.h = const_cast<void*>(xhat); a.mean = nullptr; a.rstd = const_cast<float*>(rstd);
    a.gamma = gamma; a.x1 = dx; a.y = nullptr; a.dgb = dgb_partials; a.M = M; a.d = d; a.thr = 0; a.keep_scale = 1.f;
    a.norm = 1; a.h_xhat = 1;
    return herr(launch_tail(a, io_dtype == VLPET_F32, true, (hipStream_t)stream));
}

// T5's RMS norm (my_transformers/modeling_t5.py:235-252: x * rsqrt(mean(x^2) + eps) * weight) as one pass each way on the tail
// kernels: forward out = rmsnorm(x) * gamma, rstd [M] saved; backward dx from (dout, x, rstd, gamma) + the dgamma partials.
extern "C" int vlpet_rmsnorm_fwd(const void* x, const float* gamma, void* out, float* rstd, int64_t M, int d, float eps,
                                 int io_dtype, vlpet_stream_t stream) {
    int rc = tail_common(M, d, 0.f, io_dtype);
    if (rc) return rc;
    if (!x || !gamma || !out || !rstd) return VLPET_E_NULL;
    if (!aligned16(x) || !aligned16(out)) return VLPET_E_ALIGN;
    TailArgs a{};
    a.y = nullptr; a.x1 = x; a.out = out; a.h = nullptr; a.gamma = gamma; a.beta = nullptr; a.mean = nullptr; a.rstd = rstd;
    a.keep_out = nullptr; a.dgb = nullptr; a.M = M; a.d = d; a.eps = eps; a.thr = 0; a.keep_scale = 1.f; a.norm = 1; a.rms = 1;
    return herr(launch_tail(a, io_dtype == VLPET_F32, false, (hipStream_t)stream));
}

extern "C" int vlpet_rmsnorm_bwd(const void* dout, const void* x, const float* rstd, const float* gamma, const void* dx_in,
                                 void* dx, float* dgb_partials, int64_t M, int d, int io_dtype, vlpet_stream_t stream) {
    int rc = tail_common(M, d, 0.f, io_dtype);
    if (rc) return rc;
    if (!dout || !x || !rstd || !gamma || !dx) return VLPET_E_NULL;
    if (!aligned16(dout) || !aligned16(x) || !aligned16(dx) || (dx_in && !aligned16(dx_in))) return VLPET_E_ALIGN;
    TailArgs a{};
    a.out = const_cast<void*>(dout); a.h = const_cast<void*>(x); a.mean = nullptr; a.rstd = const_cast<float*>(rstd);
    a.gamma = gamma; a.x1 = dx; a.y = nullptr; a.dgb = dgb_partials; a.M = M; a.d = d; a.thr = 0; a.keep_scale = 1.f;
    a.norm = 1; a.rms = 1; a.dres = dx_in;
    return herr(launch_tail(a, io_dtype == VLPET_F32, true, (hipStream_t)stream));
}

// T5's residual tail and the NEXT sublayer's T5LayerNorm in one pass (round 5):  sum = x1 + dropout(y);  normed = rmsnorm(sum) * gamma_next.
// Replaces my_transformers/modeling_t5.py:408 (824) followed by :366 (782) of the next sublayer.  Backward: vlpet_rmsnorm_tail_bwd.
extern "C" int vlpet_sublayer_tail_rms_fwd(const void* y, const void* x1, const float* gamma_next, void* sum, void* normed, float* rstd,
                                           int64_t M, int d, float eps, float p, uint64_t seed, int io_dtype, vlpet_stream_t stream) {
    int rc = tail_common(M, d, p, io_dtype);
    if (rc) return rc;
    if (!y || !x1 || !gamma_next || !sum || !normed || !rstd) return VLPET_E_NULL;
    if (!aligned16(y) || !aligned16(x1) || !aligned16(sum) || !aligned16(normed)) return VLPET_E_ALIGN;
    TailArgs a{};
    a.y = y; a.x1 = x1; a.out = sum; a.out2 = normed; a.gamma2 = gamma_next; a.rstd = rstd; a.M = M; a.d = d; a.eps = eps;
    a.thr = tail_thr(p); a.keep_scale = 1.0f / (1.0f - p); a.seed = seed; a.seed_ctr = g_seed_ctr.load(); a.norm = 0;
    return herr(launch_tail(a, io_dtype == VLPET_F32, false, (hipStream_t)stream));
}
// ... and its backward: d_sum = rmsnorm'(d_normed) (+ dsum_in, the gradient the sum's other readers parked);  dx1 = d_sum;
// dy = d_sum * keep / (1 - p) (written when p > 0; with p = 0 dy == dx1).  One pass instead of the norm's backward + the tail's.
extern "C" int vlpet_rmsnorm_tail_bwd(const void* d_normed, const void* sum, const float* rstd, const float* gamma_next, const void* dsum_in,
                                      void* dx1, void* dy, float* dgb_partials, int64_t M, int d, float p, uint64_t seed, int io_dtype,
                                      vlpet_stream_t stream) {
    int rc = tail_common(M, d, p, io_dtype);
    if (rc) return rc;
    if (!d_normed || !sum || !rstd || !gamma_next || !dx1) return VLPET_E_NULL;
    const uint32_t thr = tail_thr(p);
    if (thr && !dy) return VLPET_E_NULL;
    if (!aligned16(d_normed) || !aligned16(sum) || !aligned16(dx1) || (dy && !aligned16(dy)) || (dsum_in && !aligned16(dsum_in))) return VLPET_E_ALIGN;
    TailArgs a{};
    a.out = const_cast<void*>(d_normed); a.h = const_cast<void*>(sum); a.mean = nullptr; a.rstd = const_cast<float*>(rstd);
    a.gamma = gamma_next; a.x1 = dx1; a.y = dy; a.dgb = dgb_partials; a.M = M; a.d = d; a.thr = thr; a.keep_scale = 1.0f / (1.0f - p);
    a.seed = seed; a.seed_ctr = g_seed_ctr.load(); a.norm = 1; a.rms = 1; a.dres = dsum_in;
    return herr(launch_tail(a, io_dtype == VLPET_F32, true, (hipStream_t)stream));
}

// Column sums of x [M, n] in fp32 (OVERWRITTEN): the gradient of a trainable bias (dy.sum(0)).  workspace: at least
// vlpet_sublayer_tail_partials(M) * n floats.  n % 16 == 0.
extern "C" int vlpet_colsum(const void* x, int64_t M, int n, float* workspace, float* out, int io_dtype, vlpet_stream_t stream) {
    if (M <= 0 || n <= 0 || n % 16 != 0) return VLPET_E_SHAPE;
    if (!dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    if (n / (io_dtype == VLPET_F32 ? 4 : 8) > 8 * 64) return VLPET_E_SHAPE;
    if (!x || !workspace || !out) return VLPET_E_NULL;
    if (!aligned16(x)) return VLPET_E_ALIGN;
    return herr(launch_colsum(x, M, n, workspace, out, io_dtype == VLPET_F32, (hipStream_t)stream));
}

extern "C" int vlpet_colsum_partial(const void* x, int64_t M, int n, float* workspace, int io_dtype, vlpet_stream_t stream) {
    if (M <= 0 || n <= 0 || n % 16 != 0) return VLPET_E_SHAPE;
    if (!dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    if (n / (io_dtype == VLPET_F32 ? 4 : 8) > 8 * 64) return VLPET_E_SHAPE;
    if (!x || !workspace) return VLPET_E_NULL;
    if (!aligned16(x)) return VLPET_E_ALIGN;
    return herr(launch_colsum_partial(x, M, n, workspace, io_dtype == VLPET_F32, (hipStream_t)stream));
}

// Deferred weight-gradient finalize passes (csrc/wgrad.hip finalize_flush): see include/vlpet_hip.h
extern "C" int vlpet_finalize_defer(int on) { return finalize_defer(on); }
extern "C" int vlpet_finalize_pending(void) { return finalize_pending(); }
extern "C" int vlpet_finalize_discard(void) { finalize_discard(); return 0; }
extern "C" int vlpet_finalize_flush(vlpet_stream_t stream) { return herr(finalize_flush((hipStream_t)stream)); }

extern "C" int vlpet_reduce_batch(const float* const* partials, float* const* out0, float* const* out1, const int* n_partials,
                                  const int* d, int n_jobs, vlpet_stream_t stream) {
    if (n_jobs <= 0) return n_jobs == 0 ? 0 : VLPET_E_SHAPE;
    if (!partials || !out0 || !out1 || !n_partials || !d) return VLPET_E_NULL;
    for (int k0 = 0; k0 < n_jobs; k0 += VLPET_REDUCE_BATCH) {
        ReduceBatch b{};
        int max_d = 0;
        b.n = n_jobs - k0 < VLPET_REDUCE_BATCH ? n_jobs - k0 : VLPET_REDUCE_BATCH;
        for (int k = 0; k < b.n; ++k) {
            const int j = k0 + k;
            if (!partials[j] || (!out0[j] && !out1[j])) return VLPET_E_NULL;
            if (n_partials[j] <= 0 || d[j] <= 0) return VLPET_E_SHAPE;
            b.j[k] = ReduceJob{partials[j], out0[j], out1[j], n_partials[j], d[j]};
            if (d[j] > max_d) max_d = d[j];
        }
        if (int rc = herr(launch_tail_reduce_batch(b, max_d, (hipStream_t)stream))) return rc;
    }
    return 0;
}

extern "C" int vlpet_sublayer_tail_reduce(const float* dgb_partials, int n_partials, int d, float* dgamma, float* dbeta,
                                          vlpet_stream_t stream) {
    if (n_partials <= 0 || d <= 0) return VLPET_E_SHAPE;
    if (!dgb_partials || (!dgamma && !dbeta)) return VLPET_E_NULL;
    return herr(launch_tail_reduce(dgb_partials, n_partials, d, dgamma, dbeta, (hipStream_t)stream));
}

extern "C" int vlpet_sublayer_tail_fwd(const void* y, const void* x1, const float* gamma, const float* beta, void* out,
                                       void* h_save, float* mean, float* rstd, uint8_t* keep_out, int64_t M, int d,
                                       float eps, float p, uint64_t seed, int norm_mode, int io_dtype,
                                       vlpet_stream_t stream) {
    int rc = tail_common(M, d, p, io_dtype);
    if (rc) return rc;
    if (!y || !x1 || !out) return VLPET_E_NULL;
    if (norm_mode && (!gamma || !mean || !rstd)) return VLPET_E_NULL;
    if (norm_mode != 0 && norm_mode != 1) return VLPET_E_SHAPE;
    if (!aligned16(y) || !aligned16(x1) || !aligned16(out) || (h_save && !aligned16(h_save))) return VLPET_E_ALIGN;
    TailArgs a{};
    a.y = y; a.x1 = x1; a.out = out; a.h = h_save; a.gamma = gamma; a.beta = beta; a.mean = mean; a.rstd = rstd;
    a.keep_out = keep_out; a.dgb = nullptr; a.M = M; a.d = d; a.eps = eps; a.thr = tail_thr(p);
    a.keep_scale = 1.0f / (1.0f - p); a.seed = seed; a.seed_ctr = g_seed_ctr.load(); a.norm = norm_mode;
    return herr(launch_tail(a, io_dtype == VLPET_F32, false, (hipStream_t)stream));
}

extern "C" int vlpet_norm_residual_fwd(const void* y, const void* r, const float* gamma, const float* beta, void* out,
                                       float* mean, float* rstd, int64_t M, int d, float eps, int io_dtype,
                                       vlpet_stream_t stream) {
    int rc = tail_common(M, d, 0.f, io_dtype);
    if (rc) return rc;
    if (!y || !r || !out || !gamma || !mean || !rstd) return VLPET_E_NULL;
    if (!aligned16(y) || !aligned16(r) || !aligned16(out)) return VLPET_E_ALIGN;
    TailArgs a{};
    a.y = y; a.x1 = r; a.out = out; a.h = nullptr; a.gamma = gamma; a.beta = beta; a.mean = mean; a.rstd = rstd;
    a.keep_out = nullptr; a.dgb = nullptr; a.M = M; a.d = d; a.eps = eps; a.thr = 0; a.keep_scale = 1.f; a.seed = 0; a.seed_ctr = nullptr;
    a.norm = 1; a.post = 1;
    return herr(launch_tail(a, io_dtype == VLPET_F32, false, (hipStream_t)stream));
}

extern "C" int vlpet_sublayer_tail_bwd(const void* dout, const void* h_save, const float* mean, const float* rstd,
                                       const float* gamma, void* dx1, void* dy, float* dgb_partials, int64_t M, int d,
                                       float p, uint64_t seed, int norm_mode, int io_dtype, vlpet_stream_t stream) {
    int rc = tail_common(M, d, p, io_dtype);
    if (rc) return rc;
    if (norm_mode != 0 && norm_mode != 1) return VLPET_E_SHAPE;
    const uint32_t thr = tail_thr(p);
    // dx1 may be NULL for the plain residual tail under dropout (norm_mode 0, p > 0): there d/dx1 IS dout, and a caller that hands dout on
    // itself saves the copy (a third of the pass's traffic; T5 runs 60 such tails per step)
    if (!dout || (!dx1 && (norm_mode != 0 || thr == 0))) return VLPET_E_NULL;
    if (norm_mode && (!h_save || !mean || !rstd || !gamma)) return VLPET_E_NULL;
    if (thr && !dy) return VLPET_E_NULL;
    if (!aligned16(dout) || (dx1 && !aligned16(dx1)) || (dy && !aligned16(dy)) || (h_save && !aligned16(h_save))) return VLPET_E_ALIGN;
    TailArgs a{};
    a.out = const_cast<void*>(dout); a.h = const_cast<void*>(h_save); a.mean = const_cast<float*>(mean);
    a.rstd = const_cast<float*>(rstd); a.gamma = gamma; a.beta = nullptr; a.x1 = dx1; a.y = dy; a.keep_out = nullptr;
    a.dgb = dgb_partials; a.M = M; a.d = d; a.eps = 0.f; a.thr = thr; a.keep_scale = 1.0f / (1.0f - p); a.seed = seed; a.seed_ctr = g_seed_ctr.load();
    a.norm = norm_mode;
    return herr(launch_tail(a, io_dtype == VLPET_F32, true, (hipStream_t)stream));
}

// The same backward from the LayerNorm OUTPUT rows (`out_save` = what vlpet_sublayer_tail_fwd wrote to `out`) instead of the pre-norm
// sum: xhat = (out - beta) / gamma.  With it the forward is called with h_save = NULL and moves 3 row tensors instead of 4.
extern "C" int vlpet_sublayer_tail_bwd_out(const void* dout, const void* out_save, const float* rstd, const float* gamma,
                                           const float* beta, void* dx1, void* dy, float* dgb_partials, int64_t M, int d,
                                           float p, uint64_t seed, int io_dtype, vlpet_stream_t stream) {
    int rc = tail_common(M, d, p, io_dtype);
    if (rc) return rc;
    if (!dout || !dx1 || !out_save || !rstd || !gamma) return VLPET_E_NULL;
    const uint32_t thr = tail_thr(p);
    if (thr && !dy) return VLPET_E_NULL;
    if (!aligned16(dout) || !aligned16(dx1) || (dy && !aligned16(dy)) || !aligned16(out_save)) return VLPET_E_ALIGN;
    TailArgs a{};
    a.out = const_cast<void*>(dout); a.h = const_cast<void*>(out_save); a.mean = nullptr;
    a.rstd = const_cast<float*>(rstd); a.gamma = gamma; a.beta = beta; a.x1 = dx1; a.y = dy; a.keep_out = nullptr;
    a.dgb = dgb_partials; a.M = M; a.d = d; a.eps = 0.f; a.thr = thr; a.keep_scale = 1.0f / (1.0f - p); a.seed = seed; a.seed_ctr = g_seed_ctr.load();
    a.norm = 1; a.h_out = 1;
    return herr(launch_tail(a, io_dtype == VLPET_F32, true, (hipStream_t)stream));
}

static int attn_common(int B, int H, int Lq, int Lk, float p) {
    if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || Lq > VLPET_ATTN_MAX_LEN || Lk > VLPET_ATTN_MAX_LEN) return VLPET_E_SHAPE;
    if ((int64_t)B * H > 0x7fffffffLL) return VLPET_E_SHAPE;
    if (!(p >= 0.0f && p < 1.0f)) return VLPET_E_SHAPE;
    return 0;
}
static uint32_t attn_thr(float p) {
    double t = (double)p * 4294967296.0 + 0.5;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}

static int attn_ld_ok(int H, int ld_q, int ld_kv) {
    return ld_q >= H * 64 && ld_kv >= H * 64 && ld_q % 8 == 0 && ld_kv % 8 == 0;
}

// ... with separate row strides for k / dk (ld_k) and v / dv (ld_v): k may be a column block of a wider buffer (the decoder layers' fused
// key projection of the encoder output) while v keeps its own width
extern "C" int vlpet_attn_fwd_kv(const void* q, const void* k, const void* v, const uint8_t* key_mask, const float* bias, void* o,
                                 float* lse, uint8_t* keep_out, int B, int H, int Lq, int Lk, int ld_q, int ld_k, int ld_v, int causal,
                                 float scale, float p, uint64_t seed, vlpet_stream_t stream) {
    const int ld_kv = ld_k;
    int rc = attn_common(B, H, Lq, Lk, p);
    if (rc) return rc;
    if (!attn_ld_ok(H, ld_q, ld_kv) || !attn_ld_ok(H, ld_q, ld_v) || !(scale != 0.f)) return VLPET_E_SHAPE;
    if (!q || !k || !v || !o || !lse) return VLPET_E_NULL;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(o) || (bias && !aligned16(bias))) return VLPET_E_ALIGN;
    AttnArgs a{};
    a.ld_q = ld_q; a.ld_kv = ld_kv; a.ld_v = ld_v; a.bias = bias; a.bias_t = nullptr;
    a.q = (const __bf16*)q; a.k = (const __bf16*)k; a.v = (const __bf16*)v; a.o = (__bf16*)o; a.lse = lse;
    a.key_mask = key_mask; a.keep_out = keep_out; a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.causal = causal ? 1 : 0;
    a.scale = scale; a.thr = attn_thr(p); a.inv_keep = a.thr ? 1.0f / (1.0f - p) : 1.0f; a.seed = seed; a.seed_ctr = g_seed_ctr.load();
    return herr(launch_attn(a, false, (hipStream_t)stream));
}
extern "C" int vlpet_attn_fwd_bias(const void* q, const void* k, const void* v, const uint8_t* key_mask, const float* bias, void* o,
                                   float* lse, uint8_t* keep_out, int B, int H, int Lq, int Lk, int ld_q, int ld_kv, int causal,
                                   float scale, float p, uint64_t seed, vlpet_stream_t stream) {
    return vlpet_attn_fwd_kv(q, k, v, key_mask, bias, o, lse, keep_out, B, H, Lq, Lk, ld_q, ld_kv, ld_kv, causal, scale, p, seed, stream);
}

extern "C" int vlpet_attn_fwd_ld(const void* q, const void* k, const void* v, const uint8_t* key_mask, void* o, float* lse,
                                 uint8_t* keep_out, int B, int H, int Lq, int Lk, int ld_q, int ld_kv, int causal, float scale,
                                 float p, uint64_t seed, vlpet_stream_t stream) {
    return vlpet_attn_fwd_bias(q, k, v, key_mask, nullptr, o, lse, keep_out, B, H, Lq, Lk, ld_q, ld_kv, causal, scale, p, seed, stream);
}

extern "C" int vlpet_attn_fwd(const void* q, const void* k, const void* v, const uint8_t* key_mask, void* o, float* lse,
                              uint8_t* keep_out, int B, int H, int Lq, int Lk, int causal, float scale, float p, uint64_t seed,
                              vlpet_stream_t stream) {
    return vlpet_attn_fwd_ld(q, k, v, key_mask, o, lse, keep_out, B, H, Lq, Lk, H * 64, H * 64, causal, scale, p, seed, stream);
}

extern "C" int vlpet_attn_bwd_kv(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                                 const uint8_t* key_mask, const float* bias, const float* bias_t, void* dq, void* dk, void* dv,
                                 int B, int H, int Lq, int Lk, int ld_q, int ld_k, int ld_v, int causal, float scale, float p, uint64_t seed,
                                 vlpet_stream_t stream) {
    const int ld_kv = ld_k;
    int rc = attn_common(B, H, Lq, Lk, p);
    if (rc) return rc;
    if (!attn_ld_ok(H, ld_q, ld_kv) || !attn_ld_ok(H, ld_q, ld_v) || !(scale != 0.f)) return VLPET_E_SHAPE;
    if (!q || !k || !v || !o || !dout || !lse || !dq || !dk || !dv || ((bias != nullptr) != (bias_t != nullptr))) return VLPET_E_NULL;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(o) || !aligned16(dout) || !aligned16(dq) ||
        !aligned16(dk) || !aligned16(dv) || (bias && (!aligned16(bias) || !aligned16(bias_t)))) return VLPET_E_ALIGN;
    AttnArgs a{};
    a.ld_q = ld_q; a.ld_kv = ld_kv; a.ld_v = ld_v; a.bias = bias; a.bias_t = bias_t;
    a.q = (const __bf16*)q; a.k = (const __bf16*)k; a.v = (const __bf16*)v; a.o = (__bf16*)const_cast<void*>(o);
    a.lse = const_cast<float*>(lse); a.dout = (const __bf16*)dout; a.dq = (__bf16*)dq; a.dk = (__bf16*)dk; a.dv = (__bf16*)dv;
    a.key_mask = key_mask; a.keep_out = nullptr; a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.causal = causal ? 1 : 0;
    a.scale = scale; a.thr = attn_thr(p); a.inv_keep = a.thr ? 1.0f / (1.0f - p) : 1.0f; a.seed = seed; a.seed_ctr = g_seed_ctr.load();
    return herr(launch_attn(a, true, (hipStream_t)stream));
}
extern "C" int vlpet_attn_bwd_bias(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                                   const uint8_t* key_mask, const float* bias, const float* bias_t, void* dq, void* dk, void* dv,
                                   int B, int H, int Lq, int Lk, int ld_q, int ld_kv, int causal, float scale, float p, uint64_t seed,
                                   vlpet_stream_t stream) {
    return vlpet_attn_bwd_kv(q, k, v, o, dout, lse, key_mask, bias, bias_t, dq, dk, dv, B, H, Lq, Lk, ld_q, ld_kv, ld_kv, causal, scale, p, seed,
                             stream);
}

extern "C" int vlpet_attn_bwd_ld(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                                 const uint8_t* key_mask, void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk, int ld_q,
                                 int ld_kv, int causal, float scale, float p, uint64_t seed, vlpet_stream_t stream) {
    return vlpet_attn_bwd_bias(q, k, v, o, dout, lse, key_mask, nullptr, nullptr, dq, dk, dv, B, H, Lq, Lk, ld_q, ld_kv, causal, scale, p, seed,
                               stream);
}

extern "C" int vlpet_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                              const uint8_t* key_mask, void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk, int causal,
                              float scale, float p, uint64_t seed, vlpet_stream_t stream) {
    return vlpet_attn_bwd_ld(q, k, v, o, dout, lse, key_mask, dq, dk, dv, B, H, Lq, Lk, H * 64, H * 64, causal, scale, p, seed, stream);
}

static int ce_common(int64_t N, int V, int ld, int io_dtype) {
    if (N <= 0 || V <= 0 || ld < V || (ld & 7) != 0) return VLPET_E_SHAPE;
    if (io_dtype != VLPET_F32 && io_dtype != VLPET_BF16) return VLPET_E_DTYPE;
    return 0;
}

extern "C" int vlpet_ce_loss_fwd_checked(const void* logits, const int64_t* labels, float* loss, float* lse, unsigned int* bad_count,
                                         int64_t N, int V, int ld, int io_dtype, vlpet_stream_t stream) {
    int rc = ce_common(N, V, ld, io_dtype);
    if (rc) return rc;
    if (!logits || !labels || !loss || !lse) return VLPET_E_NULL;
    if (!aligned16(logits)) return VLPET_E_ALIGN;
    CeArgs a{};
    a.logits = logits; a.labels = labels; a.loss = loss; a.lse = lse; a.dloss = nullptr; a.dlogits = nullptr;
    a.N = N; a.V = V; a.ld = ld; a.bad = bad_count;
    return herr(launch_ce(a, false, io_dtype == VLPET_F32, (hipStream_t)stream));
}
extern "C" int vlpet_ce_loss_fwd(const void* logits, const int64_t* labels, float* loss, float* lse, int64_t N, int V, int ld,
                                 int io_dtype, vlpet_stream_t stream) {
    return vlpet_ce_loss_fwd_checked(logits, labels, loss, lse, nullptr, N, V, ld, io_dtype, stream);
}

extern "C" int vlpet_ce_loss_bwd(const void* logits, const int64_t* labels, const float* lse, const float* dloss, void* dlogits,
                                 int64_t N, int V, int ld, int io_dtype, vlpet_stream_t stream) {
    int rc = ce_common(N, V, ld, io_dtype);
    if (rc) return rc;
    if (!logits || !labels || !lse || !dloss || !dlogits) return VLPET_E_NULL;
    if (!aligned16(logits) || !aligned16(dlogits)) return VLPET_E_ALIGN;
    CeArgs a{};
    a.logits = logits; a.labels = labels; a.loss = nullptr; a.lse = const_cast<float*>(lse); a.dloss = dloss; a.dlogits = dlogits;
    a.N = N; a.V = V; a.ld = ld;
    return herr(launch_ce(a, true, io_dtype == VLPET_F32, (hipStream_t)stream));
}

static int act_common(int64_t n, int act, float p, int io_dtype) {
    if (n <= 0 || (n & 7) != 0) return VLPET_E_SHAPE;
    if (act != VLPET_ACT_GELU && act != VLPET_ACT_GELU_NEW && act != VLPET_ACT_RELU) return VLPET_E_SHAPE;
    if (!(p >= 0.0f && p < 1.0f)) return VLPET_E_SHAPE;
    if (io_dtype != VLPET_F32 && io_dtype != VLPET_BF16) return VLPET_E_DTYPE;
    return 0;
}

extern "C" int vlpet_act_dropout_fwd(const void* x, void* out, uint8_t* keep_out, int64_t n, int act, float p, uint64_t seed,
                                     int io_dtype, vlpet_stream_t stream) {
    int rc = act_common(n, act, p, io_dtype);
    if (rc) return rc;
    if (!x || !out) return VLPET_E_NULL;
    if (!aligned16(x) || !aligned16(out)) return VLPET_E_ALIGN;
    ActDropArgs a{};
    a.x = x; a.dy = nullptr; a.out = out; a.keep_out = keep_out; a.n = n; a.act = act; a.thr = tail_thr(p);
    a.keep_scale = a.thr ? 1.0f / (1.0f - p) : 1.0f; a.seed = seed; a.seed_ctr = g_seed_ctr.load();
    return herr(launch_act_dropout(a, false, io_dtype == VLPET_F32, (hipStream_t)stream));
}

// x [B, La + Lv, d] = dropout(cat([a [B, La, d], v [B, Lv, d]], dim = 1), p): the joint encoder's input assembly (src/modeling_bart.py:804-820)
extern "C" int vlpet_concat_dropout_fwd(const void* a, const void* v, void* x, int64_t B, int La, int Lv, int d, float p, uint64_t seed,
                                        int io_dtype, vlpet_stream_t stream) {
    if (!a || !v || !x) return VLPET_E_NULL;
    if (B <= 0 || La <= 0 || Lv <= 0 || d <= 0 || d % 8 != 0 || !(p >= 0.f && p < 1.f)) return VLPET_E_SHAPE;
    if (!dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    if (!aligned16(a) || !aligned16(v) || !aligned16(x)) return VLPET_E_ALIGN;
    const uint32_t thr = tail_thr(p);
    return herr(launch_cat_dropout(a, v, x, B, La, Lv, d, thr, thr ? 1.0f / (1.0f - p) : 1.0f, seed, g_seed_ctr.load(), false,
                                   io_dtype == VLPET_F32, (hipStream_t)stream));
}
// ... backward: dx [B, La + Lv, d] -> da [B, La, d], dv [B, Lv, d] (either NULL: not wanted), the mask regenerated from (p, seed)
extern "C" int vlpet_concat_dropout_bwd(const void* dx, void* da, void* dv, int64_t B, int La, int Lv, int d, float p, uint64_t seed,
                                        int io_dtype, vlpet_stream_t stream) {
    if (!dx || (!da && !dv)) return VLPET_E_NULL;
    if (B <= 0 || La <= 0 || Lv <= 0 || d <= 0 || d % 8 != 0 || !(p >= 0.f && p < 1.f)) return VLPET_E_SHAPE;
    if (!dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    if (!aligned16(dx) || (da && !aligned16(da)) || (dv && !aligned16(dv))) return VLPET_E_ALIGN;
    const uint32_t thr = tail_thr(p);
    return herr(launch_cat_dropout(da, dv, const_cast<void*>(dx), B, La, Lv, d, thr, thr ? 1.0f / (1.0f - p) : 1.0f, seed, g_seed_ctr.load(), true,
                                   io_dtype == VLPET_F32, (hipStream_t)stream));
}

extern "C" int vlpet_act_dropout_bwd(const void* dy, const void* x, void* dx, int64_t n, int act, float p, uint64_t seed,
                                     int io_dtype, vlpet_stream_t stream) {
    int rc = act_common(n, act, p, io_dtype);
    if (rc) return rc;
    if (!dy || !x || !dx) return VLPET_E_NULL;
    if (!aligned16(dy) || !aligned16(x) || !aligned16(dx)) return VLPET_E_ALIGN;
    ActDropArgs a{};
    a.x = x; a.dy = dy; a.out = dx; a.keep_out = nullptr; a.n = n; a.act = act; a.thr = tail_thr(p);
    a.keep_scale = a.thr ? 1.0f / (1.0f - p) : 1.0f; a.seed = seed; a.seed_ctr = g_seed_ctr.load();
    return herr(launch_act_dropout(a, true, io_dtype == VLPET_F32, (hipStream_t)stream));
}

// ---- Downsample ---------------------------------------------------------------------------------------------
extern "C" int vlpet_downsample_fwd(const void* x, void* out, int64_t n_images, int s_in, int s_out, int dim,
                                    int in_dtype, int out_dtype, vlpet_stream_t stream) {
    if (!x || !out) return VLPET_E_NULL;
    if (n_images <= 0 || s_in <= 0 || s_out <= 0 || s_out > s_in || dim <= 0 || dim % 8 != 0) return VLPET_E_SHAPE;
    if (!dtype_ok(in_dtype) || !dtype_ok(out_dtype)) return VLPET_E_DTYPE;
    if (!aligned16(x) || !aligned16(out)) return VLPET_E_ALIGN;
    PoolArgs a;
    a.x = x; a.out = out; a.n_img = n_images; a.s_in = s_in; a.s_out = s_out; a.dim = dim;
    return herr(launch_downsample(a, in_dtype == VLPET_F32, out_dtype == VLPET_F32, (hipStream_t)stream));
}

// ---- small / middleX / middleY gates (row kernels) ------------------------------------------------------------
static int row_common(int64_t M, int d, int io_dtype) {
    if (M <= 0 || d <= 0 || d % 8 != 0) return VLPET_E_SHAPE;
    if (!dtype_ok(io_dtype)) return VLPET_E_DTYPE;
    if (d / (io_dtype == VLPET_F32 ? 4 : 8) > 4 * 64) return VLPET_E_SHAPE;
    return 0;
}
static int row_launch(RowArgs& a, int op, int64_t M, int d, int io_dtype, vlpet_stream_t stream) {
    a.M = M; a.d = d;
    return herr(launch_rowgate(a, op, io_dtype == VLPET_F32, (hipStream_t)stream));
}

extern "C" int vlpet_rowgate_partials(int64_t M) { return M > 0 ? rowgate_blocks(M) : 0; }

extern "C" int vlpet_row_dot(const void* a, const void* c, const float* wa, const float* wc, float* s_out, int64_t M,
                             int d, int io_dtype, vlpet_stream_t stream) {
    int rc = row_common(M, d, io_dtype);
    if (rc) return rc;
    if (!a || !s_out) return VLPET_E_NULL;
    if (!wa && !c) return VLPET_E_NULL;                 // pair mode needs both row operands
    if (wa && c && !wc) return VLPET_E_NULL;
    if (!aligned16(a) || (c && !aligned16(c))) return VLPET_E_ALIGN;
    RowArgs r{};
    r.a = a; r.c = c; r.va = wa; r.vc = wc; r.rs = s_out;
    return row_launch(r, ROW_DOT, M, d, io_dtype, stream);
}

extern "C" int vlpet_row_affine(const void* h, const float* alpha, const float* gamma, void* y, int64_t M, int d,
                                int io_dtype, vlpet_stream_t stream) {
    int rc = row_common(M, d, io_dtype);
    if (rc) return rc;
    if (!h || !alpha || !y) return VLPET_E_NULL;
    if (!aligned16(h) || !aligned16(y)) return VLPET_E_ALIGN;
    RowArgs r{};
    r.a = h; r.ra = alpha; r.rb = gamma; r.o1 = y;
    return row_launch(r, ROW_AFFINE, M, d, io_dtype, stream);
}

extern "C" int vlpet_rowgate_bwd(const void* dy, const void* x1, const void* h, const float* alpha, const float* beta,
                                 const float* wa, const float* wc, void* dh, void* dx1, float* partials, int64_t M,
                                 int d, int io_dtype, vlpet_stream_t stream) {
    int rc = row_common(M, d, io_dtype);
    if (rc) return rc;
    if (!dy || !x1 || !h || !alpha || !beta || !wa || !wc || !dh || !dx1 || !partials) return VLPET_E_NULL;
    if (!aligned16(dy) || !aligned16(x1) || !aligned16(h) || !aligned16(dh) || !aligned16(dx1)) return VLPET_E_ALIGN;
    RowArgs r{};
    r.a = dy; r.c = x1; r.e = h; r.ra = alpha; r.rb = beta; r.va = wa; r.vc = wc; r.o1 = dh; r.o2 = dx1; r.part = partials;
    return row_launch(r, ROW_BWD, M, d, io_dtype, stream);
}

extern "C" int vlpet_vecgate_fwd(const void* h, const float* v, const float* u, void* y, int64_t M, int d, int io_dtype,
                                 vlpet_stream_t stream) {
    int rc = row_common(M, d, io_dtype);
    if (rc) return rc;
    if (!h || !v || !y) return VLPET_E_NULL;
    if (!aligned16(h) || !aligned16(y)) return VLPET_E_ALIGN;
    RowArgs r{};
    r.a = h; r.va = v; r.vc = u; r.o1 = y;
    return row_launch(r, VEC_FWD, M, d, io_dtype, stream);
}

extern "C" int vlpet_vecgate_bwd(const void* dy, const void* h, const float* v, void* dh, float* partials, int64_t M,
                                 int d, int io_dtype, vlpet_stream_t stream) {
    int rc = row_common(M, d, io_dtype);
    if (rc) return rc;
    if (!dy || !h || !v || !dh || !partials) return VLPET_E_NULL;
    if (!aligned16(dy) || !aligned16(h) || !aligned16(dh)) return VLPET_E_ALIGN;
    RowArgs r{};
    r.a = dy; r.c = h; r.va = v; r.o1 = dh; r.part = partials;
    return row_launch(r, VEC_BWD, M, d, io_dtype, stream);
}

// ---- fused clip + AdamW over the flat trainable buffer ------------------------------------------------------
extern "C" int vlpet_optim_blocks(int64_t n) { return n > 0 ? optim_blocks(n) : 0; }

extern "C" int vlpet_grad_sumsq(const float* g, int64_t n, float* partials, vlpet_stream_t stream) {
    if (!g || !partials) return VLPET_E_NULL;
    if (n <= 0) return VLPET_E_SHAPE;
    if (!aligned16(g)) return VLPET_E_ALIGN;
    return herr(launch_sumsq(g, n, partials, (hipStream_t)stream));
}

extern "C" int vlpet_adamw_step(float* p, float* g, float* m, float* v, const uint8_t* decay_mask, int64_t n,
                                const float* partials, int n_partials, float max_norm, float grad_scale, float lr,
                                float beta1, float beta2, float eps, float weight_decay, int step, int variant,
                                int zero_grad, float* norm_out, vlpet_stream_t stream) {
    if (!p || !g || !m || !v || !partials) return VLPET_E_NULL;
    if (n <= 0 || n_partials <= 0 || step <= 0 || (variant != 0 && variant != 1)) return VLPET_E_SHAPE;
    if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v) || (decay_mask && ((uintptr_t)decay_mask & 3)))
        return VLPET_E_ALIGN;
    AdamwArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.decay = decay_mask; a.n = n; a.partials = partials; a.n_partials = n_partials;
    a.max_norm = max_norm; a.grad_scale = grad_scale; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.weight_decay = weight_decay;
    a.bias_c1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bias_c2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    a.decay_first = variant == 1; a.eps_scaled = variant == 1; a.zero_grad = zero_grad; a.norm_out = norm_out;
    a.slice_of = nullptr; a.slice_bc = nullptr;
    return herr(launch_adamw(a, (hipStream_t)stream));
}

extern "C" int vlpet_adamw_step_sliced(float* p, float* g, float* m, float* v, const uint8_t* decay_mask, int64_t n,
                                       const float* partials, int n_partials, float max_norm, float grad_scale, float lr,
                                       float beta1, float beta2, float eps, float weight_decay, const int32_t* slice_of,
                                       const float* slice_bc, int variant, int zero_grad, float* norm_out,
                                       vlpet_stream_t stream) {
    if (!p || !g || !m || !v || !partials || !slice_of || !slice_bc) return VLPET_E_NULL;
    if (n <= 0 || n_partials <= 0 || (variant != 0 && variant != 1)) return VLPET_E_SHAPE;
    if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v) || !aligned16(slice_of) ||
        (decay_mask && ((uintptr_t)decay_mask & 3)))
        return VLPET_E_ALIGN;
    AdamwArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.decay = decay_mask; a.n = n; a.partials = partials; a.n_partials = n_partials;
    a.max_norm = max_norm; a.grad_scale = grad_scale; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.weight_decay = weight_decay; a.bias_c1 = 1.f; a.bias_c2_sqrt = 1.f;
    a.decay_first = variant == 1; a.eps_scaled = variant == 1; a.zero_grad = zero_grad; a.norm_out = norm_out;
    a.slice_of = slice_of; a.slice_bc = slice_bc;
    return herr(launch_adamw(a, (hipStream_t)stream));
}

