/* vlpet_hip.h -- C ABI of the MI355X-native VL-PET hot path (libvlpet_hip.so).
 *
 * The reference (HenryHZY/VL-PET) has no FFI: its boundary for this path is the Python module
 * contract between the forked transformer blocks and the PET operator packages.  Each entry
 * point below names the reference op chain it replaces (paths relative to /root/reference/src);
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch caching allocator in the
 *     shipped host code); the library allocates nothing and keeps no state between calls;
 *   - activations are row-major [M, d], contiguous, 16-byte aligned; d % 64 == 0;
 *   - io_dtype: VLPET_BF16 (performance mode) or VLPET_F32 (parity mode: bf16 hi/lo split
 *     products, fp32 accumulate); gradients of parameters are always fp32;
 *   - work is enqueued on `stream` (a hipStream_t); no call synchronises;
 *   - return value: 0 on success, a negative VLPET_E_* code for argument errors, or a positive
 *     hipError_t; nothing is thrown across the boundary.
 */
#ifndef VLPET_HIP_H
#define VLPET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VLPET_F32 0
#define VLPET_BF16 1

#define VLPET_E_SHAPE (-1)      /* d % 64 != 0, M <= 0, r <= 0 ... */
#define VLPET_E_RANK (-2)       /* bottleneck larger than 192 */
#define VLPET_E_ALIGN (-3)      /* pointer not 16-byte aligned */
#define VLPET_E_WORKSPACE (-4)  /* workspace too small */
#define VLPET_E_NULL (-5)       /* required pointer is NULL */
#define VLPET_E_DTYPE (-6)

/* gate modes (config flags of the reference) */
#define VLPET_GATE_NONE 0       /* adapter only                                              */
#define VLPET_GATE_MUL 1        /* use_encoder_adapter_gating_large_x_lowrank:  h * sigmoid  */
#define VLPET_GATE_ADD 2        /* ... + use_encoder_adapter_gating_add:         h + sigmoid */

typedef void* vlpet_stream_t;   /* hipStream_t */

int vlpet_version(void);
const char* vlpet_error_string(int code);

/* Number of 32-wide bottleneck tiles the kernels are instantiated for: 1 (r<=32), 3 (r<=96),
 * 6 (r<=192); -2 (VLPET_E_RANK) above.  Ranks are zero-padded to 32*tiles inside the pack. */
int vlpet_rank_tiles(int r);

/* ---- weight packing -----------------------------------------------------------------------
 * Re-pack one (down, up) projection pair into MFMA fragment order (+ cast / hi-lo split).  Call
 * once per optimizer step per pair; forward and backward share the result.
 *   wd_heads[n_heads]: blocks [r/n_heads, d] of the down weight -- the reference keeps the
 *       multi-head down projection as a ModuleList of N_h Linears
 *       (my_transformers/modeling_bart.py:1044-1051); adapters / LoRA pass n_heads = 1;
 *   bd_heads[n_heads]: bias blocks or NULL (LoRA);  wu [d, r];  bu [d] or NULL;
 *   param_dtype: dtype of the parameter tensors (VLPET_F32 / VLPET_BF16).
 * `tiles` may exceed vlpet_rank_tiles(r) (e.g. to give adapter and gate the same tile count). */
size_t vlpet_packed_bytes(int tiles, int d, int io_dtype);
int vlpet_pack_pair(const void* const* wd_heads, const void* const* bd_heads, int n_heads,
                    const void* wu, const void* bu, int r, int d, int tiles,
                    int param_dtype, int io_dtype, void* packed, vlpet_stream_t stream);

/* n pairs of ONE geometry (same n_heads, r, d, tiles, dtypes; every pair with or without biases alike) in as few launches as
 * possible (8 pairs per launch): wd_heads_flat / bd_heads_flat hold n * n_heads pointers (pair-major), wu / bu / packed n
 * pointers.  What a trainer calls once per optimizer step for all its adapters instead of ~30 single packs. */
int vlpet_pack_pairs(int n, const void* const* wd_heads_flat, const void* const* bd_heads_flat, int n_heads,
                     const void* const* wu, const void* const* bu, int r, int d, int tiles,
                     int param_dtype, int io_dtype, void* const* packed, vlpet_stream_t stream);

/* ---- K1: encoder granularity-controlled adapter + gate ------------------------------------
 * out = ( x2_scale*x2 + delta_scale*up(gelu_new(down(x2))) ) (*|+) sigmoid(up_g(gelu_new(down_g(x1)))) * gate_scale
 * replaces my_transformers/modeling_bart.py:1147-1155,1195-1209,1256-1257 (attention sublayer),
 * :1270-1278,1317-1325,1372-1373 (FFN sublayer); my_transformers/modeling_t5.py:366-379,385-390,
 * 405-406 and :782-795,801-806,821-822.  gate_mode VLPET_GATE_NONE ignores x1 / packed_g. */
int vlpet_adapter_gate_fwd(const void* x1, const void* x2, const void* packed_a, const void* packed_g,
                           void* out, int64_t M, int d, int tiles, int gate_mode,
                           float delta_scale, float x2_scale, float gate_scale,
                           int io_dtype, vlpet_stream_t stream);

/* Training form of the gated forward: also leaves the bottleneck activations the backward needs
 * (z = gelu_new(pre) and gelu_new'(pre) of both chains, four [M, 32*tiles] tensors of the IO dtype --
 * what torch.autograd keeps for the same lines of the reference, in a quarter of the bytes) in `saved`
 * (vlpet_saved_bytes(M, tiles, io_dtype) bytes, 16-byte aligned).  vlpet_adapter_gate_bwd_saved then
 * skips the recompute of the two down projections: it reads x1 not at all and x2 once.  VLPET_GATE_NONE is accepted too
 * (adapter chain only: the first half of the block is used). */
size_t vlpet_saved_bytes(int64_t M, int tiles, int io_dtype);
int vlpet_adapter_gate_fwd_save(const void* x1, const void* x2, const void* packed_a, const void* packed_g,
                                void* out, void* saved, int64_t M, int d, int tiles, int gate_mode,
                                float delta_scale, float x2_scale, float gate_scale,
                                int io_dtype, vlpet_stream_t stream);

size_t vlpet_bwd_workspace_bytes(int64_t M, int d, int tiles, int has_gate, int io_dtype);

/* Backward of the above (torch.autograd through the same lines in the reference).  Recomputes
 * the forward intermediates from (x1, x2).  dx2 includes the residual path (x2_scale*dh); the
 * parameter gradients are fp32, OVERWRITTEN (not accumulated):
 *   dwd [r, d] (head blocks stacked), dbd [r], dwu [d, r], dbu [d]; same four for the gate with rg. */
int vlpet_adapter_gate_bwd(const void* dy, const void* x1, const void* x2,
                           const void* packed_a, const void* packed_g,
                           void* dx1, void* dx2,
                           float* dwd, float* dbd, float* dwu, float* dbu,
                           float* dwgd, float* dbgd, float* dwgu, float* dbgu,
                           int r, int rg, void* workspace, size_t workspace_bytes,
                           int64_t M, int d, int tiles, int gate_mode,
                           float delta_scale, float x2_scale, float gate_scale,
                           int io_dtype, vlpet_stream_t stream);

/* The two halves of vlpet_adapter_gate_bwd, selectable so a profiler / bench can bracket them with its
 * own events: phases bit 0 = row-parallel kernel (dx1, dx2 + side products in the workspace),
 * bit 1 = column-parallel weight gradients (reads the side products).  phases = 3 is the full call. */
int vlpet_adapter_gate_bwd_phase(int phases, const void* dy, const void* x1, const void* x2,
                                 const void* packed_a, const void* packed_g,
                                 void* dx1, void* dx2,
                                 float* dwd, float* dbd, float* dwu, float* dbu,
                                 float* dwgd, float* dbgd, float* dwgu, float* dbgu,
                                 int r, int rg, void* workspace, size_t workspace_bytes,
                                 int64_t M, int d, int tiles, int gate_mode,
                                 float delta_scale, float x2_scale, float gate_scale,
                                 int io_dtype, vlpet_stream_t stream);

/* Which form vlpet_adapter_gate_bwd_saved takes for this shape: 2 = pass 1 + the column-parallel pass of csrc/pet_cols.hip
 * (bf16, r, r_g <= 96, d % 128 == 0), 1 = pass 1 + the older column-parallel pass (csrc/pet_gate_bwd3.hip: r = 192, fp32 small M),
 * 0 = row kernel + weight-gradient kernels; < 0: bad arguments.  (What a bench labels its kernel brackets with.) */
int vlpet_adapter_gate_bwd_form(int64_t M, int d, int tiles, int io_dtype);
/* 1: that form ends in a separate finalize launch at this shape (`phases` bit 4 runs it alone); 0: it does not -- since round 6 the
 * column-parallel pass at r, r_g <= 96 and M >= 8,192 rows sums its row-chunk partials INSIDE the launch (csrc/cols_reduce.h: the workgroups of a column
 * block each take a slice once all of them have published; bounded waits, the last arriver finishes what an owner gave up, results
 * bit-identical to the two-launch form).  `phases` bit 5 (32) of the *_bwd_saved* entry points keeps the round-3 two-launch form for
 * same-box A/Bs.  The reduce-scatter's control words are zeroed by pass 1, so `phases` = 2 alone presumes that pass 1 of the same
 * backward ran on the same workspace before (as the bench's per-kernel brackets do). */
int vlpet_adapter_gate_bwd_finalize_launch(int64_t M, int d, int tiles, int io_dtype);
/* Process-wide: 0 = pass 2 of the gated K1 backward always leaves its partial slabs to a finalize launch; 1 (default) = at r <= 96 and
 * from 8,192 rows it sums them inside the launch (below that the finalize launch measured ahead).  Turn it off where other kernels run BESIDE the backward (gradient
 * collectives on their own stream overlapping it, a second process on the device): a workgroup waiting for partners that cannot
 * start keeps its CU for as long as the foreign kernel lasts (bounded: ~3 ms, then it gives up and the last arriver sums its slice --
 * results unchanged, time lost), where the two-launch form runs in two rounds.  Returns the previous setting. */
int vlpet_set_in_launch_reduce(int on);

/* Dropout seeds under graph replay (train.Trainer(graph=True): forward + backward of a step captured once with hipGraph and replayed).
 * A replayed launch repeats its kernel arguments, so the per-call `seed` values of the dropout-carrying entry points below
 * (vlpet_lora_delta_*, vlpet_sublayer_tail_*, vlpet_act_dropout_*, vlpet_attn_*) would give every step the same masks.  With a
 * counter registered here -- a 64-bit word in DEVICE memory that the caller increments once per step, outside the captured
 * region or as its first node -- every such kernel uses seed + counter * 0x9E3779B97F4A7C15 instead (one scalar load in its
 * prologue); forward and backward of a step read the same value, so regenerated masks still match.  NULL (the default) = seeds
 * are used as passed.  Process-wide; the only state the library keeps between calls.  No reference counterpart: the reference
 * draws from torch's global generator (my_transformers/modeling_bart.py:1259, lora/controller.py:66). */
int vlpet_set_seed_counter(const uint64_t* device_counter);

/* 1 for a diagnosis build (make DEBUG=1): the only builds whose kernels' experiment switches (csrc/tuning.h) can be set from
 * VLPET_* environment variables, read once at load time.  The product library (0) reads nothing from the environment. */
int vlpet_debug_build(void);

/* Test instrument for the kernels whose workgroups hand data to each other inside a launch (K4's statistics exchange): occupies CUs
 * from another stream so that such a launch does NOT have the GPU to itself.  `workgroups` workgroups of 64 threads, each holding
 * `lds_bytes` of LDS (<= 160 KiB: one per CU, and nothing larger than the rest fits beside it), poll the 32-bit word *release_flag
 * (device memory) until it is nonzero or max_ms milliseconds (capped at 10,000) have passed.  No reference counterpart. */
int vlpet_test_hold_cus(int workgroups, int lds_bytes, void* release_flag, int max_ms, vlpet_stream_t stream);

/* vlpet_adapter_gate_bwd_phase with the activations saved by vlpet_adapter_gate_fwd_save (x1 is
 * still an argument because the gate's down-weight gradient contracts it).
 * It runs as TWO PASSES that move every [M, d] tensor once each (csrc/pet_gate_bwd3.hip pass 1, csrc/pet_cols.hip pass 2):
 * phases bit 0 = pass 1 (row-parallel: dpre of both chains into the workspace, no [M, d] output),
 * bit 1 = pass 2 (column-parallel: dx1, dx2 AND the eight weight / bias gradients from recomputed dh / dq).  A caller that
 * needs dx1 / dx2 right after bit 0 (weight gradients on another stream) adds bit 2 to BOTH calls: the previous split
 * (bit 0: dx1, dx2 + [M, d] side products; bit 1: weight gradients from them).  phases = 3 is the full call either way.
 * For event brackets around single kernels (form 2 only): bit 3 with bit 1 = pass 2 WITHOUT the finalize launch (the row-chunk
 * partial sums stay in the workspace), bit 4 alone = the finalize launch only. */
int vlpet_adapter_gate_bwd_saved(int phases, const void* dy, const void* x1, const void* x2, const void* saved,
                                 const void* packed_a, const void* packed_g,
                                 void* dx1, void* dx2,
                                 float* dwd, float* dbd, float* dwu, float* dbu,
                                 float* dwgd, float* dbgd, float* dwgu, float* dbgu,
                                 int r, int rg, void* workspace, size_t workspace_bytes,
                                 int64_t M, int d, int tiles, int gate_mode,
                                 float delta_scale, float x2_scale, float gate_scale,
                                 int io_dtype, vlpet_stream_t stream);

/* vlpet_adapter_gate_bwd_saved with an incoming residual-stream gradient:  dx1 = dx1_in + (gradient of the gate branch).
 * x1 (the sublayer input) feeds both the gate and the sublayer tail LayerNorm(x1 + dropout(y))
 * (my_transformers/modeling_bart.py:1196, 1259-1261); handing the tail's dx1 in here replaces the elementwise add
 * autograd would launch for the two contributions.  The chain-split row kernel adds it in its epilogue (one more row
 * stream of the gate-chain wave); the other forms run one extra pass.  dx1_in: [M, d] IO dtype, must not alias dx1.
 * With phases, dx1 is complete after the call that carries bit 0 (previous split) or bit 1 (two-pass form). */
int vlpet_adapter_gate_bwd_saved_acc(int phases, const void* dy, const void* x1, const void* x2, const void* saved,
                                     const void* packed_a, const void* packed_g, const void* dx1_in, void* dx1, void* dx2,
                                     float* dwd, float* dbd, float* dwu, float* dbu,
                                     float* dwgd, float* dbgd, float* dwgu, float* dbgu,
                                     int r, int rg, void* workspace, size_t workspace_bytes,
                                     int64_t M, int d, int tiles, int gate_mode,
                                     float delta_scale, float x2_scale, float gate_scale,
                                     int io_dtype, vlpet_stream_t stream);

/* vlpet_adapter_gate_bwd_saved / _acc with the forward's OUTPUT y = gate_scale * h * g at hand (a training framework keeps it for
 * free: it is the input of the sublayer tail).  With the multiplicative gate  dq = dh * h * (1 - g) = dy * y * (1 - g), so pass 1 of
 * the two-pass forms recomputes the gate's up projection only -- the adapter chain's drops out, the row stream carries y instead of
 * x2 (same bytes).  y: [M, d] IO dtype or NULL (then exactly the two functions above); dx1_in: optional.  The additive gate ignores
 * y.  Autograd of my_transformers/modeling_bart.py:1147-1155, 1195-1209 (T5: modeling_t5.py:366-390, 782-806). */
int vlpet_adapter_gate_bwd_saved_y(int phases, const void* dy, const void* x1, const void* x2, const void* y, const void* saved,
                                   const void* packed_a, const void* packed_g, const void* dx1_in, void* dx1, void* dx2,
                                   float* dwd, float* dbd, float* dwu, float* dbu,
                                   float* dwgd, float* dbgd, float* dwgu, float* dbgu,
                                   int r, int rg, void* workspace, size_t workspace_bytes,
                                   int64_t M, int d, int tiles, int gate_mode,
                                   float delta_scale, float x2_scale, float gate_scale,
                                   int io_dtype, vlpet_stream_t stream);

/* ---- K2: parallel adapter (decoder cross-attention value path) ----------------------------
 * out = y + scale * up(gelu_new(down(x)))
 * replaces adapters/adapter_modeling.py:55-61 + adapters/adapter_controller.py:149-162 as called at
 * my_transformers/modeling_bart.py:427-430 and my_transformers/modeling_t5.py:600-603. */
int vlpet_parallel_adapter_fwd(const void* x, const void* y, const void* packed, void* out,
                               int64_t M, int d, int tiles, float scale,
                               int io_dtype, vlpet_stream_t stream);
/* dx is the adapter-branch gradient only; d/dy is dy itself and is not materialised. */
int vlpet_parallel_adapter_bwd(const void* dy, const void* x, const void* packed, void* dx,
                               float* dwd, float* dbd, float* dwu, float* dbu, int r,
                               void* workspace, size_t workspace_bytes,
                               int64_t M, int d, int tiles, float scale,
                               int io_dtype, vlpet_stream_t stream);

/* Training form of K2 (as vlpet_adapter_gate_fwd_save / _bwd_saved): the forward leaves z and gelu_new'(pre) in `saved`
 * (the first half of a vlpet_saved_bytes(M, tiles, io_dtype) block is used), the backward starts after the recompute. */
int vlpet_parallel_adapter_fwd_save(const void* x, const void* y, const void* packed, void* out, void* saved,
                                    int64_t M, int d, int tiles, float scale, int io_dtype, vlpet_stream_t stream);
int vlpet_parallel_adapter_bwd_saved(const void* dy, const void* x, const void* saved, const void* packed, void* dx,
                                     float* dwd, float* dbd, float* dwu, float* dbu, int r,
                                     void* workspace, size_t workspace_bytes, int64_t M, int d, int tiles,
                                     float scale, int io_dtype, vlpet_stream_t stream);

/* ---- K3: LoRA low-rank update on top of the frozen linear ---------------------------------
 * out = base + scaling * ((dropout(x) @ A^T) @ B^T),  base = F.linear(x, W, b) computed by the caller
 * replaces lora/controller.py:61-68 (the base GEMM at :59 stays a library GEMM).
 * Square projections only (in_features == out_features == d: q_proj / v_proj,
 * my_transformers/modeling_bart.py:767-768).  A = lora_As[task] [r, d], B = lora_Bs[task] [d, r].
 * Dropout (lora/controller.py:66, training only): p in [0, 1), p = 0 = none.  With keep_mask == NULL the mask comes from
 * the library's counter-based generator (Philox-4x32-7, the one of vlpet_sublayer_tail_*): a function of (seed, element
 * index) only, regenerated -- not stored -- by the backward, which must be given the same p and seed; keep_out (optional,
 * uint8 [M, d], 16-byte aligned) receives the 0/1 mask that was applied (parity tests).  With keep_mask != NULL
 * (uint8 [M, d], 1 = keep, 16-byte aligned) that mask is applied instead; kept elements are scaled by 1/(1-p) either way. */
int vlpet_lora_delta_fwd(const void* x, const void* base, const void* packed,
                         const uint8_t* keep_mask, float p, uint64_t seed, uint8_t* keep_out, void* out,
                         int64_t M, int d, int tiles, float scaling,
                         int io_dtype, vlpet_stream_t stream);
/* dx is the LoRA share of the input gradient (the caller adds dy @ W). */
int vlpet_lora_delta_bwd(const void* dy, const void* x, const void* packed,
                         const uint8_t* keep_mask, float p, uint64_t seed, void* dx,
                         float* da, float* db, int r,
                         void* workspace, size_t workspace_bytes,
                         int64_t M, int d, int tiles, float scaling,
                         int io_dtype, vlpet_stream_t stream);

/* Training form of K3 (as the K1 / K2 forms): the forward also leaves z = dropout(x) @ A^T  [M, 32*tiles] (IO dtype) and
 * the dropout mask it applied (1 bit per element) in `saved` (vlpet_lora_saved_bytes): the backward neither re-reads x
 * for the down projection nor regenerates the mask (x is still read once, by the weight gradient of A). */
size_t vlpet_lora_saved_bytes(int64_t M, int d, int tiles, int io_dtype);
int vlpet_lora_delta_fwd_save(const void* x, const void* base, const void* packed,
                              const uint8_t* keep_mask, float p, uint64_t seed, uint8_t* keep_out, void* out,
                              void* saved, int64_t M, int d, int tiles, float scaling,
                              int io_dtype, vlpet_stream_t stream);
/* K3 at rank r <= 8 (bf16, d % 64 == 0, d <= 1024 -- vlpet_lora_r8_applies): the same delta as a streaming row kernel without the
 * matrix cores (csrc/lora8.hip; eight multiply-adds per element are not a matrix product).  Same `packed` pair, same dropout
 * generator and masks as the entry points above; `saved` NULL = inference form, else the saved block of
 * vlpet_lora_saved_bytes(M, d, 1, io_dtype) as vlpet_lora_delta_fwd_save leaves it, so vlpet_lora_delta_bwd_saved takes it as is.
 * Replaces lora/controller.py:56-70 at lora_dim <= 8.  VLPET_E_SHAPE when the form does not apply. */
int vlpet_lora_r8_applies(int64_t M, int d, int r, int io_dtype);
int vlpet_lora_delta_fwd_r8(const void* x, const void* base, const void* packed, const uint8_t* keep_mask, float p,
                            uint64_t seed, uint8_t* keep_out, void* out, void* saved, int64_t M, int d, int r,
                            float scaling, int io_dtype, vlpet_stream_t stream);
int vlpet_lora_delta_bwd_saved(const void* dy, const void* x, const void* saved, const void* packed,
                               const uint8_t* keep_mask, float p, uint64_t seed, void* dx,
                               float* da, float* db, int r,
                               void* workspace, size_t workspace_bytes,
                               int64_t M, int d, int tiles, float scaling,
                               int io_dtype, vlpet_stream_t stream);

/* ---- K4: visual-feature projection -----------------------------------------------------------
 * out = LayerNorm(feats @ W^T + b) * gamma + beta (+ R)
 * replaces the feat_embedding branch of VisualEmbedding.forward (src/modeling_bart.py:91-110,157;
 * T5: src/modeling_t5.py:56-66 with rms = 1, beta = NULL).  R [M, d_out] (IO dtype, may be NULL) is added
 * after the norm: the caller passes the position branch + order embeddings (:162-183).
 * feats [M, F], F % 64 == 0; d_out in {64, 128, 768}.  xhat [M, d_out] / rstd [M] (may be NULL) are the
 * normalised activations and 1/sigma the backward needs.  W is packed once per optimizer step. */
size_t vlpet_visproj_packed_bytes(int d_out, int feat_dim, int io_dtype);
int vlpet_visproj_pack(const void* w, const void* b, int d_out, int feat_dim, int param_dtype, int io_dtype,
                       void* packed, vlpet_stream_t stream);
int vlpet_visproj_fwd(const void* feats, const void* packed, const float* gamma, const float* beta,
                      const void* r, void* out, void* xhat, float* rstd,
                      int64_t M, int feat_dim, int d_out, float eps, int rms,
                      int io_dtype, vlpet_stream_t stream);
/* The same forward as a tiled GEMM whose column tiles exchange the LayerNorm statistics (round 5; csrc/visproj_gemm.hip): bf16 IO,
 * d_out a multiple of 256 (<= 1024), feat_dim a multiple of 64.  w_io [d_out, feat_dim] = the weight in the IO dtype, row-major (no
 * pack); bias [d_out] fp32 or NULL; mean [M] optional; xhat is REQUIRED when d_out > 256.  workspace:
 * vlpet_visproj_gemm_workspace_bytes (0 = shape not supported by this form); the CALLER zeroes it once -- every call leaves the
 * exchange area (bytes 256 .. 256 + vlpet_visproj_gemm_exchange_bytes) zeroed, so it is reused across calls without a memset; calls
 * sharing one workspace must be ordered (one stream).
 * The column tiles of a row block wait for each other's statistics, which presumes that they run at the same time; where the GPU is
 * shared (a collective on another stream, a second process) a workgroup may give up on a partner (bounded polling, about a second).
 * The call repairs that itself (round 6): it launches a second, normally empty, kernel that re-normalises exactly the tiles that
 * gave up from the complete per-tile statistics and cleans the exchange area -- the caller NEVER receives rows normalised with
 * partial statistics.  The first 32-bit word of the workspace is a sticky status (nonzero = it has happened since the caller last
 * cleared the word; informational), word 1 counts the tiles that gave up; bytes 64..119: wall-clock stamps of workgroup 0.
 * replaces: src/modeling_bart.py:91-110,157 (T5: src/modeling_t5.py:56-66). */
size_t vlpet_visproj_gemm_workspace_bytes(int64_t M, int feat_dim, int d_out);
size_t vlpet_visproj_gemm_exchange_bytes(int d_out);
int vlpet_visproj_fwd_gemm(const void* feats, const void* w_io, const float* bias, const float* gamma, const float* beta,
                           const void* r, void* out, void* xhat, float* rstd, float* mean, void* workspace,
                           size_t workspace_bytes, int64_t M, int feat_dim, int d_out, float eps, int rms, int io_dtype,
                           vlpet_stream_t stream);
/* ... with the LDS ring form (bits 0-7 of `form`: 1: 64 input features per stage / 2 slots, 2: 32 / 4, 3: 32 / 3, 4-6: the same with
 * spread requests; 0: default) and the rows per workgroup (128 / 192 / 256; 0: by shape) forced: measurement and parity of the
 * non-default forms.  Bits 8-12 of `form`, if nonzero: log2 of the polls before a wave gives up on a partner, 31 = a single poll, 30 = every workgroup gives up without polling (tests of the repair path). */
int vlpet_visproj_fwd_gemm_cfg(const void* feats, const void* w_io, const float* bias, const float* gamma, const float* beta,
                               const void* r, void* out, void* xhat, float* rstd, float* mean, void* workspace,
                               size_t workspace_bytes, int64_t M, int feat_dim, int d_out, float eps, int rms, int io_dtype,
                               int form, int rows_per_workgroup, vlpet_stream_t stream);
/* out = srcs[0] + ... + srcs[n - 1] over len elements of the IO dtype (len % 8 == 0; srcs = HOST array of n device pointers; out may be
 * srcs[0]): fp32 accumulation, one rounding per launch of up to eight sources.  The gradient of a tensor several consumers read -- the
 * encoder output under every decoder layer's cross-attention -- which autograd would sum pairwise (n - 1 passes of three row units).
 * replaces: autograd's accumulation for my_transformers/modeling_bart.py:2300-2330 (every decoder layer takes encoder_hidden_states). */
int vlpet_sum_n(const void* const* srcs, int n, void* out, int64_t len, int io_dtype, vlpet_stream_t stream);
/* K4's position / order branch, the R the feature projection adds behind its norm:
 *     R[b, n, :] = norm_p( W_p [x1, x2, y1, y2, area] + b_p ) + img_order_embedding[img_id[b, n]] + obj_order_embedding[V - 1 - obj_id[b, n]]
 * with area = (y2 - y1) (x2 - x1); norm_p = LayerNorm (BART) or T5LayerNorm (rms = 1: no mean, beta ignored).  One launch; rows are
 * (b, n) pairs, M = B N.  pos [M, 4] fp32; w [d, 5], b, gamma, beta [d] fp32; tables [n_img, d] / [obj_rows, d] in fp32 or bf16
 * (both NULL: no order embeddings -- config.use_vis_order_embedding off); ids int64 [B, N] (batch stride N) or [1, N] (stride 0), NULL =
 * the reference's defaults (image 0, object n); ids outside their table are clamped.  out [M, d] in the IO dtype (the fp32 value
 * rounded once).  d a multiple of 256 (<= 1024), n_img <= 4: vlpet_vispos_applies.
 * replaces: src/modeling_bart.py:129-141,162-183 (T5: src/modeling_t5.py:109-122,143-165). */
int vlpet_vispos_applies(int d, int n_img);
int vlpet_vispos_fwd(const float* pos, const float* w, const float* b, const float* gamma, const float* beta,
                     const void* img_table, int img_table_dtype, int n_img, const int64_t* img_ids, int64_t img_ids_bstride,
                     const void* obj_table, int obj_table_dtype, int64_t obj_rows, const int64_t* obj_ids, int64_t obj_ids_bstride,
                     void* out, int64_t M, int N, int d, float eps, int rms, int io_dtype, vlpet_stream_t stream);
/* ... and its backward from dout = the gradient of the visual embedding's output (= dR), [M, d] IO dtype.  WRITES dw [d, 5], db, dgamma,
 * dbeta [d] (dbeta NULL with rms), dimg [n_img, d] (NULL: no image-order table); two launches (per-workgroup column partials, then
 * their sum in a fixed order: deterministic).  No gradient for the object-order table (the shared token table, frozen in every launch
 * script).  workspace: vlpet_vispos_bwd_workspace_bytes.
 * replaces: autograd of the lines above. */
size_t vlpet_vispos_bwd_workspace_bytes(int64_t M, int d, int n_img);
int vlpet_vispos_bwd(const void* dout, const float* pos, const float* w, const float* b, const float* gamma,
                     int n_img, const int64_t* img_ids, int64_t img_ids_bstride,
                     float* dw, float* db, float* dgamma, float* dbeta, float* dimg,
                     void* workspace, size_t workspace_bytes, int64_t M, int N, int d, float eps, int rms, int io_dtype,
                     vlpet_stream_t stream);
/* Weight gradient of the projection:  dw [d_out, F] = dpre^T @ feats,  db [d_out] = column sums of dpre
 * (dpre = gradient w.r.t. the pre-norm activations, [M, d_out], IO dtype).  fp32, overwritten. */
size_t vlpet_visproj_wgrad_workspace_bytes(int64_t M, int feat_dim, int d_out);
/* The same per IO dtype (the bf16 form keeps fp32 split-K partials: ~100 MB at feat_dim 2048 -> 768; the fp32 path needs a few MB). */
size_t vlpet_visproj_wgrad_workspace_bytes_io(int64_t M, int feat_dim, int d_out, int io_dtype);
int vlpet_visproj_wgrad(const void* dpre, const void* feats, float* dw, float* db,
                        void* workspace, size_t workspace_bytes,
                        int64_t M, int feat_dim, int d_out, int io_dtype, vlpet_stream_t stream);

/* ---- f4: low-rank visual projector -------------------------------------------------------------
 * LowRankVisualEmbedding (src/modeling_bart.py:195-334), the feature branch:
 *     fe  = up(gelu_new(cat_i down_i(feats)))                                    (:278-284)
 *     fe *= sigmoid(gup(gelu_new(gdown(feats))))  [+ fe  with gate_residual]      (:286-295)
 *     out = LayerNorm(fe) + r        (r = position branch + order embeddings)     (:298-299, 324-325)
 * as the K1 kernels in a rectangular form: both chains read the same [M, feat_dim] feature rows
 * (one LDS tile per stage), the down phase runs over feat_dim, the up phase over d_out, no residual
 * term, and -- the features being data -- no input gradients.  feat_dim % 64 == 0, d_out % 64 == 0,
 * tiles 1 or 3 (r, r_g <= 96, both chains padded to the same tile count).
 *
 * vlpet_lowrank_pack replaces the module's raw parameters (visual_projector_multihead_down[i].weight
 * [r/n_heads, feat_dim] / .bias, visual_projector_multihead_up.weight [d_out, r] / .bias; same for the
 * gate pair) by one fragment-ordered buffer per pair: a square pair pack at d_out (up side + biases)
 * followed by a down-only pack at feat_dim.  Re-pack after every optimizer step, as for K1.
 *
 * packed_g == NULL selects the ungated projector (the gate chain then runs with multiplier 0, offset 1
 * on packed_a, and the gate's weight gradients are neither computed nor written).
 * saved: vlpet_saved_bytes(M, tiles, io_dtype) bytes, written by the forward, read by the backward.
 * Weight gradients: fp32, overwritten; dwd [r, feat_dim], dbd [r], dwu [d_out, r], dbu [d_out] (gate alike). */
size_t vlpet_lowrank_packed_bytes(int tiles, int feat_dim, int d_out, int io_dtype);
int vlpet_lowrank_pack(const void* const* wd_heads, const void* const* bd_heads, int n_heads,
                       const void* wu, const void* bu, int r, int feat_dim, int d_out, int tiles,
                       int param_dtype, int io_dtype, void* packed, vlpet_stream_t stream);
int vlpet_lowrank_gate_fwd(const void* feats, const void* packed_a, const void* packed_g, void* fe,
                           void* saved, int64_t M, int feat_dim, int d_out, int tiles, int gate_residual,
                           int io_dtype, vlpet_stream_t stream);
size_t vlpet_lowrank_bwd_workspace_bytes(int64_t M, int feat_dim, int d_out, int tiles, int io_dtype);
int vlpet_lowrank_gate_bwd(const void* dfe, const void* feats, const void* saved, const void* packed_a,
                           const void* packed_g, float* dwd, float* dbd, float* dwu, float* dbu,
                           float* dwgd, float* dbgd, float* dwgu, float* dbgu, int r, int rg,
                           void* workspace, size_t workspace_bytes, int64_t M, int feat_dim, int d_out,
                           int tiles, int gate_residual, int io_dtype, vlpet_stream_t stream);
/* out = LayerNorm(y) * gamma + beta + r, one pass (the K5 kernel with the residual joining after the norm;
 * src/modeling_bart.py:298-299 then :324-325).  mean / rstd [M] are saved for the backward, which is
 * vlpet_sublayer_tail_bwd(dout, h_save = y, ..., p = 0, norm_mode = 1): its dx1 is d/dy; d/dr is dout. */
int vlpet_norm_residual_fwd(const void* y, const void* r, const float* gamma, const float* beta, void* out,
                            float* mean, float* rstd, int64_t M, int d, float eps, int io_dtype,
                            vlpet_stream_t stream);

/* ---- K5: sublayer tail ----------------------------------------------------------------------
 * The step right after K1 (and after every decoder sublayer):
 *     norm_mode 1:  out = LayerNorm(x1 + dropout(y)) * gamma + beta
 *                   (my_transformers/modeling_bart.py:1259-1261, 1375-1377)
 *     norm_mode 0:  out = x1 + dropout(y)        (my_transformers/modeling_t5.py:408, 824)
 * one pass over [M, d] forward, one pass backward.  d % 8 == 0, d <= 4096 (bf16) / 2048 (fp32).
 * Dropout: counter-based (Philox-4x32) mask that is a function of (seed, element index) only; the
 * backward regenerates it from the same seed -- nothing is stored.  p = 0 switches it off.
 *   h_save [M, d] (IO dtype), mean [M], rstd [M]: saved for the backward (norm_mode 1; h_save may
 *       be NULL for inference);  keep_out: optional [M, d] uint8 export of the mask (tests).
 * Backward: dx1 [M, d] (may be NULL for norm_mode 0 with p > 0: there d/dx1 is dout itself and the
 * caller hands dout on -- no copy); dy [M, d] only when p > 0 (with p = 0, dy == dx1: pass NULL and
 * alias); dgb_partials [vlpet_sublayer_tail_partials(M)][2][d] fp32 = per-workgroup partial
 * sums of (dgamma, dbeta), or NULL when the LayerNorm is frozen. */
int vlpet_sublayer_tail_partials(int64_t M);
int vlpet_sublayer_tail_fwd(const void* y, const void* x1, const float* gamma, const float* beta, void* out,
                            void* h_save, float* mean, float* rstd, uint8_t* keep_out, int64_t M, int d,
                            float eps, float p, uint64_t seed, int norm_mode, int io_dtype,
                            vlpet_stream_t stream);
int vlpet_sublayer_tail_bwd(const void* dout, const void* h_save, const float* mean, const float* rstd,
                            const float* gamma, void* dx1, void* dy, float* dgb_partials, int64_t M, int d,
                            float p, uint64_t seed, int norm_mode, int io_dtype, vlpet_stream_t stream);
/* The same backward from the LayerNorm OUTPUT rows (out_save = the `out` of vlpet_sublayer_tail_fwd, which the next sublayer keeps
 * anyway) instead of the pre-norm sum: xhat = (out - beta) / gamma (0 where gamma is 0).  The forward is then called with
 * h_save = NULL and writes one row tensor instead of two (norm_mode 1 only; mean is not needed). */
int vlpet_sublayer_tail_bwd_out(const void* dout, const void* out_save, const float* rstd, const float* gamma, const float* beta,
                                void* dx1, void* dy, float* dgb_partials, int64_t M, int d,
                                float p, uint64_t seed, int io_dtype, vlpet_stream_t stream);
/* The LayerNorm parameter gradients of that backward (autograd of `self_attn_layer_norm` / `final_layer_norm`,
 * my_transformers/modeling_bart.py:1261, 1377): dgamma [d], dbeta [d] (fp32, OVERWRITTEN; either may be NULL) = the sum of
 * the n_partials = vlpet_sublayer_tail_partials(M) rows of dgb_partials.  One launch; the caller may point dgamma / dbeta
 * straight at the parameters' slots of its flat gradient buffer. */
int vlpet_sublayer_tail_reduce(const float* dgb_partials, int n_partials, int d, float* dgamma, float* dbeta,
                               vlpet_stream_t stream);
/* ---- deferred finalize of the weight gradients ------------------------------------------------
 * Every backward entry point above that produces weight gradients ends with a "finalize" launch: the sum of its row-chunk partial
 * blocks (in the call's workspace) into dW / db.  A training step holds one per adapter call and nobody reads a weight gradient
 * before the optimizer, so a caller may collect them:
 *   vlpet_finalize_defer(1)   from now on backward calls made by THIS host thread queue that launch instead of issuing it (returns
 *                             the previous setting); the queue itself is per process -- a framework's autograd worker thread
 *                             fills it, the thread that called backward() flushes it.  The call's workspace and its gradient outputs must stay allocated -- and the
 *                             outputs are incomplete -- until the flush.
 *   vlpet_finalize_flush(s)   launches everything queued on stream s (which must be ordered after the queued calls' streams), up to
 *                             16 calls per launch; same arithmetic and order of additions per call: bit-identical results.
 *   vlpet_finalize_pending()  queue length;  vlpet_finalize_discard()  drops the queue (error paths).
 * (The reference leaves this to autograd's per-parameter AccumulateGrad: src/trainer_base.py / multitask.py:682-695.) */
int vlpet_finalize_defer(int on);
int vlpet_finalize_pending(void);
int vlpet_finalize_discard(void);
int vlpet_finalize_flush(vlpet_stream_t stream);

/* Deferred form of the two reductions above, for a trainer (nobody reads a parameter gradient before the optimizer step):
 * vlpet_colsum_partial runs only the first pass of vlpet_colsum (workspace [vlpet_sublayer_tail_partials(M)][n]); vlpet_reduce_batch
 * sums n_jobs partial blocks in ONE launch per 96 jobs: job j reduces partials[j] viewed as [n_partials[j]][2 d[j]] into out0[j] [d]
 * and out1[j] [d] (OVERWRITTEN; either may be NULL) -- a LayerNorm's (dgamma, dbeta) with d = d_model, or a bias gradient of width
 * n as (out, out + n / 2) with d = n / 2.  Host arrays; the reference's counterpart is autograd's per-parameter sum(0) + accumulate
 * (trainer_base.py:339-344 unfreezes every bias in the LoRA runs). */
int vlpet_colsum_partial(const void* x, int64_t M, int n, float* workspace, int io_dtype, vlpet_stream_t stream);
int vlpet_reduce_batch(const float* const* partials, float* const* out0, float* const* out1, const int* n_partials,
                       const int* d, int n_jobs, vlpet_stream_t stream);
/* LayerNorm backward from the NORMALISED rows: xhat [M, d] (IO dtype) and rstd [M] as vlpet_visproj_fwd saves them
 * (autograd of `feat_embedding`'s LayerNorm, src/modeling_bart.py:157, 171).  dx [M, d] = gradient of the pre-norm rows,
 * dgb_partials as above (NULL when the LayerNorm is frozen); one pass over [M, d]. */
/* T5LayerNorm (my_transformers/modeling_t5.py:235-252; the RMS norm in front of every T5 sublayer, :366, :782, and the final
 * norms): out = x * rsqrt(mean(x^2) + eps) * gamma, statistics in fp32, one pass each way over [M, d].  rstd [M] is saved by the
 * forward; the backward writes dx [M, d] and, when dgb_partials is non-NULL, the partial sums of dgamma in the layout of
 * vlpet_sublayer_tail_bwd (reduce with vlpet_sublayer_tail_reduce(partials, n, d, dgamma, NULL)).  dx_in: optional [M, d], added
 * to dx -- the gradient of the other reader of x (a pre-norm sublayer's residual add, :408, and K1's gate input), so that
 * autograd has one gradient for the stream instead of two to add; may alias dx. */
int vlpet_rmsnorm_fwd(const void* x, const float* gamma, void* out, float* rstd, int64_t M, int d, float eps, int io_dtype,
                      vlpet_stream_t stream);
int vlpet_rmsnorm_bwd(const void* dout, const void* x, const float* rstd, const float* gamma, const void* dx_in, void* dx,
                      float* dgb_partials, int64_t M, int d, int io_dtype, vlpet_stream_t stream);
/* T5's residual tail + the NEXT sublayer's T5LayerNorm in one pass (round 5): sum = x1 + dropout(y) (modeling_t5.py:408 / 824), normed =
 * rmsnorm(sum) * gamma_next (:366 / 782 of the next sublayer; the statistic is taken on the sum rounded to the IO dtype, as a second
 * launch would have read it); rstd [M] saved for the backward. */
int vlpet_sublayer_tail_rms_fwd(const void* y, const void* x1, const float* gamma_next, void* sum, void* normed, float* rstd,
                                int64_t M, int d, float eps, float p, uint64_t seed, int io_dtype, vlpet_stream_t stream);
/* ... backward: d_sum = rmsnorm'(d_normed) + dsum_in (optional: what the sum's other readers parked); dx1 = d_sum; dy = d_sum * keep / (1 - p)
 * (p > 0; else dy == dx1, pass NULL); dgb_partials as in vlpet_rmsnorm_bwd. */
int vlpet_rmsnorm_tail_bwd(const void* d_normed, const void* sum, const float* rstd, const float* gamma_next, const void* dsum_in,
                           void* dx1, void* dy, float* dgb_partials, int64_t M, int d, float p, uint64_t seed, int io_dtype,
                           vlpet_stream_t stream);
/* out [n] (fp32, OVERWRITTEN) = column sums of x [M, n] (IO dtype): the gradient of a trainable Linear bias -- the reference's LoRA
 * runs train every bias (src/trainer_base.py `unfreeze "bias"` rule; autograd's dy.sum(0) of my_transformers/modeling_bart.py:791-811).
 * workspace: vlpet_sublayer_tail_partials(M) * n floats.  n % 16 == 0, n <= 4096 (bf16) / 2048 (fp32).  Two launches. */
int vlpet_colsum(const void* x, int64_t M, int n, float* workspace, float* out, int io_dtype, vlpet_stream_t stream);
int vlpet_layernorm_bwd_xhat(const void* dout, const void* xhat, const float* rstd, const float* gamma, void* dx,
                             float* dgb_partials, int64_t M, int d, int io_dtype, vlpet_stream_t stream);

/* ---- FFN activation + dropout (the backbone step between fc1 and fc2 of the sublayers K1 / K5 close) ----------
 * out = dropout(act(x), p) as one pass; backward dx = dy * mask / (1 - p) * act'(x), mask regenerated from the seed
 * (same counter-based generator as the sublayer tail; group of 8 consecutive elements = one Philox call).
 * Replaces `activation_fn(fc1(x))` + `F.dropout(..., p=activation_dropout)` (my_transformers/modeling_bart.py:1264-1265,
 * 1750-1756) and `F.relu` + `nn.Dropout` of T5DenseReluDense (my_transformers/modeling_t5.py:262-265).
 *   act: VLPET_ACT_GELU (erf form), VLPET_ACT_GELU_NEW (tanh form), VLPET_ACT_RELU;  n: elements, multiple of 8;
 *   keep_out: optional [n] uint8 export of the mask (tests), NULL otherwise. */
#define VLPET_ACT_GELU 0
#define VLPET_ACT_GELU_NEW 1
#define VLPET_ACT_RELU 2
int vlpet_act_dropout_fwd(const void* x, void* out, uint8_t* keep_out, int64_t n, int act, float p, uint64_t seed,
                          int io_dtype, vlpet_stream_t stream);
int vlpet_act_dropout_bwd(const void* dy, const void* x, void* dx, int64_t n, int act, float p, uint64_t seed,
                          int io_dtype, vlpet_stream_t stream);
/* The joint encoder's input assembly: x [B, La + Lv, d] = dropout(cat([a [B, La, d], v [B, Lv, d]], dim = 1), p) in one pass, and its
 * backward dx -> da, dv (either NULL: not wanted) with the mask regenerated from (p, seed) -- instead of a concatenation pass, a dropout
 * pass that also writes a byte mask, a masked-scale pass and one copy per slice of the gradient.
 * replaces: src/modeling_bart.py:804-820 (inputs_embeds = torch.cat([inputs_embeds, vis_embeds], dim=1); F.dropout), T5: src/modeling_t5.py:263, 300. */
int vlpet_concat_dropout_fwd(const void* a, const void* v, void* x, int64_t B, int La, int Lv, int d, float p, uint64_t seed,
                             int io_dtype, vlpet_stream_t stream);
int vlpet_concat_dropout_bwd(const void* dx, void* da, void* dv, int64_t B, int La, int Lv, int d, float p, uint64_t seed,
                             int io_dtype, vlpet_stream_t stream);

/* ---- LM-head cross entropy (the loss of the training step; src/modeling_bart.py:1574-1586, src/modeling_t5.py:680-694) --
 * CrossEntropyLoss(ignore_index=-100, reduction='none') on logits [N, V] stored with a row stride of ld elements
 * (ld % 8 == 0, ld >= V; the caller pads the LM-head weight so that rows are 16-byte aligned; columns [V, ld) are ignored
 * and get zero gradient).  labels: int64 [N], negative = ignored token (loss 0, zero gradient row).
 * Forward: loss [N] fp32 and lse [N] fp32 (row log-sum-exp, kept for the backward).  Backward:
 * dlogits[n, v] = dloss[n] * (exp(logits[n, v] - lse[n]) - [v == labels[n]]) in the IO dtype, [N, ld]. */
int vlpet_ce_loss_fwd(const void* logits, const int64_t* labels, float* loss, float* lse, int64_t N, int V, int ld,
                      int io_dtype, vlpet_stream_t stream);
int vlpet_ce_loss_bwd(const void* logits, const int64_t* labels, const float* lse, const float* dloss, void* dlogits,
                      int64_t N, int V, int ld, int io_dtype, vlpet_stream_t stream);
/* The forward with a tally of suspicious labels: a label outside [0, V) is treated like ignore_index (torch's loss raises a device
 * assert for it); *bad_count (a 32-bit word in device memory, may be NULL) is incremented once per label outside [0, V) that is not
 * -100, so that the caller can notice a tokenizer / vocabulary mismatch without synchronising every step. */
int vlpet_ce_loss_fwd_checked(const void* logits, const int64_t* labels, float* loss, float* lse, unsigned int* bad_count,
                              int64_t N, int V, int ld, int io_dtype, vlpet_stream_t stream);

/* ---- Short-sequence attention of the frozen backbone (my_transformers/modeling_bart.py:283-566, BartAttention.forward) ----
 * o = dropout(softmax(scale * q k^T + mask), p) v per (batch, head), bf16, head dim 64, at most VLPET_ATTN_MAX_LEN keys and
 * queries per sequence (the encoder's 20-40 text + 36 / 72 visual tokens; decoder self / cross attention).  One
 * workgroup per (batch, head), everything on chip; HBM traffic = q, k, v read once, o written (see csrc/attn.hip).
 *   q, o, dout, dq: [B, Lq, H, 64] contiguous;  k, v, dk, dv: [B, Lk, H, 64] contiguous (i.e. the [B, L, H*64] projection
 *   outputs as they are: no head transpose);  key_mask: [B, Lk] uint8, 1 = attend, or NULL;  causal: key j visible to query
 *   i iff j <= i + (Lk - Lq);  lse: [B, H, Lq] fp32 scratch written by the forward and read by the backward;
 *   keep_out: optional [B, H, Lq, Lk] uint8 export of the dropout mask (tests), NULL otherwise.
 * Dropout: element (b, h, i, j) is kept iff a hash of (seed, element index) >= p * 2^32; the backward regenerates it.
 * A query row with no visible key yields zeros (the library path yields NaN there). */
#define VLPET_ATTN_MAX_LEN 128
int vlpet_attn_fwd(const void* q, const void* k, const void* v, const uint8_t* key_mask, void* o, float* lse, uint8_t* keep_out,
                   int B, int H, int Lq, int Lk, int causal, float scale, float p, uint64_t seed, vlpet_stream_t stream);
int vlpet_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                   const uint8_t* key_mask, void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk, int causal,
                   float scale, float p, uint64_t seed, vlpet_stream_t stream);
/* The same with explicit row strides (in elements, multiples of 8, >= H*64) for q / dq (ld_q) and k, v / dk, dv (ld_kv): the
 * columns of a fused [B, L, 3*H*64] q|k|v projection output (and of its gradient) are read / written in place, so the frozen
 * q_proj / k_proj / v_proj of a self-attention (my_transformers/modeling_bart.py:791-811) run as ONE library GEMM each way and
 * autograd has no three input gradients to sum.  o, dout and lse keep the dense layout. */
int vlpet_attn_fwd_ld(const void* q, const void* k, const void* v, const uint8_t* key_mask, void* o, float* lse, uint8_t* keep_out,
                      int B, int H, int Lq, int Lk, int ld_q, int ld_kv, int causal, float scale, float p, uint64_t seed,
                      vlpet_stream_t stream);
int vlpet_attn_bwd_ld(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                      const uint8_t* key_mask, void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk, int ld_q, int ld_kv,
                      int causal, float scale, float p, uint64_t seed, vlpet_stream_t stream);
/* The same with an additive score bias shared by the batch: softmax(scale * q k^T + bias[h] + mask) -- T5's relative position bias
 * (my_transformers/modeling_t5.py:520-560 compute_bias, added at :640-660; T5 runs with scale = 1).  bias: [H, Lqp, Lkp] fp32,
 * Lqp / Lkp = Lq / Lk rounded up to a multiple of 32, zero padded, 16-byte aligned; the backward also takes its transpose
 * bias_t [H, Lkp, Lqp] (its key-major phase reads the bias along queries).  The bias gets no gradient (a frozen embedding in every
 * script).  NULL bias = the plain entry points above. */
int vlpet_attn_fwd_bias(const void* q, const void* k, const void* v, const uint8_t* key_mask, const float* bias, void* o,
                        float* lse, uint8_t* keep_out, int B, int H, int Lq, int Lk, int ld_q, int ld_kv, int causal,
                        float scale, float p, uint64_t seed, vlpet_stream_t stream);
int vlpet_attn_bwd_bias(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                        const uint8_t* key_mask, const float* bias, const float* bias_t, void* dq, void* dk, void* dv,
                        int B, int H, int Lq, int Lk, int ld_q, int ld_kv, int causal, float scale, float p, uint64_t seed,
                        vlpet_stream_t stream);
/* ... and with separate row strides for k / dk (ld_k) and v / dv (ld_v) (round 6): the decoder layers' key projections of the encoder
 * output run as ONE GEMM [M, d] -> [M, n_layers * d] (my_transformers/modeling_bart.py:2300-2330: every layer's encoder_attn projects the
 * same encoder_hidden_states); a layer's attention reads its column block in place (ld_k = n_layers * d) and writes dk into the same block
 * of one gradient buffer, which a single dgrad GEMM with K = n_layers * d consumes; v -- behind the value-parallel adapter -- keeps ld_v = d. */
int vlpet_attn_fwd_kv(const void* q, const void* k, const void* v, const uint8_t* key_mask, const float* bias, void* o,
                      float* lse, uint8_t* keep_out, int B, int H, int Lq, int Lk, int ld_q, int ld_k, int ld_v, int causal,
                      float scale, float p, uint64_t seed, vlpet_stream_t stream);
int vlpet_attn_bwd_kv(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                      const uint8_t* key_mask, const float* bias, const float* bias_t, void* dq, void* dk, void* dv,
                      int B, int H, int Lq, int Lk, int ld_q, int ld_k, int ld_v, int causal, float scale, float p, uint64_t seed,
                      vlpet_stream_t stream);

/* ---- Downsample (the step before K4) -------------------------------------------------------
 * AdaptiveMaxPool2d(s_in x s_in -> s_out x s_out) over the token grid of x [n_images, s_in*s_in, dim]
 * -> out [n_images, s_out*s_out, dim], with the cast to the compute dtype fused (in_dtype may be
 * fp32 CLIP features, out_dtype the IO dtype).  Replaces Downsample.downsample_inputs
 * (src/modeling_bart.py:565-581); NLVR's two-image form is n_images = 2B on the same buffer.
 * dim % 8 == 0.  No parameters, no backward (the features carry no gradient). */
int vlpet_downsample_fwd(const void* x, void* out, int64_t n_images, int s_in, int s_out, int dim,
                         int in_dtype, int out_dtype, vlpet_stream_t stream);

/* ---- K1 variants: small / middleX / middleY gates -------------------------------------------
 * The other three granularity gates (my_transformers/modeling_bart.py:1210-1231, 1326-1347;
 * my_transformers/modeling_t5.py:391-403, 807-819) act on h = vlpet_adapter_gate_fwd(..., VLPET_GATE_NONE):
 *     small   : g_b   = mean_S sigmoid(w . [x1 ; h] + b)        y = (h * g_b | h + g_b) * gs
 *     middleX : g_row = sigmoid(w . (x1 + h) + b)               y = (h * g_row | h + g_row) * gs
 *     middleY :                                                 y = (h + h*z | h + 1 + z) * gs
 * The [M, d] passes are the row kernels below; the O(M) scalar algebra (sigmoid of a row scalar,
 * the sequence mean, alpha / beta of the backward) is the caller's.  d % 8 == 0, d <= 2048 (bf16) /
 * 1024 (fp32).  All per-feature vectors and per-row scalars are fp32.
 *   vlpet_row_dot     s[row] = a[row,:] . wa + c[row,:] . wc   (c, wc optional);  wa == NULL: s[row] = a[row,:] . c[row,:]
 *   vlpet_row_affine  y[row,:] = h[row,:] * alpha[row] + gamma[row]              (gamma optional)
 *   vlpet_rowgate_bwd dh = alpha[row]*dy + beta[row]*wc;  dx1 = beta[row]*wa;
 *                     partials[vlpet_rowgate_partials(M)][2][d] = per-workgroup sums of (beta*x1, beta*h) = (dwa, dwc)
 *   vlpet_vecgate_fwd y[row,:] = h[row,:] * v + u                                 (u optional)
 *   vlpet_vecgate_bwd dh = dy * v;  partials[...][2][d] = per-workgroup sums of (dy*h, dy) */
int vlpet_rowgate_partials(int64_t M);
int vlpet_row_dot(const void* a, const void* c, const float* wa, const float* wc, float* s_out, int64_t M, int d,
                  int io_dtype, vlpet_stream_t stream);
int vlpet_row_affine(const void* h, const float* alpha, const float* gamma, void* y, int64_t M, int d,
                     int io_dtype, vlpet_stream_t stream);
int vlpet_rowgate_bwd(const void* dy, const void* x1, const void* h, const float* alpha, const float* beta,
                      const float* wa, const float* wc, void* dh, void* dx1, float* partials, int64_t M, int d,
                      int io_dtype, vlpet_stream_t stream);
int vlpet_vecgate_fwd(const void* h, const float* v, const float* u, void* y, int64_t M, int d, int io_dtype,
                      vlpet_stream_t stream);
int vlpet_vecgate_bwd(const void* dy, const void* h, const float* v, void* dh, float* partials, int64_t M, int d,
                      int io_dtype, vlpet_stream_t stream);

/* ---- fused global-norm clip + AdamW over the flat trainable buffer ---------------------------
 * Replaces torch.nn.utils.clip_grad_norm_ (multitask.py:279-300) + the per-tensor AdamW step
 * (trainer_base.py:633-701) with two launches over the flat fp32 buffers p, g, m, v [n]:
 *   vlpet_grad_sumsq   partials[vlpet_optim_blocks(n)] = per-workgroup sums of g^2
 *   vlpet_adamw_step   g' = g * grad_scale * min(1, max_norm / (||g * grad_scale|| + 1e-6));  AdamW(p, g', m, v)
 * variant 0 = transformers.optimization.AdamW as the reference configures it (bias correction on, eps
 * added to sqrt(v), decoupled decay after the update); variant 1 = torch.optim.AdamW (decay first,
 * eps added to sqrt(v / (1 - beta2^t))).  decay_mask [n] uint8 (4-byte aligned): 1 where weight decay
 * applies (NULL: everywhere).  step = 1-based update count.  grad_scale folds the data-parallel average
 * (1 / world_size) in.  zero_grad != 0 clears g.  norm_out (optional device float) receives the pre-clip norm. */
int vlpet_optim_blocks(int64_t n);
int vlpet_grad_sumsq(const float* g, int64_t n, float* partials, vlpet_stream_t stream);
int vlpet_adamw_step(float* p, float* g, float* m, float* v, const uint8_t* decay_mask, int64_t n,
                     const float* partials, int n_partials, float max_norm, float grad_scale, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int step, int variant,
                     int zero_grad, float* norm_out, vlpet_stream_t stream);

/* The same with per-parameter step counts and activity: transformers.AdamW keeps state['step'] per parameter and skips a
 * parameter whose grad is None (no decay, no moment update), which is what happens to the other tasks' adapters / LoRA
 * matrices when use_single_adapter / use_single_lora is off (multitask.py:296-297 sets grads to None every step).
 * slice_of [n] int32 (16-byte aligned): index of the parameter every element belongs to;  slice_bc [n_params][2] fp32:
 * {1 - beta1^t_k, sqrt(1 - beta2^t_k)} of parameter k with ITS 1-based update count t_k, or a value <= 0 in [k][0] when
 * the parameter received no gradient this step (its elements of p, m, v are left untouched; g is still cleared). */
int vlpet_adamw_step_sliced(float* p, float* g, float* m, float* v, const uint8_t* decay_mask, int64_t n,
                            const float* partials, int n_partials, float max_norm, float grad_scale, float lr,
                            float beta1, float beta2, float eps, float weight_decay, const int32_t* slice_of,
                            const float* slice_bc, int variant, int zero_grad, float* norm_out,
                            vlpet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VLPET_HIP_H */
