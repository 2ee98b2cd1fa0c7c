// Internal kernel argument blocks and launchers (the public C ABI is include/vlpet_hip.h).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rng.h"
#include "tuning.h"

#define VLPET_MAX_HEADS 16

struct PackArgs {
    const void* wd[VLPET_MAX_HEADS];   // N_h blocks of the down weight, each [r/N_h, d]
    const void* bd[VLPET_MAX_HEADS];   // N_h bias blocks (bd[0] == nullptr -> no bias)
    const void* wu;                    // up weight [d, r]
    const void* bu;                    // up bias [d] or nullptr
    int n_heads, rows_per_head;
    int r, d, RT;                      // true rank, width, padded rank / 32
    int src_bf16;                      // parameter dtype: 0 fp32, 1 bf16
    int n_packs;                       // 4 (adapter pair) or 1 (down pack only: visual projection; biases follow it)
    uint8_t* out;                      // packed pair (see pack_geom)
};
hipError_t launch_pack_pair(const PackArgs& a, int NS, hipStream_t stream);
#define VLPET_PACK_BATCH 8
struct PackBatch { PackArgs p[VLPET_PACK_BATCH]; int n; };      // pairs of one geometry (d, RT, n_packs, dtypes)
hipError_t launch_pack_pairs(const PackBatch& b, int NS, hipStream_t stream);

// flags
#define PET_GATE 1        // chain G present: out = (res*s2 + sd*delta) (*|+) sigmoid(gate) * gs
#define PET_GATE_ADD 2    // additive gate (use_encoder_adapter_gating_add)
#define PET_ACT_IDENTITY 4  // LoRA: no nonlinearity between down and up

struct PetFwdArgs {
    const void* xa;     // chain-A input   [M,d]  (K1: x2, K2/K3: x)
    const void* res;    // residual        [M,d]  (K1: x2, K2: y = v_proj(x), K3: base linear output)
    const void* xg;     // gate input x1   [M,d]  (K1) or nullptr
    void* out;          // [M,d]
    const uint8_t* pk_a;   // packed pair, chain A
    const uint8_t* pk_g;   // packed pair, chain G (gate) or nullptr
    DropSpec drop;         // LoRA dropout on the chain-A input (rng.h): explicit mask, in-kernel generator, or none
    int64_t M;
    int d, RT;
    float s2, sd, gs;      // x2 scale, delta scale, gate scale
    int flags;
    void* save;            // optional (gated K1, training): bottleneck activations for the backward, four [M, 32*RT] IO-dtype
    int64_t save_stride;   //   tensors at save + k*save_stride bytes: z_a, gelu'(pre_a), z_g, gelu'(pre_g)
    int dbg;               // ablation bits (env VLPET_DBG; 0 in production): 1 no weight stream, 2 no row loads, 8 no stores, 16 timestamps
    unsigned long long* dbg_ts;   // [blocks][8] s_memtime stamps of wave 0 (only when dbg & 16)
    // low-rank visual projector (LowRankVisualEmbedding, src/modeling_bart.py:195-334): both chains read the SAME
    // [M, d_in] input (xa), the output is d wide; down weights come from separate down-only packs of width d_in
    int d_in;                  // 0: square form (input width = d)
    const uint8_t* pk_a_dn;    // down pack (n_packs = 1 form) of chain A at width d_in
    const uint8_t* pk_g_dn;    // same, chain G
    float gm, go;              // gate value = gm * sigmoid(.) + go  (gated: gm = 1, go = 1 with use_visual_projector_residual_connection
                               //   else 0; ungated projector: gm = 0, go = 1)
};
hipError_t launch_pet_fwd(const PetFwdArgs& a, int io_fp32, hipStream_t stream);
hipError_t launch_pet_gate_fwd(const PetFwdArgs& a, int io_fp32, hipStream_t stream);   // PET_GATE only
hipError_t launch_pet_lowrank_fwd(const PetFwdArgs& a, int io_fp32, hipStream_t stream);   // low-rank visual projector form (d_in != d)
// gated K1 forward, training form, as two passes with the weights resident in registers (pet_fwd2p.hip; bf16, d = 768)
bool k1_fwd2p_applies(const PetFwdArgs& a, int io_fp32);
bool k1_fwd2p_preferred(const PetFwdArgs& a);      // by shape: where the two-pass form measured faster than the one-kernel forward
hipError_t launch_k1_fwd2p(const PetFwdArgs& a, int passes, hipStream_t stream);     // passes: bit 0 = pass A (down), bit 1 = pass B (up)

struct PetBwdArgs {
    const void* dy;     // [M,d]
    const void* xa;     // chain-A input
    const void* res;    // residual (only read when PET_GATE: needed for h)
    const void* xg;     // gate input
    void* dxa;          // K1: d/dx2 (includes the residual path); K2/K3: d/dx of the adapter branch only
    void* dxg;          // K1: d/dx1 of the gate branch
    const void* dxg_in; // optional [M,d]: added to dxg (the residual-stream gradient of the sublayer tail: x1 feeds both the gate
                        //   and the tail, and summing here saves autograd's separate add pass); may alias dxg
    // row-major side products consumed by the weight-gradient kernel (IO dtype)
    void* z_a; void* dp_a;      // [M, 32*RT] each
    void* z_g; void* dp_g;      // [M, 32*RT] each (gate)
    void* dh; void* dq;         // [M, d] each (gate only; without gate the wgrad reads dy itself)
    const uint8_t* pk_a;
    const uint8_t* pk_g;
    DropSpec drop;          // LoRA dropout on the chain-A input (same mask as the forward: same explicit mask or same seed)
    const void* saved;      // optional (gated K1): the forward's PetFwdArgs::save block -- skips the recompute of the
    int64_t saved_stride;   //   bottleneck activations (phase 1: no x1 / second x2 read) and the z side products
    int64_t M;
    int d, RT;
    float s2, sd, gs;
    int flags;
    float gm, go;           // low-rank visual projector: gate value = gm * sigmoid(.) + go (unused otherwise)
    int fsplit;             // pass 1 of the two-pass form (pet_dz2.hip), small M: feature blocks (> 1: partial dz in dz_part, summed by a second launch)
    float* dz_part;         //   [fsplit][M][2][32*RT] fp32
    const void* y;          // optional (gated K1, multiplicative gate, saved form): the forward's OUTPUT [M,d] -- pass 1 then forms
                            //   dq = dy * y * (1 - g) and skips the adapter chain's up projection (pet_dz2.hip / pet_dz6.hip)
    unsigned* red_ctrl;     // optional: the control words of pass 2's in-launch reduce-scatter (cols_reduce.h) -- pass 1's first
    int red_words;          //   workgroup zeroes them, so no memset node sits between the passes
};
hipError_t launch_pet_bwd(const PetBwdArgs& a, int io_fp32, hipStream_t stream);
// chain-split form of the same (pet_gate_bwd2.hip): gated K1 with saved activations
bool pet_gate_bwd2_applies(const PetBwdArgs& a);
hipError_t launch_pet_gate_bwd2(const PetBwdArgs& a, int io_fp32, hipStream_t stream);
hipError_t launch_pet_lowrank_bwd(const PetBwdArgs& a, int io_fp32, hipStream_t stream);   // low-rank visual projector form: dh, dq, dpre only

// Weight gradients:  Out[c, n] = scale * sum_m P[m, c] * X[m, n]   (P skinny, X wide), plus the
// column sums of X (bias of the "up" side) and of P (bias of the "down" side).
struct WgradJob {
    const void* P; int ldp; int pcols;     // P [M, ldp], columns [0, pcols) used, pcols = 32*RT
    const void* X; int ldx; int xcols;     // X [M, ldx], xcols multiple of 64
    DropSpec drop; int has_drop;           // optional dropout mask applied to X (LoRA down grad; X must be the full [M, ldx] tensor)
    float scale;
    float* out; int ldo; int transposed;   // transposed: out[n*ldo + c] else out[c*ldo + n]
    int out_rows;                          // true rank r (rows c >= r are dropped)
    float* colsum_x;                       // [xcols] or nullptr   (scale applied)
    float* colsum_p;                       // [out_rows] or nullptr (no scale)
};
struct WgradArgs {
    WgradJob job[4];
    int njobs;
    int64_t M;
    int RT;
    int row_chunks;        // RC
    int64_t rows_per_chunk;  // multiple of 128
    float* partial;        // workspace, see wgrad_workspace_bytes
    int nslice;            // 64-column slices of the widest job (set by the launcher)
};
// partial-sum workspace of the weight gradients: per job [row_chunks][32*RT][xcols] tiles, then [row_chunks][xcols]
// column sums of X, then [row_chunks][32*RT] column sums of P
struct WgradLayout {
    int64_t off[4];      // float offset of each job's partial block
    int64_t total;       // floats
};
__host__ __device__ inline WgradLayout wgrad_layout(const WgradArgs& a) {
    WgradLayout L;
    int64_t o = 0;
    const int PR = 32 * a.RT;
    for (int j = 0; j < 4; ++j) {
        L.off[j] = o;
        if (j < a.njobs) o += (int64_t)a.row_chunks * ((int64_t)PR * a.job[j].xcols + a.job[j].xcols + PR);
    }
    L.total = o;
    return L;
}
size_t wgrad_workspace_bytes(int njobs, int RT, int xcols_max, int row_chunks);
hipError_t launch_wgrad_finalize(const WgradArgs& a, hipStream_t stream);      // sums the row-chunk partials into the outputs
// ... or queues that pass (per host thread) while finalize_defer(1) is in force; finalize_flush launches the queue, up to 16 calls per launch
int finalize_defer(int on);            // returns the previous setting
int finalize_pending();
void finalize_discard();
hipError_t finalize_flush(hipStream_t stream);
void wgrad_plan(int64_t M, int njobs, int xcols_max, int* row_chunks, int64_t* rows_per_chunk);
hipError_t launch_wgrad(const WgradArgs& a, int io_fp32, hipStream_t stream);
// Backward without a gate (K2, adapter-only K1, K3 without dropout) in two passes (pet_cols_ng.hip; bf16, r <= 96, saved activations)
bool ng_two_pass_applies(const PetBwdArgs& a, int io_fp32);
void ng_cols_plan(int64_t M, int d, int* row_chunks, int64_t* rows_per_chunk);
hipError_t launch_ng_two_pass(const PetBwdArgs& b, const WgradArgs& g, int passes, hipStream_t stream);
// K4 weight gradient as a tiled split-K GEMM (visproj_wgrad.hip; bf16, F % 256 == 0, d_out % 384 == 0)
bool k4_wgrad2_applies(int64_t M, int F, int d_out, int io_fp32);
size_t k4_wgrad2_workspace_bytes(int64_t M, int F, int d_out);
hipError_t launch_k4_wgrad2(const void* dpre, const void* feats, float* dw, float* db, void* workspace, int64_t M, int F, int d_out,
                            hipStream_t stream);

// Two-pass gated K1 backward (pet_gate_bwd3.hip): pass 1 writes dpre only, pass 2 (column-parallel) recomputes dh / dq per
// feature block and produces the input gradients and the four weight gradients.
bool pet_gate_bwd3_applies(const PetBwdArgs& a);
void gate_bwd3_plan(int64_t M, int d, int io_fp32, int* row_chunks, int64_t* rows_per_chunk, int* GS, int* NG);
hipError_t launch_pet_gate_dz(const PetBwdArgs& a, int io_fp32, hipStream_t stream);
// the same pass with the stage's features split over the two waves of a row group (pet_dz2.hip; bf16, r <= 96)
bool k1_dz2_applies(const PetBwdArgs& a, int io_fp32);
hipError_t launch_k1_dz2(const PetBwdArgs& a, hipStream_t stream);
int k1_dz2_feature_blocks(int64_t M, int d);
// the same pass at six tiles (r = 192 / 128): four waves per workgroup, one per SIMD, each with the whole register file (pet_dz6.hip)
bool k1_dz6_applies(const PetBwdArgs& a, int io_fp32);
hipError_t launch_k1_dz6(const PetBwdArgs& a, hipStream_t stream);
int k1_dz6_feature_blocks(int64_t M, int d);
hipError_t launch_k1_dz_reduce(const PetBwdArgs& a, int PR, hipStream_t stream);     // sums the feature blocks of either split form (pet_dz2.hip)
hipError_t launch_pet_gate_cols(const PetBwdArgs& a, const WgradArgs& g, int GS, int NG, int io_fp32, hipStream_t stream);

// Column-parallel pass 2 of the gated K1 backward, round-3 form (pet_cols.hip): weights resident in registers, row tensors streamed
// once; bf16, saved activations, r <= 96.  Partials in the workspace layout of the weight-gradient kernels (wgrad_layout).
// in-launch reduce-scatter of the row-chunk partials (cols_reduce.h has the device side and the protocol)
struct ColsRedJob {                          // one weight gradient (the fields of a finalize job, wgrad.hip)
    float* out; int ldo, transposed, out_rows; float scale;
    float* colsum_x;                         // [d]: bias gradient = column sums of the job's X operand (scaled), or nullptr
    float* colsum_p;                         // [out_rows]: column sums of the job's P operand (unscaled), or nullptr
};
struct ColsRedArgs {
    float* slab;                             // [RC][NCB] slabs, each 8 waves x NJB x RT x 4 x 64 units of 16 bytes
    float* bias_x;                           // [NXS][RC][d] column-sum partials over the columns
    float* bias_p;                           // [RC][NPT * 32] column-sum partials of the bottleneck tiles
    unsigned* ctrl;                          // [NCB][COLS_RED_STRIDE], zeroed by pass 1
    ColsRedJob job[4];
    unsigned spin_limit;
};
#define COLS_RED_STRIDE 320                  // words per column block: 2 + up to 256 row chunks (cols_groups_max(1)), padded
#define COLS_RED_SPIN_DEFAULT (1u << 13)     // polls (~0.3-1 us each) before a workgroup gives up on the rest of its column block
struct ColzArgs {
    const void* dy; const void* x1; const void* x2;      // [M, d] bf16
    const void* dxin;                                    // optional [M, d]: added to dx1
    const void* y;                                       // optional [M, d]: the forward's output (pet_cols6y.hip: dq = dy * y * (1 - g))
    const void* z_a; const void* z_g;                    // the forward's saved z          [M, 32*RT]
    const void* dp_a; const void* dp_g;                  // pass 1's dpre                  [M, 32*RT]
    void* dx1; void* dx2;
    const uint8_t* pk_a; const uint8_t* pk_g;
    int64_t M;
    int d;
    float s2, sd, gs;
    int flags;
    int row_chunks; int64_t rows_per_chunk;              // rows_per_chunk % 32 == 0
    float* part[4];                                      // per job (0 dWd, 1 dWu, 2 dWgd, 3 dWgu): [RC][32RT][d] tiles, [RC][d] column sums of X, [RC][32RT] of P
    ColsRedArgs red;                                     // red.slab != nullptr: the row chunks are summed inside the launch (cols_reduce.h), no finalize pass
};
void k1_cols_plan(int64_t M, int d, int* row_chunks, int64_t* rows_per_chunk);
bool k1_cols_applies(const PetBwdArgs& a, int io_fp32);
hipError_t launch_k1_cols(const ColzArgs& c, int RT, hipStream_t stream);
// the same pass for six tiles (r = 192; pet_cols6.hip: four roles per column quarter, 64-column workgroups)
void k1_cols6_plan(int64_t M, int d, int* row_chunks, int64_t* rows_per_chunk);
bool k1_cols6_applies(const PetBwdArgs& a, int io_fp32);
// ... with the up-side weight gradients one step late and the elementwise block from the forward's output (pet_cols6y.hip)
hipError_t launch_k1_cols6y(const ColzArgs& c, hipStream_t stream);

// K4: out = LN(feats . W^T + b) * gamma + beta (+ R); optionally stores xhat and rstd for the backward
struct VisprojArgs {
    const void* feats;      // [M, F]  IO dtype
    const uint8_t* pk;      // down pack of W [d_out, F] (fragments (stage, u, ct)) followed by fp32 bias[d_out]
    const float* gamma;     // [d_out]
    const float* beta;      // [d_out] or nullptr (T5 RMS norm)
    const void* R;          // [M, d_out] IO dtype or nullptr: added after the norm (position branch + order embeddings)
    void* out;              // [M, d_out] IO dtype
    void* xhat;             // [M, d_out] IO dtype or nullptr
    float* rstd;            // [M] or nullptr
    int64_t M;
    int F, d_out;
    float eps;
    int rms;                // 1: no mean subtraction (T5LayerNorm)
};
hipError_t launch_visproj_fwd(const VisprojArgs& a, int io_fp32, hipStream_t stream);

// K4 forward as a tiled GEMM with the LayerNorm statistics exchanged between the column tiles (visproj_gemm.hip, round 5)
struct VisGemmArgs {
    const void* feats; const void* w;
    const float* bias; const float* gamma; const float* beta;
    const void* R; void* out; void* xhat; float* rstd; float* mean;
    unsigned long long* xch;        // granules [2 parities][nteams][NT consumers][NT producers][BM * 2]: zero before the first launch, left zero
    unsigned* status;               // workspace header: [0] != 0: a statistics exchange timed out (and was repaired), [1] tiles given up so far,
                                    // [2] that count at the last repair, [3] repair workgroups finished
    unsigned* flags;                // [row_blocks][NT]: != 0 = this tile gave up; its pre-norm rows sit where xhat goes (visproj_gemm_repair_kernel)
    unsigned long long* stats;      // [rows][NT] {mean, M2} of a row over a column tile, as floats (lo, hi): every workgroup's, never cleared
    unsigned spin_limit;            // polls of a partner's granules before a wave gives up (0: the default, about a second)
    int64_t M; int F, d_out; float eps; int rms;
    int nteams, row_blocks;
};

bool visproj_gemm_applies(int64_t M, int F, int d_out, int io_fp32);
size_t visproj_gemm_workspace_bytes(int64_t M, int F, int d_out);
size_t visproj_gemm_exchange_bytes(int d_out);                        // the granule area that every launch leaves zeroed (workspace bytes 256 ..)
hipError_t launch_visproj_gemm(VisGemmArgs& a, void* ws, int form, int bm, hipStream_t stream);

// K4's position / order branch R = LN_p(W_p [box, area] + b_p) + img_order_embedding[.] + obj_order_embedding[V - 1 - .] (vispos.hip, round 6)
struct VisPosArgs {
    const float* pos;               // [M, 4] fp32 (x1, x2, y1, y2); M = B N rows, row r = (b, n) = (r / N, r % N)
    const float* w; const float* b; // Linear(5 -> d): [d, 5], [d]
    const float* gamma; const float* beta;      // the branch's norm ([d]; beta nullptr with rms)
    const void* img_tab; const void* obj_tab;   // [n_img, d], [obj_rows, d] (nullptr both: no order embeddings); fp32 or bf16 each
    int img_tab_bf16, obj_tab_bf16;
    const int64_t* img_ids; const int64_t* obj_ids;     // [B or 1, N] (nullptr: image 0 / object n); batch stride 0 broadcasts
    int64_t img_bstride, obj_bstride;
    int n_img; int64_t obj_rows;
    void* out;                      // fwd: R [M, d] IO dtype
    const void* dout;               // bwd: dR [M, d] IO dtype
    float* partial;                 // bwd workspace: [workgroups][8 + image slots][d]
    float* dw; float* db; float* dgamma; float* dbeta; float* dimg;     // bwd results (written, not accumulated); dbeta nullptr with rms, dimg [n_img, d] or nullptr
    int64_t M; int N, d; float eps; int rms;
};
bool vispos_applies(int d, int n_img);
size_t vispos_bwd_workspace_bytes(int64_t M, int d, int n_img);
hipError_t launch_vispos_fwd(const VisPosArgs& a, int io_fp32, hipStream_t stream);
hipError_t launch_vispos_bwd(const VisPosArgs& a, int io_fp32, hipStream_t stream);

// K5 sublayer tail: out = LayerNorm(x1 + dropout(y)) (norm = 1) or x1 + dropout(y) (norm = 0); tail.hip
struct TailArgs {
    const void* y;          // fwd: sublayer output [M, d];          bwd: dy  (written when thr != 0)
    const void* x1;         // fwd: residual [M, d];                 bwd: dx1 (written)
    void* out;              // fwd: result [M, d];                   bwd: dout (read)
    void* h;                // pre-norm sum x1 + dropout(y) [M, d] (fwd: written if non-null; bwd: read), norm = 1 only
    const float* gamma;     // [d] (norm = 1)
    const float* beta;      // [d] or nullptr
    float* mean;            // [M] (norm = 1)
    float* rstd;            // [M]
    uint8_t* keep_out;      // optional [M, d] 0/1 export of the dropout mask (tests)
    float* dgb;             // bwd: partial sums [tail_blocks(M)][2][d] of dgamma / dbeta, or nullptr
    int64_t M;
    int d;
    float eps;
    uint32_t thr;           // drop iff 16-bit uniform < thr;  0 = no dropout
    float keep_scale;       // 1 / (1 - p)
    uint64_t seed;
    const uint64_t* seed_ctr;   // optional device step counter mixed into the seed (rng.h vlpet_eff_seed)
    int norm;
    int post;               // 1: out = LayerNorm(dropout(y)) + x1  (x1 is added AFTER the norm: the visual projectors' position /
                            //    order-embedding term, src/modeling_bart.py:298-299, 324-325); forward only -- the backward of
                            //    that form is the plain one with h = y (dx1 := d/dy, the caller passes dout on as d/dx1)
    int h_xhat;             // bwd, norm = 1: `h` holds the normalised rows xhat (what K4's forward saves), `mean` is not read
    int h_out;              // bwd, norm = 1: `h` holds the LayerNorm OUTPUT rows; xhat = (h - beta) / gamma (`beta` read, `mean` not)
    const void* dres;       // bwd, optional [M, d]: added to dx1 (the gradient another reader of the norm's input parked: the residual
                            //    stream of a pre-norm sublayer feeds the norm and the tail's add, my_transformers/modeling_t5.py:366, 408)
    int rms;                // norm = 1 as T5's RMS norm (my_transformers/modeling_t5.py:235-252): no mean subtraction, no beta; `mean`
                            //    is neither written nor read.  fwd: `y` may be null (out = rmsnorm(x1)); bwd: `h` = the norm's input rows
    void* out2;             // fwd, norm = 0 only: second output = rmsnorm(out) * gamma2 (the NEXT sublayer's T5LayerNorm applied to the sum this
    const float* gamma2;    //    tail produces: one pass instead of two, my_transformers/modeling_t5.py:408 + :366 of the next sublayer); rstd [M] written
};
hipError_t launch_tail(const TailArgs& a, int io_fp32, bool bwd, hipStream_t stream);
int tail_blocks(int64_t M);

// K3 at rank <= 8 as a streaming row kernel (lora8.hip)
struct Lora8Args {
    const void* x; const void* base; void* out;     // [M, d] bf16
    const uint8_t* pk;      // the pair's packs (bf16 plane, one tile): the kernel decodes A [r, d] and B [d, r] from the down / up packs
    DropSpec drop;          // dropout of x (generator / explicit mask); bits_out = where the packed mask goes in the training form
    void* save;             // training form: z [M, 32] bf16 (columns >= 8 zero), or nullptr
    int64_t M; int d;
    float scaling;
};
bool lora8_applies(int64_t M, int d, int r, int io_fp32);
hipError_t launch_lora8_fwd(const Lora8Args& a, hipStream_t stream);
hipError_t launch_tail_reduce(const float* part, int nb, int d, float* dgamma, float* dbeta, hipStream_t stream);
hipError_t launch_colsum(const void* x, int64_t M, int n, float* part, float* out, int io_fp32, hipStream_t stream);
hipError_t launch_colsum_partial(const void* x, int64_t M, int n, float* part, int io_fp32, hipStream_t stream);
// batched form of launch_tail_reduce: part [nb][2 d] -> out0 [d], out1 [d] (either may be null) per job, one launch
struct ReduceJob { const float* part; float* out0; float* out1; int nb; int d; };
#define VLPET_REDUCE_BATCH 96
struct ReduceBatch { ReduceJob j[VLPET_REDUCE_BATCH]; int n; };
hipError_t launch_tail_reduce_batch(const ReduceBatch& b, int max_d, hipStream_t stream);

// Downsample (adaptive max pool over the token grid), downsample.hip
struct PoolArgs {
    const void* x;      // [n_img, s_in*s_in, dim]
    void* out;          // [n_img, s_out*s_out, dim]
    int64_t n_img;
    int s_in, s_out, dim;
};
hipError_t launch_downsample(const PoolArgs& a, int in_fp32, int out_fp32, hipStream_t stream);

// Row kernels of the small / middleX / middleY gates, rowgate.hip
enum RowOp { ROW_DOT = 0, ROW_AFFINE = 1, ROW_BWD = 2, VEC_FWD = 3, VEC_BWD = 4 };
struct RowArgs {
    const void* a;      // [M, d] IO dtype
    const void* c;      // [M, d] or nullptr
    const void* e;      // [M, d] or nullptr
    void* o1;           // [M, d] output or nullptr
    void* o2;           // [M, d] output or nullptr
    const float* va;    // [d] or nullptr
    const float* vc;    // [d] or nullptr
    const float* ra;    // [M] or nullptr
    const float* rb;    // [M] or nullptr
    float* rs;          // [M] output (ROW_DOT)
    float* part;        // [rowgate_blocks(M)][2][d] partial column sums (ROW_BWD, VEC_BWD)
    int64_t M;
    int d;
};
hipError_t launch_rowgate(const RowArgs& a, int op, int io_fp32, hipStream_t stream);
int rowgate_blocks(int64_t M);

// Fused clip + AdamW over the flat trainable buffer, optim.hip
struct AdamwArgs {
    float* p; float* g; float* m; float* v;       // [n] fp32, 16-byte aligned
    const uint8_t* decay;                          // [n] 1 = weight decay applies, or nullptr (all decay)
    int64_t n;
    const float* partials; int n_partials;         // sum of squares of g per workgroup of sumsq_kernel
    float max_norm;                                // <= 0: no clipping
    float grad_scale;                              // 1 / world_size (gradient averaging folded in)
    float lr, beta1, beta2, eps, weight_decay;
    float bias_c1, bias_c2_sqrt;                   // 1 - beta1^t, sqrt(1 - beta2^t)
    int decay_first;                               // 1: torch.optim.AdamW order, 0: transformers.AdamW order
    int eps_scaled;                                // 1: eps added to sqrt(v)/sqrt(bc2) (torch), 0: to sqrt(v) (transformers)
    int zero_grad;                                 // 1: clear g after use
    float* norm_out;                               // optional device scalar: the pre-clip global norm
    const int32_t* slice_of;                       // optional [n]: parameter index of every element (per-parameter steps)
    const float* slice_bc;                         //   [n_slices][2]: 1 - b1^t, sqrt(1 - b2^t) per parameter; <= 0 = no grad this step
};
hipError_t launch_add_inplace(void* dst, const void* src, int64_t n, int io_fp32, hipStream_t stream);   // dst += src (IO dtype)
struct SumNArgs { const void* src[8]; void* out; };
hipError_t launch_sum_n(const SumNArgs& a, int n, int64_t len, int io_fp32, hipStream_t stream);      // out = src[0] + ... + src[n - 1], n in 2 .. 8 (optim.hip)
hipError_t launch_sumsq(const float* g, int64_t n, float* partials, hipStream_t stream);
hipError_t launch_adamw(const AdamwArgs& a, hipStream_t stream);
int optim_blocks(int64_t n);

// Row groups (32 rows each) per workgroup.  One workgroup per CU is resident (LDS), a workgroup's time is
// roughly (rows + a fixed part for the weight stream and the prologue), and the chip takes the workgroups in
// rounds of 256: pick the size with the cheapest rounds x (RG + 1).  At M = 46,648 that is 3 (486 workgroups,
// two rounds of 96 rows instead of two of 128); at M = 15,272 it is 2 (239 workgroups in one round instead of
// 120 CUs working and 136 idle).
inline int pick_row_groups(int64_t M, int max_rg, int min_rg) {
    if (const int rg = vlpet_tuning().rg; rg >= min_rg && rg <= max_rg && rg > 0) return rg;       // (diagnosis builds only)
    int best = max_rg;
    int64_t best_cost = -1;
    for (int rg = max_rg; rg >= min_rg; --rg) {
        const int64_t wgs = (M + 32 * rg - 1) / (32 * rg);
        const int64_t cost = ((wgs + 255) / 256) * (rg + 1);
        if (best_cost < 0 || cost < best_cost) { best = rg; best_cost = cost; }
    }
    return best;
}

// FFN activation + dropout of the backbone (actdrop.hip): out = dropout(act(x)) / dx = dy * mask * act'(x)
#define VLPET_ACT_GELU 0        // erf form (F.gelu; BART)
#define VLPET_ACT_GELU_NEW 1    // tanh form
#define VLPET_ACT_RELU 2        // T5 DenseReluDense
struct ActDropArgs {
    const void* x;          // [n] pre-activation
    const void* dy;         // backward: gradient of the output
    void* out;              // forward: dropout(act(x)); backward: dx
    uint8_t* keep_out;      // forward only: optional 0/1 export of the mask (tests)
    int64_t n;              // elements, multiple of 8
    int act;
    uint32_t thr;           // drop iff 16-bit uniform < thr (0: no dropout)
    float keep_scale;
    uint64_t seed;
    const uint64_t* seed_ctr;   // optional device step counter mixed into the seed (rng.h vlpet_eff_seed)
};
hipError_t launch_act_dropout(const ActDropArgs& a, bool bwd, int io_fp32, hipStream_t stream);
// x = dropout(cat([a, v], dim = 1), p) over [B, La | Lv, d] (actdrop.hip); bwd: x = dx in, a / v = da / dv out (nullptr: not wanted)
hipError_t launch_cat_dropout(const void* a, const void* v, void* x, int64_t B, int La, int Lv, int d, uint32_t thr, float keep_scale,
                              uint64_t seed, const uint64_t* seed_ctr, bool bwd, int io_fp32, hipStream_t stream);

// token-level cross entropy over the LM-head logits (celoss.hip)
struct CeArgs {
    const void* logits;     // [N, ld] IO dtype, columns [0, V) valid
    const int64_t* labels;  // [N]; < 0 = ignored
    float* loss;            // forward: [N]
    float* lse;             // [N] log-sum-exp of the row (written by the forward, read by the backward)
    const float* dloss;     // backward: [N]
    void* dlogits;          // backward: [N, ld]
    int64_t N;
    int V, ld;
    unsigned int* bad;      // forward, optional: += number of labels outside [0, V) other than ignore_index -100 (treated as ignored)
};
hipError_t launch_ce(const CeArgs& a, bool bwd, int io_fp32, hipStream_t stream);

// short-sequence attention of the backbone (attn.hip): bf16, head dim 64, Lq, Lk <= 128; tensors [B, L, H, 64] contiguous
struct AttnArgs {
    const __bf16* q; const __bf16* k; const __bf16* v;
    __bf16* o;              // forward output; backward: the forward's output (read)
    float* lse;             // [B, H, Lq] log2-domain log-sum-exp of the scaled scores (written fwd, read bwd)
    const __bf16* dout;     // backward
    __bf16* dq; __bf16* dk; __bf16* dv;
    const uint8_t* key_mask;   // [B, Lk] 1 = attend, or nullptr
    const float* bias;      // optional additive score bias shared by the batch (T5's relative position bias, my_transformers/modeling_t5.py
                            //   :520-560): [H, Lqp, Lkp] fp32, Lqp / Lkp = Lq / Lk rounded up to 32, zero padded; nullptr = none
    const float* bias_t;    // the same bias transposed, [H, Lkp, Lqp] (the key-major phase of the backward reads it along queries)
    uint8_t* keep_out;      // forward only: optional [B, H, Lq, Lk] export of the dropout mask (tests)
    int B, H, Lq, Lk, causal;
    int ld_q, ld_kv;        // row stride (elements) of q / dq and of k / dk: H*64 for separate projection outputs, 3*H*64
                            // for the columns of a fused [B, L, 3*H*64] q|k|v buffer (o and dout are always H*64 wide)
    int ld_v;               // row stride of v / dv (round 6: k may be a column block of the decoder layers' fused key projection,
                            // [B, Lk, n_layers*H*64], while v -- behind the value-parallel adapter -- stays H*64 wide)
    float scale;
    uint32_t thr;           // drop iff hash < thr (p * 2^32); 0 = no dropout
    float inv_keep;
    uint64_t seed;
    const uint64_t* seed_ctr;   // optional device step counter mixed into the seed (rng.h vlpet_eff_seed)
    int dbg;                // diagnosis build only (VLPET_DBG): backward ablations -- 1 no units, 4 no stores, 8 no compute (zeros stored)
};
size_t attn_lds_bytes(int Lq, int Lk, int bwd);
hipError_t launch_attn(const AttnArgs& a, bool bwd, hipStream_t stream);
