/*
 * rankpo_hip.h -- C ABI of librankpo_hip.so: the MI355X (gfx950) scoring hot path of a
 * contrastive / RankPO embedding trainer.
 *
 * The reference (yflyzhang/RankPO) is pure Python and has no FFI; each entry point below replaces the
 * stock-PyTorch op sequence at the cited reference lines.  Plain device pointers + sizes + a HIP stream;
 * no torch types.  Every call is asynchronous on `stream`, allocates nothing, never synchronises with
 * the host, never throws and never exits: it returns RPO_OK or a negative rpo_status.  All buffers are
 * owned by the caller.  The library keeps no global mutable state and reads no environment variable, so
 * forward may be called from the Python main thread and backward from PyTorch's autograd thread
 * concurrently on different streams, and two callers in one process cannot disagree about a kernel variant.
 *
 * Deviation from SURVEY.md §8b(5), deliberate: there is no `rpo_allgather_qp`.  The cross-device exchange of the
 * path (modeling.py:287-290, 331-404) is ONE rank-major all-gather of a [B + B G, d] block with no arithmetic in it;
 * the RCCL communicator it needs is the one torch.distributed already owns (bootstrap, stream ordering against the
 * caching allocator, async work handles), and a second communicator created behind the C ABI would double RCCL's
 * per-peer xGMI buffers and need its own bootstrap for nothing.  The gather therefore stays on the host side
 * (rankpo_amd/distributed.py: FusedQPGather); what the library contributes to the exchange is that the InfoNCE
 * kernels take the GATHERED matrices plus this rank's row window (q_row0 / q_rows / p_row0 / p_rows below) and
 * write gradients for exactly those rows, which is what removes the backward collective.
 *
 * Embedding matrices are row-major [rows, d] with a contiguous inner dimension; dtype selects the
 * storage type of embeddings / hidden states / scores (f32 or bf16; accumulation is always f32).
 */
#ifndef RANKPO_HIP_H
#define RANKPO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rpo_stream_t; /* hipStream_t */

typedef enum {
    RPO_OK = 0,
    RPO_ERR_INVALID_ARG = -1,   /* null pointer, non-positive size, bad enum */
    RPO_ERR_UNSUPPORTED = -2,   /* shape / dtype / alignment this build does not handle */
    RPO_ERR_WORKSPACE = -3,     /* workspace too small (see *_workspace_bytes) */
    RPO_ERR_LAUNCH = -4         /* hipGetLastError() != hipSuccess after the launch */
} rpo_status;

typedef enum { RPO_DT_F32 = 0, RPO_DT_BF16 = 1, RPO_DT_F16 = 2 } rpo_dtype;
typedef enum { RPO_POOL_LAST = 0, RPO_POOL_CLS = 1 } rpo_pool_mode;
/* INBATCH: scores [Q,P], target_i = i * (P / Q)  (modeling.py:292-302)
 * FIRST:   scores [Q,G] with G = P / Q, row i scored against p rows i*G .. i*G+G-1, target 0 (305-311) */
typedef enum { RPO_TARGET_INBATCH = 0, RPO_TARGET_FIRST = 1 } rpo_target_mode;
typedef enum { RPO_LOSS_SIGMOID = 0, RPO_LOSS_HINGE = 1 } rpo_loss_type;

int rpo_version(void);
/* What this build of the library contains beyond the default: a bit set of rpo_build_flag.  RPO_BUILD_ONEWAVE64: the head_dim-64
 * one-wave-per-SIMD attention kernels (q_block = 64 at head_dim 64 in rpo_flash_attn_fwd / _bwd; `make ONEWAVE64=1`).  They are
 * correct and measured 4-10 % slower than the default kernels at head_dim 64, so the default build answers RPO_ERR_UNSUPPORTED there. */
typedef enum { RPO_BUILD_ONEWAVE64 = 1 } rpo_build_flag;
int rpo_build_flags(void);
const char* rpo_status_string(int status);
/* hipGetErrorString of the HIP error behind the calling thread's most recent RPO_ERR_LAUNCH (diagnostics). */
const char* rpo_last_hip_error(void);

/* ---------------------------------------------------------------------------------------------
 * (1) pooling + L2 normalisation.
 * Replaces modeling.py:224-236 (== rankpo_trainer.py:409-417, modeling.py:523-534):
 *   idx_n = (argmin_l mask[n,l] - 1) mod L   (first index of the row minimum; RPO_POOL_LAST)
 *        or 0                                  (RPO_POOL_CLS, modeling.py:231-232)
 *   x_n = h[n, idx_n, :];  out_n = normalize ? x_n / max(||x_n||_2, eps) : x_n
 * h: [N, L, d] with element strides (h_stride_n, h_stride_l, 1).  mask: int64 [N, L] contiguous
 * (may be NULL for RPO_POOL_CLS).  Saved for backward: idx_out int32 [N], norm_out f32 [N] (= ||x_n||).
 * --------------------------------------------------------------------------------------------- */
int rpo_pool_normalize_fwd(const void* h, int64_t h_stride_n, int64_t h_stride_l, const int64_t* mask,
                           int64_t N, int64_t L, int64_t d, int dtype, int pool_mode, int normalize,
                           float eps, void* out, int32_t* idx_out, float* norm_out, rpo_stream_t stream);

/* Backward of (1) (autograd of index-select + F.normalize).  grad_out, out: [N, d].
 *   dx_n = normalize ? (norm_n >= eps ? (g_n - out_n <out_n, g_n>) / norm_n : g_n / eps) : g_n
 * If dh != NULL the dense gradient [N, L, d] (contiguous) is written IN FULL: zeros everywhere except
 * row idx_n of sample n (one pass; replaces zeros() + index_put_).  If drow != NULL, dx is also
 * written as [N, d] (for callers that scatter themselves).  At least one of them must be non-NULL. */
int rpo_pool_normalize_bwd(const void* grad_out, const void* out, const int32_t* idx, const float* norm,
                           int64_t N, int64_t L, int64_t d, int dtype, int normalize, float eps,
                           void* dh, void* drow, rpo_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (2) similarity + temperature + InfoNCE cross-entropy, fused.
 * Replaces modeling.py:292-314 (training) and :321 (eval: pass lse_out = loss_out = NULL, temperature 1).
 *   INBATCH: scores[i,j] = <q_i, p_j> / T                      [Q, P]
 *   FIRST:   scores[i,g] = <q_i, p_{iG+g}> / T                  [Q, G]
 *   lse_i = log sum_j exp(scores[i,j]);  loss = mean_i (lse_i - scores[i, target_i])
 * scores_out has the storage dtype; with bf16 storage the rounding points are the reference's
 * (dot -> bf16, / T -> bf16) and lse / loss are computed in f32 from the stored (rounded) scores.
 * lse_out f32 [Q], loss_out f32 [1].  workspace: rpo_infonce_workspace_bytes(), 256-byte aligned.
 * --------------------------------------------------------------------------------------------- */
size_t rpo_infonce_workspace_bytes(int64_t Q, int64_t P, int64_t d, int dtype);

int rpo_infonce_fwd(const void* q, const void* p, int64_t Q, int64_t P, int64_t d, int dtype,
                    float temperature, int target_mode, void* scores_out, float* lse_out, float* loss_out,
                    void* workspace, size_t workspace_bytes, rpo_stream_t stream);

/* Fault injection for the tests (not on the hot path): sets every arrival-counter slot of the multi-block single-launch
 * forward (Q <= 64 with several blocks, e.g. the 64 x 384 matrix of 8 ranks) to `word`, as a launch that never finished
 * would leave it.  The next rpo_infonce_fwd must still write lse and loss (the slots are epoch-stamped). */
int rpo_infonce_debug_poison_tickets(uint64_t word, rpo_stream_t stream);

/* Backward of (2): gradients of grad_loss[0] * loss with respect to the caller's OWN rows only
 * (q rows [q_row0, q_row0 + q_rows), p rows [p_row0, p_row0 + p_rows)): with cross-device negatives
 * the other rows of the gathered matrices are constants (modeling.py:374-377), so nothing else is
 * needed and no backward collective exists.  grad_loss: device f32 scalar.  dq_out [q_rows, d],
 * dp_out [p_rows, d] in the storage dtype.  Either output may be NULL (then it is skipped). */
int rpo_infonce_bwd(const void* q, const void* p, const void* scores, const float* lse,
                    const float* grad_loss, int64_t Q, int64_t P, int64_t d, int dtype, float temperature,
                    int target_mode, int64_t q_row0, int64_t q_rows, int64_t p_row0, int64_t p_rows,
                    void* dq_out, void* dp_out, void* workspace, size_t workspace_bytes, rpo_stream_t stream);

/* GEMM form of the backward for large Q*P (in-batch mode): writes the softmax gradient itself,
 *   ds_out  [q_rows, P] = dS[q_row0 .. , :]      and      dst_out [p_rows, Q] = (dS[:, p_row0 ..])^T
 * with dS[i,j] = grad_loss/(Q T) (exp(S[i,j] - lse_i) - [j == i (P/Q)]), in the storage dtype, so that
 * dq = ds_out @ p and dp = dst_out @ q are two plain library GEMMs (hipBLASLt via the caller).  Either output
 * may be NULL. */
int rpo_infonce_ds(const void* scores, const float* lse, const float* grad_loss, int64_t Q, int64_t P, int dtype,
                   float temperature, int64_t q_row0, int64_t q_rows, int64_t p_row0, int64_t p_rows,
                   void* ds_out, void* dst_out, rpo_stream_t stream);

/* The two products of that GEMM form on the FORWARD kernel's own MFMA frame (round 5; replaces autograd's
 * `grad_scores @ p` / `grad_scores.t() @ q` of modeling.py:252's matmul, which rounds 1-4 handed to hipBLASLt):
 *   C [rows_b, rows_a] = B [rows_b, K] A [rows_a, K]^T      bf16 in, f32 accumulation, one rounding to bf16
 * 256 x 256 tiles, K-step 64, 16-byte LDS-DMA staging, phased MFMA schedule (sim_tile256_kernel with a plain epilogue).
 * Both operands are contiguous along the reduction, as q and p are in the forward:
 *   dq [q_rows, d] = rpo_sim_gemm_nt(a = p_all^T [d, P], b = ds_out  [q_rows, P], K = P)
 *   dp [p_rows, d] = rpo_sim_gemm_nt(a = q_all^T [d, Q], b = dst_out [p_rows, Q], K = Q)
 * with the transposed embeddings from rpo_transpose.  lda / ldb / ldc: row strides in elements.  Requires K % 64 == 0,
 * strides % 8 == 0 and 16-byte aligned pointers (RPO_ERR_UNSUPPORTED otherwise: the caller keeps the library GEMM). */
int rpo_sim_gemm_nt(const void* a, int64_t rows_a, int64_t lda, const void* b, int64_t rows_b, int64_t ldb, int64_t K,
                    void* c, int64_t ldc, rpo_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (3) RankPO paired scoring + loss + metrics.
 * Replaces rankpo_trainer.py:436-443 (scores), 545-566 (rankpo_loss), 482-520 (loss mix + metrics).
 *   scores[b,g] = <q_b, p_{2b+g}>   (g = 0 chosen, 1 rejected; unscaled)
 *   adv = (c - r) - (reference_free ? 0 : ref_c - ref_r);  logits = adv / T - gamma_beta_ratio
 *   sigmoid: -logsig(beta z)(1 - eps) - logsig(-beta z) eps ;  hinge: relu(1 - beta z)
 *   loss = rankpo_weight * mean(losses) [if > 0] + sft_weight * CE(scores / T, 0) [if > 0]
 * ref_chosen / ref_rejected: f32 [B] or NULL (no reference model -> 0).  Outputs (all f32):
 * scores_out [B,2], losses_out [B], loss_out [1], metrics_out [RPO_NUM_METRICS], dscores_out [B,2]
 * (= d loss / d scores, consumed by rpo_rankpo_bwd).
 * --------------------------------------------------------------------------------------------- */
typedef struct {
    float beta;
    float temperature;
    float gamma_beta_ratio;
    float label_smoothing;
    float rankpo_weight;
    float sft_weight;
    int32_t loss_type;      /* rpo_loss_type */
    int32_t reference_free; /* 0 / 1 */
} rpo_rankpo_params;

enum {
    RPO_METRIC_RANKPO_LOSS = 0,      /* valid iff rankpo_weight > 0 */
    RPO_METRIC_SFT_LOSS = 1,         /* valid iff sft_weight > 0 */
    RPO_METRIC_REWARDS_CHOSEN = 2,   /* mean beta (c - ref_c) */
    RPO_METRIC_REWARDS_REJECTED = 3,
    RPO_METRIC_REWARDS_ACCURACIES = 4,
    RPO_METRIC_REWARDS_MARGINS = 5,
    RPO_METRIC_SCORES_CHOSEN = 6,
    RPO_METRIC_SCORES_REJECTED = 7,
    RPO_METRIC_SCORES_MARGINS = 8,
    RPO_NUM_METRICS = 9
};

int rpo_rankpo_fwd(const void* q, const void* p, const float* ref_chosen, const float* ref_rejected,
                   int64_t B, int64_t d, int dtype, const rpo_rankpo_params* params, float* scores_out,
                   float* losses_out, float* loss_out, float* metrics_out, float* dscores_out,
                   rpo_stream_t stream);

/* dq_b = gl (ds[b,0] p_{2b} + ds[b,1] p_{2b+1});  dp_{2b+g} = gl ds[b,g] q_b;  gl = grad_loss[0]. */
int rpo_rankpo_bwd(const void* q, const void* p, const float* dscores, const float* grad_loss, int64_t B,
                   int64_t d, int dtype, void* dq_out, void* dp_out, rpo_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (4) "next" row f2: flat-buffer AdamW step (replaces the DeepSpeed ZeRO-1 bf16 optimizer step the reference
 * scripts configure: configs/ds_zero1_config_llama.json, scripts/train/run_contrastive.sh:33-40).
 * torch.optim.AdamW semantics on n elements (n % 4 == 0, 16-byte aligned buffers):
 *   g = grad * (grad_scale ? grad_scale[0] : 1);  w *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
 *   w -= (lr / bias_corr1) * m / (sqrt(v) / sqrt(bias_corr2) + eps)
 * dtype = storage type of param and grad.  bf16 params need an f32 `master` copy (updated, then rounded into
 * param); for f32 params master may be NULL.  grad_scale: device f32 scalar or NULL (clip / 1/GAS factor).
 * rpo_sumsq_partial: partial_out[b] = sum of squares of block b's share of x (for the global grad norm).
 * --------------------------------------------------------------------------------------------- */
int rpo_adamw_step(void* param, float* master, const void* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   int dtype, float lr, float beta1, float beta2, float eps, float weight_decay,
                   float bias_corr1, float bias_corr2, const float* grad_scale, rpo_stream_t stream);
int rpo_sumsq_partial(const void* x, int64_t n, int dtype, float* partial_out, int nblocks, rpo_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (5) fused elementwise pieces of the Llama block used by rankpo_amd/encoder.py (the encoder itself stays under
 * PyTorch-ROCm).  16-byte aligned; element counts multiples of 16 / sizeof(elem).
 *   rpo_swiglu_fwd: out = silu(g) * u                   (replaces F.silu + mul of HF LlamaMLP).  g, u: [rows, cols]
 *     with row stride ld_gu elements (g and u may be the two halves of ONE fused gate|up projection output: u = g + cols,
 *     ld_gu = 2 cols); out: [rows, cols] with row stride ld_out.
 *   rpo_swiglu_bwd: dg = dout * u * silu'(g), du = dout * silu(g); dg / du with row stride ld_dgu (again possibly the
 *     two halves of one [rows, 2 cols] buffer).  prod_out (may be NULL): also writes silu(g) * u, row stride ld_prod --
 *     the recomputed forward product that the weight gradient of the following projection needs; prod_out == dout is
 *     allowed (every element of dout is read before it is overwritten), which makes the recompute free of extra traffic.
 *   rpo_rope: x_out = rot(x_in), both [rows, heads, head_dim] (row stride `row_stride` elements; x_in == x_out allowed), HF
 *     rotate_half convention: (x1, x2) -> (x1 cos - x2 sin, x2 cos + x1 sin) with x1 / x2 the low / high half of each
 *     head; cos_tab / sin_tab: f32 [period, head_dim / 2]; row r uses table row r % period.  backward != 0 applies
 *     the inverse rotation (the gradient of the forward rotation).
 * --------------------------------------------------------------------------------------------- */
int rpo_swiglu_fwd(const void* g, const void* u, void* out, int64_t rows, int64_t cols, int64_t ld_gu, int64_t ld_out,
                   int dtype, rpo_stream_t stream);
int rpo_swiglu_bwd(const void* g, const void* u, const void* dout, void* dg, void* du, void* prod_out, int64_t rows,
                   int64_t cols, int64_t ld_gu, int64_t ld_dout, int64_t ld_dgu, int64_t ld_prod, int dtype,
                   rpo_stream_t stream);
/* rpo_swiglu_bwd with the recomputed product written TRANSPOSED: prod_t_out [cols, rows] (row stride ld_t >= rows), the
 * operand layout in which the weight gradient of the following (down) projection, dW = dY^T prod, has BOTH operands contiguous
 * along the token reduction.  dgu_t_out (may be NULL): dg and du ALSO written transposed, [2 cols, rows] with the same row stride
 * (dg^T in rows 0 .. cols - 1, du^T behind it), the same for the weight gradient of the fused gate|up projection; dg / du are
 * written row-major in any case (the input-gradient GEMM reads them).  The transposed outputs may not overlay the inputs. */
int rpo_swiglu_bwd_t(const void* g, const void* u, const void* dout, void* dg, void* du, void* prod_t_out, void* dgu_t_out,
                     int64_t rows, int64_t cols, int64_t ld_gu, int64_t ld_dout, int64_t ld_dgu, int64_t ld_t, int dtype,
                     rpo_stream_t stream);
int rpo_rope(const void* x_in, void* x_out, int64_t row_stride, const float* cos_tab, const float* sin_tab,
             int64_t rows, int64_t heads, int64_t head_dim, int64_t period, int dtype, int backward,
             rpo_stream_t stream);

/* 2-D transpose out[cols, rows] = in[rows, cols]^T, row-major with row strides ld_in / ld_out (elements); HBM-bound.  Used
 * by the encoder's backward to hand hipBLASLt ONE operand of the weight-gradient GEMM dW = dY^T X (reduction over the
 * tokens) contiguous along the reduction, and W^T to the input-gradient GEMM (stands where PyTorch's autograd of
 * nn.Linear issues `grad_output.t() @ input` / `grad_output @ weight`). */
int rpo_transpose(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out, int dtype,
                  rpo_stream_t stream);

/* Residual add + RMSNorm, fused (HF LlamaDecoderLayer: `h = residual + delta; y = rmsnorm(h) * w`).
 *   fwd: x_new = x + delta (delta NULL: x_new = x, x_out unused); rstd_r = rsqrt(mean(x_new_r^2) + eps);
 *        y = x_new * rstd * w.  x, delta, x_out, y: [rows, d]; rstd_out: f32 [rows].
 *   bwd: g = dy * w; c_r = mean(g_r * xhat_r); dx = (g - xhat c) * rstd + dres (dres NULL: no residual gradient);
 *        dw_partial: f32 [rpo_add_rmsnorm_waves(rows), d], one partial sum of dy * xhat per wave (every row of it is
 *        written); the caller sums over the first dimension.
 * d % (16 / sizeof(elem)) == 0, d <= 8192 (bf16) / 4096 (f32) forward, half of that backward. */
int rpo_add_rmsnorm_waves(int64_t rows);
int rpo_add_rmsnorm_fwd(const void* x, const void* delta, const void* weight, float eps, void* x_out, void* y_out,
                        float* rstd_out, int64_t rows, int64_t d, int dtype, rpo_stream_t stream);
int rpo_add_rmsnorm_bwd(const void* dy, const void* x_new, const void* weight, const float* rstd, const void* dres,
                        void* dx_out, float* dw_partial, int64_t rows, int64_t d, int dtype, rpo_stream_t stream);

/* Causal variable-length flash attention, forward, head_dim 64 or 128, bf16, grouped-query heads (encoder side; the packed
 * encoder path of rankpo_amd/encoder.py).  q: [T, num_heads, hd] with token stride q_stride elements (heads contiguous),
 * k / v: [T, num_kv_heads, hd] likewise (all three may be views of one fused projection output).  cu_seqlens: int32
 * [N + 1].  tiles = the query-tile work list, one entry per block of 128 queries, in the format tile_cols names:
 *   2: int32 [ntiles][2] = (sequence id, first query row inside the sequence), heaviest first; one launch block per
 *      (entry, head);
 *   3: int32 [ntiles][3] = (sequence id, first query row, head), ntiles % 8 == 0: launch block b takes entry
 *      (b % 8) * ntiles / 8 + b / 8, so the blocks that share an XCD walk one eighth of the list in order; the caller puts
 *      all entries of one (sequence, kv head) next to each other in one eighth (their K / V then stay in that XCD's L2);
 *      entries whose first query row is >= 2^30 are padding.
 * out: [T, num_heads * hd] (token stride out_stride), lse = log sum_j exp(scale * <q_i, k_j>)
 * over the keys j <= i of the same sequence, f32, laid out [num_heads][T] when lse_max_len == 0 or padded
 * [N][num_heads][lse_max_len] (the layout PyTorch's flash-attention backward reads) when lse_max_len > 0.
 * lse may be NULL: a forward-only caller (ModelForInference.encode, modeling.py:473-554; the RankPO ref_model under
 * inference_mode, rankpo_trainer.py:468-477) -- no row statistics are written, nothing else changes.
 * rope_cos / rope_sin (both NULL, or both f32 [rope_period][hd / 2], 16-byte aligned; token t uses row t % rope_period): q
 * arrives UN-rotated and is rotated IN PLACE (q is written!) by the block that owns each (128 queries x head) piece -- the same
 * arithmetic as rpo_rope -- so that the separate rotary pass only has the k heads left; k must arrive rotated (every query
 * block reads it).  The backward entry point reads the rotated q from memory.
 * q_block = the query rows one work-list entry stands for: 128 (0 means 128), or 64 where (num_heads / num_kv_heads) % 4 == 0: an
 * entry is then 64 queries x the FOUR consecutive q heads that begin at the entry's head (format 3: the head column, a multiple
 * of 4; format 2: one launch block per (entry, group of 4 heads)), which share one kv head -- the one-wave-per-SIMD forward
 * (fa_fwd128w_kernel / fa_fwd64w_kernel); out must be 16-byte aligned and out_stride % 8 == 0 there.  Anything else:
 * RPO_ERR_UNSUPPORTED.  The backward's query list stays a 128-row list either way. */
int rpo_flash_attn_fwd(const void* q, const void* k, const void* v, int64_t q_stride, int64_t k_stride,
                       int64_t v_stride, const int* cu_seqlens, const int* tiles, int64_t ntiles, int64_t tile_cols,
                       int64_t total_tokens,
                       int64_t num_heads, int64_t num_kv_heads, int64_t head_dim, float scale, void* out,
                       int64_t out_stride, float* lse, int64_t lse_max_len, const float* rope_cos, const float* rope_sin,
                       int64_t rope_period, int64_t q_block, rpo_stream_t stream);

/* Backward of rpo_flash_attn_fwd, head_dim 64 or 128 (two launches: dQ, which also computes the row constants, then dK/dV; no
 * atomics, deterministic).  lse: f32 [num_heads][T] as written by the forward with lse_max_len == 0; delta: f32
 * [2][num_heads][T] scratch (written here: -rowsum(dout * out) and -lse / scale, the initial accumulators of the dP and S
 * chains).  q_tiles as in the forward (q_tile_cols = its format); k_tiles: int32 [n_k_tiles][3] =
 * (sequence id, kv head, first key of a key block); key_block = the number of keys one entry stands for and thereby the dK/dV
 * kernel that consumes the table.  head_dim 64: 256 (one wave per SIMD, entries dealt to the 8 XCDs in equal eighths, padded
 * with first key >= 2^30) or 64 (the 8-wave kernel; entries sorted by (sequence, head, key)); head_dim 128: 128 (one wave per
 * SIMD, 32 keys per wave; table dealt and padded like the 256 one).  Any other value is RPO_ERR_UNSUPPORTED: the meaning of
 * the table is an argument, never process-global state.  sweep_down (key_block 256 and 128; ignored at 64, and when the q-head group
 * is not a power of two): the dK/dV blocks sweep the query slices from the last one downwards, the group's q heads innermost (pairs
 * with a work list that keeps a (sequence, kv head)'s key blocks next to each other: they then read each Q / dO slice at about the
 * same time, once from HBM).  dk / dv differ from the ascending sweep by summation order only; either order is deterministic.
 * dq: [T, num_heads, hd], dk / dv: [T, num_kv_heads, hd] (token strides given), every valid row is written.
 * rope_cos / rope_sin (both NULL, or both f32 [rope_period][hd / 2], 16-byte aligned; token t uses row t % rope_period; the
 * tables rpo_rope rotated q and k with): dq and dk then leave as the gradients w.r.t. the PRE-rotary q / k -- the inverse
 * rotation of rpo_rope(backward = 1) applied in the kernels' epilogues in f32 before the one rounding to bf16, instead of a
 * separate pass over d(q|k).  Not offered with key_block 64 (RPO_ERR_UNSUPPORTED).
 * q_block = the query rows one entry of q_tiles stands for: 128 (0 means 128), or 64 with head_dim 64 and
 * (num_heads / num_kv_heads) % 4 == 0 -- the 64-query x 4-head entries of rpo_flash_attn_fwd's q_block = 64, walked by the
 * one-wave-per-SIMD dQ kernel (fa_bwd_dq64w_kernel; dq 16-byte aligned, dq_stride % 8 == 0).  Anything else: RPO_ERR_UNSUPPORTED. */
int rpo_flash_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, int64_t q_stride,
                       int64_t k_stride, int64_t v_stride, int64_t out_stride, int64_t dout_stride,
                       const int* cu_seqlens, const int* q_tiles, int64_t n_q_tiles, int64_t q_tile_cols, const int* k_tiles,
                       int64_t n_k_tiles, int64_t key_block, int64_t sweep_down, int64_t total_tokens, int64_t num_heads,
                       int64_t num_kv_heads, int64_t head_dim, float scale, const float* lse, float* delta, void* dq, void* dk, void* dv,
                       int64_t dq_stride, int64_t dk_stride, int64_t dv_stride, const float* rope_cos, const float* rope_sin,
                       int64_t rope_period, int64_t q_block, rpo_stream_t stream);

/* One-query-per-sequence attention (rankpo_amd/csrc/lastq_attention.hip): the attention of the LAST block of a last-token-pooled
 * encoder.  The reference pools `last_hidden_state[n, last real token]` (modeling.py:224-230, rankpo_trainer.py:409-413), so the
 * last block's attention output is read for ONE query per sequence -- its last token, which sees every key of its sequence (no
 * mask).  Stands where the packed encoder path called PyTorch's flash-attention op (an AOTriton kernel) with one-row queries.
 *   q: [num_seqs, num_heads, hd] bf16 (sequence stride q_stride elements, heads contiguous) -- already rotated;
 *   k / v: [T, num_kv_heads, hd] packed tokens (token strides k_stride / v_stride; k rotated), cu_seqlens: int32 [num_seqs + 1];
 *   out: [num_seqs, num_heads * hd] bf16; lse: f32 [num_seqs][num_heads] = log sum_j exp(scale <q, k_j>).
 * HBM-bound: K / V are read exactly once (one block per (sequence, kv head) serves the whole query-head group); f32 online
 * softmax, deterministic (fixed merge order, no atomics).  head_dim 64 or 128, num_heads / num_kv_heads in {1, 2, 4}; anything
 * else is RPO_ERR_UNSUPPORTED.
 * Backward: one pass over K / V; writes EVERY row of dk / dv ([T, num_kv_heads, hd], token strides given: e.g. the two halves of
 * one fused d(k|v) buffer) and dq [num_seqs, num_heads, hd]; out / dout / lse as produced / received by the forward. */
int rpo_lastq_attn_fwd(const void* q, int64_t q_stride, const void* k, const void* v, int64_t k_stride, int64_t v_stride,
                       const int* cu_seqlens, int64_t num_seqs, int64_t num_heads, int64_t num_kv_heads, int64_t head_dim,
                       float scale, void* out, int64_t out_stride, float* lse, rpo_stream_t stream);
int rpo_lastq_attn_bwd(const void* q, int64_t q_stride, const void* k, const void* v, int64_t k_stride, int64_t v_stride,
                       const int* cu_seqlens, int64_t num_seqs, int64_t num_heads, int64_t num_kv_heads, int64_t head_dim,
                       float scale, const void* out, int64_t out_stride, const void* dout, int64_t dout_stride, const float* lse,
                       void* dq, int64_t dq_stride, void* dk, void* dv, int64_t dk_stride, int64_t dv_stride, rpo_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (8) exact top-k over score rows, merged chunk by chunk ("next" row f3: the k-selection of faiss.IndexFlatIP.search,
 * reference src/utils.py:58-80).  scores: [rows, cols] (row stride ld elements) of the chunk whose first column is corpus
 * row col0; best_val f32 [rows, k] / best_idx int64 [rows, k]: the winners so far, best first (value descending, ties by
 * the smaller corpus index); first != 0: best_* hold nothing yet.  Unfilled slots are (-inf, INT64_MAX).  k <= 1024.
 * --------------------------------------------------------------------------------------------- */
int rpo_topk_merge(const void* scores, int64_t ld, int64_t rows, int64_t cols, int64_t col0, int k, int dtype,
                   float* best_val, int64_t* best_idx, int first, rpo_stream_t stream);
/* The same with `split` winner lists per score row: list r * split + s collects columns [s seg, (s + 1) seg) of row r, seg =
 * ceil(cols / split) rounded up to whole 16-byte vectors; best_val / best_idx: [rows * split, k].  A search of a FEW query rows
 * (256, the reference's faiss_search batch: utils.py:58-80) is one block per list: split = 4 fills the 256 CUs four times over
 * where one list per row leaves each CU one block.  The caller merges the `split` lists of a row once, at the end of the search
 * (value descending, ties by the smaller index: the same order).  split == 1 is rpo_topk_merge. */
int rpo_topk_merge_split(const void* scores, int64_t ld, int64_t rows, int64_t cols, int64_t col0, int k, int dtype, int split,
                         float* best_val, int64_t* best_idx, int first, rpo_stream_t stream);

/* The search step with the k-selection FUSED into the scoring kernel (round 6; bf16 operands -- dtype RPO_DT_BF16 --, d a multiple of 64, both operands below
 * 4 GB): the [Q, P] score matrix of the query block against one corpus chunk (p: rows [col0, col0 + P) of the corpus) is never written.
 * A score -- <q_r, p_c> accumulated in f32, rounded to bf16 once: bit for bit what the eval similarity (rpo_infonce_fwd without
 * temperature) stores -- that beats row r's current k-th winner, (best_val, best_idx)[r k + k - 1] in the order of rpo_topk_merge,
 * is appended to row r's candidate list: slot = atomic increment of cand_cnt[r]; cand_val f32 / cand_idx int64 [Q, cap]; a slot
 * >= cap is dropped (the counter still counts it).  cand_cnt must be zero on entry (rpo_topk_merge_candidates leaves it so).
 * The caller's winners must be FULL (k real entries per row: e.g. the first chunk went through rpo_topk_merge), otherwise every
 * column passes.
 * rpo_topk_merge_candidates: row r's min(cand_cnt[r], cap) candidates + its k winners -> its k winners (same order); resets
 * cand_cnt[r]; sets *overflow = 1 if any list had dropped entries (the caller then redoes the search through the score matrix:
 * retrieval.FlatIPIndex.search).  k + cap <= 4096.
 * rpo_sim_topk_filter_ok(Q, P, d): 1 for the shapes the filter takes -- those rpo_infonce_fwd scores with the same 256 x 256 kernel
 * frame (more than 64 query rows, d % 64 == 0, >= 192 tiles, operands below 4 GB), so that the fused step and the score-matrix path
 * agree bit for bit; everything else returns RPO_ERR_UNSUPPORTED and stays on the score-matrix path. */
int rpo_sim_topk_filter_ok(int64_t Q, int64_t P, int64_t d);
int rpo_sim_topk_filter(const void* q, const void* p, int64_t Q, int64_t P, int64_t d, int dtype, int64_t col0, int k,
                        int round_scores, const float* best_val, const int64_t* best_idx, float* cand_val, int64_t* cand_idx,
                        int32_t* cand_cnt, int cap, rpo_stream_t stream);
/* round_scores = 0 and rpo_sim_scores_f32: the search of an F32 index (the reference's faiss.IndexFlatIP dtype, utils.py:38-51) whose
 * embeddings are exactly representable in bf16 (dtype RPO_DT_BF16) or fp16 (RPO_DT_F16; f32 scores only) -- what an encoder that
 * computes in bf16 / fp16 hands over (modeling.py:533-539).  Products of such values are exact in f32 and the sums are f32 sums: an
 * f32 inner product in this kernel's summation order, at the 16-bit MFMA rate (16 x the f32 MFMA's).  The score is the unrounded
 * f32 sum; rpo_sim_scores_f32 writes the [Q, P] score matrix (row stride ldc floats) of the same frame for the search's first
 * chunk, so that equal rows score equal wherever they lie.  Same shapes as the filter. */
int rpo_sim_scores_f32(const void* q, const void* p, int64_t Q, int64_t P, int64_t d, int dtype, float* scores, int64_t ldc,
                       rpo_stream_t stream);
int rpo_topk_merge_candidates(const float* cand_val, const int64_t* cand_idx, int32_t* cand_cnt, int64_t rows, int cap, int k,
                              float* best_val, int64_t* best_idx, int32_t* overflow, rpo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RANKPO_HIP_H */
