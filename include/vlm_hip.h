/* vlm_hip.h -- C ABI of libvlm_hip.so: the MI355X (gfx950) hot path of ylsung/vl-merging.
 *
 * The reference (/root/reference) is 100 % Python and has no native boundary of its own; every entry
 * point below replaces a stock-PyTorch op site of the reference's hot path and cites it (file:line under
 * /root/reference/src).  Conventions:
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless named *_host;
 *   - every call is stream-ordered on `stream` (a hipStream_t passed as void*), re-entrant, allocates
 *     nothing: the caller owns outputs and workspaces;
 *   - return 0 on success, a negative VLM_ERR_* otherwise; nothing throws across the boundary;
 *   - process-wide state is limited to (1) diagnostic switches read ONCE from the environment at first use (VLM_GEMM_BIG,
 *     VLM_GEMM_BIGT, VLM_GEMM_STAGE, VLM_GEMM_SPLITK, VLM_GEMM_SPLITK_SLOTS, VLM_GEMM_GROUP_M, VLM_GEMM_BIG_GROUP_M,
 *     VLM_GEMM_TAIL_SPLIT, VLM_GEMM_CUS, VLM_ATT_DB_GROUPS, VLM_MERGE_VARIANT: thread-safe function-local statics,
 *     immutable afterwards) and (2) the hooks vlm_gemm_set_big_tile_mode and vlm_set_cu_budget (one atomic int each).  None changes results
 *     beyond the fp32 summation order of a GEMM or of the bias-table gradient.
 *
 * Token layout ("segment-major"): a pass over B samples with n0 text and n1 image tokens per sample keeps
 * activations as a [rows, D] matrix whose rows are  base0 + b*n0 + t  (t < n0)  and  base1 + b*n1 + (t-n0).
 * Modality-specific experts (all_moe) then see contiguous row ranges (vision_transformer.py:607-681 slices
 * and torch.cat's per layer instead).
 */
#ifndef VLM_HIP_H
#define VLM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VLM_OK 0
#define VLM_ERR_ARG (-1)
#define VLM_ERR_LAUNCH (-2)
#define VLM_ERR_WORKSPACE (-3)
#define VLM_ERR_UNSUPPORTED (-4)

#define VLM_ABI_VERSION 10
int vlm_abi_version(void);
/* Number of compute units grid sizing and split-K slice counts plan for, or negative error: the current device's count,
 * or the smaller budget set by VLM_GEMM_CUS=n (environment, read once) / vlm_set_cu_budget(n) -- room for RCCL's kernels
 * in a data-parallel job (0 or negative: back to the device's count).  Results do not depend on it beyond fp32 summation
 * order (the number of K slices of a wgrad). */
int vlm_device_cus(void);
int vlm_set_cu_budget(int cus);
/* Measurement tool (no reference counterpart; run.py:263-288's DDP gets its contention from NCCL): `workgroups` workgroups of
 * `threads` threads that each allocate `lds_bytes` of LDS and spin for `microseconds` on `stream` -- the single-GPU stand-in for
 * the compute units a gradient collective's kernels hold during backward (ddp.FlatGradReducer(standin=...)). */
int vlm_debug_occupy(int workgroups, int threads, int lds_bytes, int microseconds, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Checkpoint merge (K12/K13/K14-bias): modules/vilt_module.py:533-638 (merge_weights),
 * :640-746 (sum_task_vectors), :436-457/:486-529 (regmean's plain averages).
 * One job = one output tensor.  Arithmetic is fp32, unfused (no FMA), in source order:
 *   VLM_MERGE_LERP    : acc = 0;        acc = acc + ratio[m] * src[m]          (:590-601)
 *   VLM_MERGE_TASKVEC : acc = base;     acc = acc + ratio[m] * (src[m] - acc)  (:700-706, in-place alias
 *                       of the central tensor => recurrence, SURVEY.md 8a checklist item 8)
 *   VLM_MERGE_MEAN    : acc = 0;        acc = acc + src[m];  acc / n_src       (:436-457)
 * dst may alias base.  Bit-exact with the reference's CPU path.
 */
#define VLM_MERGE_LERP 0
#define VLM_MERGE_TASKVEC 1
#define VLM_MERGE_MEAN 2
#define VLM_MERGE_MAX_SRC 4

typedef struct {
  void* dst;                            /* f32 [n_elem] */
  const void* base;                     /* f32 [n_elem], TASKVEC only (central weight) */
  const void* src[VLM_MERGE_MAX_SRC];   /* f32 [n_elem] each */
  float ratio[VLM_MERGE_MAX_SRC];
  int32_t n_src;
  int32_t mode;
  uint64_t n_elem;
} vlm_merge_job_t;

/* Bytes of device workspace a plan for n_jobs jobs over total_elems elements needs. */
size_t vlm_merge_plan_bytes(int n_jobs, uint64_t total_elems);
/* Build the chunk table on the host and copy jobs + table into `workspace`.  The source is a temporary pageable host
 * buffer, so the call SYNCHRONISES `stream` before it returns (once per merge plan; vlm_merge_run is asynchronous). */
int vlm_merge_plan_upload(const vlm_merge_job_t* jobs_host, int n_jobs, void* workspace, size_t workspace_bytes,
                          void* stream);
/* Run an uploaded plan: ONE kernel launch over all jobs. */
int vlm_merge_run(const void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * bf16 MFMA GEMM with fused epilogue (K1/K6/K8/K9/K10): replaces F.linear at
 * modules/vision_transformer.py:335 (qkv + cat(q_bias,0,v_bias)), :360 (proj), :291/:295 (fc1/fc2),
 * LayerScale + residual at :586/:603 (x + drop_path(gamma * branch)), heads.py:14,27,36,49, and the
 * autograd backward of each (dgrad / wgrad).
 *   C[M,N] = epilogue( op(A)[M,K] . op(B)[K,N] ),   fp32 accumulation
 *   ta=0: A is [M][K] (K contiguous)   ta=1: A is [K][M]
 *   tb=0: B is [N][K] (nn.Linear weight layout)   tb=1: B is [K][N]
 *   v   = alpha*acc + bias[n]
 *   act = NONE/GELU : if aux != NULL, aux[m,n] = bf16(v) (pre-activation / pre-LayerScale branch output)
 *         GELU      : v = gelu_erf(v)               (vision_transformer.py:292, exact erf GELU)
 *         GELU_BWD  : v = v * gelu_erf'(aux[m,n])   (aux is an INPUT: the saved pre-activation)
 *         GELU_DERIV: aux[m,n] = bf16(gelu_erf'(v)); v = gelu_erf(v)   (aux REQUIRED: the forward pass saves the derivative --
 *                     one erf / exp evaluation serves both -- so that the backward epilogue is a multiplication)
 *         MUL_AUX   : v = v * aux[m,n]              (aux is an INPUT: the derivative GELU_DERIV saved)
 *   out = residual[m,n] + row_scale[m] * col_scale[n] * v     (each factor optional)
 *   C   = out  (bf16 or f32)  or  C += out (f32, accumulate != 0)
 *   col_sum[n] += sum_m C[m,n] (optional, fp32 atomics, value before the bf16 rounding): the bias gradient of the
 *         layer whose dY this GEMM produces (fc1 bias from the GELU-backward dgrad), saving a pass over dY
 * Requirements: lda, ldb multiples of 8; ldc, ld_aux, ld_res multiples of 4; 16-B aligned bases; K % 64 == 0
 * unless both operands are K-strided (ta=1 and tb=1).  Ragged M and N are handled in hardware.
 */
#define VLM_ACT_NONE 0
#define VLM_ACT_GELU 1
#define VLM_ACT_GELU_BWD 2
#define VLM_ACT_GELU_DERIV 3
#define VLM_ACT_MUL_AUX 4

typedef struct {
  const float* bias;       /* f32 [N] or NULL */
  const float* col_scale;  /* f32 [N] or NULL (LayerScale gamma) */
  const float* row_scale;  /* f32 [M] or NULL (DropPath keep-mask / keep_prob per row) */
  const float* residual;   /* f32 [M, ld_res] or NULL */
  int64_t ld_res;
  void* aux;               /* bf16 [M, ld_aux] or NULL */
  int64_t ld_aux;
  int32_t act;
  float alpha;
  int32_t accumulate;
  int32_t reserved;
  float* col_sum;          /* f32 [N] or NULL: accumulated column sums of the values written to C */
  float* col_sum_ws;       /* optional with col_sum (N % 128 == 0): f32 [M/128][2][N]; complete 128-row tiles store their
                              column sums at [tile][0][:] instead of adding them to col_sum -- the caller folds rows
                              0 .. M/128-1 into col_sum with vlm_colreduce_batch; the ragged last tile still adds directly */
  float* splitk_ws;        /* optional scratch for ta = tb = 1 (wgrad) calls: with it the reduction over K may be cut into
                              slices that store fp32 tiles [slice][M][N] here and are added into C by a second launch on
                              the same stream (no float atomics); NULL or too small: the atomic split-K path.  The caller
                              keeps one scratch per stream that runs such calls */
  uint64_t splitk_ws_bytes;
} vlm_epilogue_t;

int vlm_gemm_bf16(int ta, int tb, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                  int ldc, int c_is_f32, const vlm_epilogue_t* epi, void* stream);
/* Grouped form of the ta = tb = 0 call (K9): row ranges of ONE activation matrix go through DIFFERENT weights -- the
 * modality experts of an all_moe block act on the text rows and on the image rows of the segment-major token matrix
 * (modules/vision_transformer.py:607-681: x[:, :max_text_len] through mlp['l'] / attn['l'], the rest through ['v'], then
 * torch.cat) -- in one launch: the groups' 256-row tiles form one grid, so the small expert's tiles fill the large one's
 * rounds instead of under-filling a launch of their own.  A, C, residual, aux and row_scale are the WHOLE matrices
 * (a group's rows are [row0, row0 + rows) of each); bias, col_sum and col_sum_ws come per group (col_sum_ws rows are
 * counted from the group's row0) and must be NULL in `epi`, as must splitk_ws; accumulate is not supported.  Groups are
 * ascending and disjoint; empty groups are allowed.  Shapes the 256x256 kernel does not serve run as one vlm_gemm_bf16
 * call per group on the same stream: same results either way. */
#define VLM_GEMM_MAX_GROUPS 4
typedef struct {
  int32_t row0, rows;
  const void* B;        /* bf16 [N][ldb]: this group's weight (nn.Linear layout) */
  int32_t ldb;
  int32_t reserved;
  const float* bias;    /* f32 [N] or NULL */
  float* col_sum;       /* f32 [N] or NULL */
  float* col_sum_ws;    /* as vlm_epilogue_t.col_sum_ws: f32 [rows/128][2][N] */
} vlm_gemm_group_t;
int vlm_gemm_bf16_grouped(int n_groups, const vlm_gemm_group_t* groups, int N, int K, const void* A, int lda, void* C,
                          int ldc, int c_is_f32, const vlm_epilogue_t* epi, void* stream);
/* Grouped weight gradients (the autograd backward of the grouped call): for every group of TOKEN rows [row0, row0 + rows) of
 * A = dY [tokens][lda] and B = X [tokens][ldb] (both bf16, token-major: ta = tb = 1 of vlm_gemm_bf16),
 *   C_g[M,N] (+)= A[rows_g]^T . B[rows_g]      (f32, ldc; accumulate per group)
 * in one launch of the 256x256 wgrad kernel plus one reduce launch: the K slices of all groups share one round of
 * workgroups, each group's share following its token count.  splitk_ws as in vlm_epilogue_t (without it, or for shapes the
 * kernel does not serve: one vlm_gemm_bf16 call per group on the same stream; an empty group leaves C_g unchanged when
 * accumulating and zeroes it otherwise). */
typedef struct {
  int32_t row0, rows;
  float* C;
  int32_t accumulate;
  int32_t reserved;
} vlm_wgrad_group_t;
int vlm_gemm_wgrad_grouped(int n_groups, const vlm_wgrad_group_t* groups, int M, int N, const void* A, int lda,
                           const void* B, int ldb, int ldc, float* splitk_ws, uint64_t splitk_ws_bytes, void* stream);
/* Which kernel serves ta = tb = 0 calls: 0 the 128x128 tile always, 1 by shape (default; VLM_GEMM_BIG in the environment),
 * 2 the 256x256 tile whenever the call is legal for it, 3 by shape with the last partial round of tiles handed to the
 * 128x128 kernel as a second launch over the remaining rows, -1 back to the environment.  Tests and benchmarks only. */
int vlm_gemm_set_big_tile_mode(int mode);

/* ------------------------------------------------------------------------------------------------
 * Row-wise kernels (K5 + LayerScale backward + bias gradients).  D % 4 == 0, D <= 1024.
 * LayerNorm: nn.LayerNorm(eps=1e-6) modules/vision_transformer.py:831, Block.apply_ln :495-523 (a
 * modality-specific LayerNorm is one call per contiguous row range in the segment-major layout);
 * eps=1e-12 for BertEmbeddings / the MLM transform (modules/vilt_module.py:63, heads.py:43).
 *   fwd: y = (x-mean)*rstd*gamma+beta (bf16 or f32), stats[m] = {mean, rstd} (may be NULL)
 *   bwd: dx = dLN/dx (+ dres if given: the residual-path gradient, fused add); dgamma/dbeta are
 *        ACCUMULATED (atomicAdd) so grads of a weight shared by several passes add up in place.
 */
int vlm_layernorm_fwd(const float* x, int ldx, int M, int D, const float* gamma, const float* beta, float eps,
                      void* y, int ldy, int y_is_f32, float* stats, void* stream);
/* workspace (optional, f32, >= VLM_ROW_WS_BYTES(D)): per-workgroup partial column sums folded by a second tiny
 * launch; without it the column sums fall back to (heavily contended) float atomics.
 * deferred_blocks (optional, host int*): when non-NULL the fold is NOT launched; the call stores the number of partial
 * rows it wrote to `workspace` and the caller folds several such workspaces later with ONE vlm_colreduce_batch
 * (a transformer block's backward has four row kernels: three launches saved per block evaluation). */
#define VLM_ROW_WS_BYTES(D) ((size_t)1536 * 2 * (size_t)(D) * sizeof(float))
int vlm_layernorm_bwd(const void* dy, int lddy, int dy_is_f32, const float* x, int ldx, const float* stats,
                      const float* gamma, int M, int D, const float* dres, int lddres, float* dx, int lddx,
                      float* dgamma, float* dbeta, float* workspace, size_t workspace_bytes, int* deferred_blocks,
                      void* stream);
/* vlm_layernorm_bwd followed by vlm_layerscale_bwd on the row it has just produced, in one pass: `dx` (the residual-stream
 * gradient in front of this LayerNorm) is also the gradient arriving at the LayerScale of the branch below it
 * (vision_transformer.py:586,:603), so that branch's dy / dgamma / dbias are taken while the row is in registers -- the same
 * operations in the same order as the two calls, bit for bit.  workspace / workspace_bytes in `scale`: its own partial
 * region (a second fold job when deferred: [deferred_blocks][2][D], dgamma then dbias). */
typedef struct {
  const void* y;          /* bf16 [M, D]: the branch's saved output (incl. its bias); NULL = the LayerScale is folded into the
                             branch's projection (vlm_layerscale_fold): gamma and dgamma must be NULL too, dy = bf16(row_scale dx) */
  int32_t ldy;
  const float* gamma;     /* LayerScale vector [D] or NULL (ones) */
  const float* row_scale; /* DropPath row factors [M] or NULL */
  void* dy;               /* bf16 [M, D] out */
  int32_t lddy;
  float* dgamma;          /* += sum_m row_scale*dx*y, or NULL */
  float* dbias;           /* += sum_m dy, or NULL */
  float* workspace;
  size_t workspace_bytes;
} vlm_layerscale_t;
int vlm_layernorm_bwd_scale(const void* dy, int lddy, int dy_is_f32, const float* x, int ldx, const float* stats,
                            const float* gamma, int M, int D, const float* dres, int lddres, float* dx, int lddx,
                            float* dgamma, float* dbeta, float* workspace, size_t workspace_bytes,
                            const vlm_layerscale_t* scale, int* deferred_blocks, void* stream);
#define VLM_MAX_FOLD_JOBS 16
typedef struct {
  const float* partials; /* workspace written by a deferred row kernel: [nblocks][2][D] */
  int32_t nblocks;
  int32_t D;
  float* out0;           /* += sum_b partials[b][0][:]  (dgamma) or NULL */
  float* out1;           /* += sum_b partials[b][1][:]  (dbeta / dbias) or NULL */
} vlm_fold_job_t;
int vlm_colreduce_batch(const vlm_fold_job_t* jobs_host, int n_jobs, void* stream);
/* Backward of x_new = x + row_scale[m]*gamma[n]*y[m,n] (vision_transformer.py:586,:603) w.r.t. the branch:
 *   dy = bf16(row_scale*gamma*dx); dgamma[n] += sum_m row_scale*dx*y; dbias[n] += sum_m dy.
 * y_bf16 NULL (dgamma NULL): only dy and its column sums -- the folded form of vlm_layerscale_fold (gamma normally NULL). */
int vlm_layerscale_bwd(const float* dx, int lddx, const void* y_bf16, int ldy, const float* gamma,
                       const float* row_scale, int M, int D, void* dy_bf16, int lddy, float* dgamma, float* dbias,
                       float* workspace, size_t workspace_bytes, int* deferred_blocks, void* stream);
/* LayerScale folded into the branch's output projection (vision_transformer.py:489-491, :586, :603:
 * x = x + drop_path(gamma * branch(x)), the branch ending in attn.proj / mlp.fc2).  gamma (.) (a W^T + b) = a W'^T + b' with
 * W' = diag(gamma) W and b' = gamma (.) b, so the forward / dgrad / wgrad GEMMs run on folded operands and the saved copy of
 * the branch output (and the O(M N) pass over it) goes.  One job = one weight:
 *   vlm_layerscale_fold:    shadow[n,k] = bf16(gamma[n] weight[n,k]);  bias_out[n] = gamma[n] bias[n]      (after every update)
 *   vlm_layerscale_finish:  with the RAW sums raw_w = g^T a (= dL/dW') and raw_b = colsum(g) (= dL/db'), g = bf16(row_scale dx):
 *                           dweight[n,:] += gamma[n] raw_w[n,:];  dbias[n] += gamma[n] raw_b[n];
 *                           dgamma[n] += sum_k weight[n,k] raw_w[n,k] + bias[n] raw_b[n]   (atomic: experts share gamma);
 *                           raw_w, raw_b are ZEROED (the call composes with gradient accumulation).
 * gamma NULL = ones; bias / bias_out / raw_b / dbias / dgamma may be NULL; weight, raw_w, dweight 16-byte aligned, K % 4 == 0. */
#define VLM_MAX_LAYERSCALE_JOBS 32
typedef struct {
  const float* weight;  /* fp32 master [N, K] */
  const float* gamma;   /* [N] */
  const float* bias;    /* [N] or NULL */
  void* shadow;         /* fold: bf16 [N, K] out */
  float* bias_out;      /* fold: [N] out or NULL */
  float* raw_w;         /* finish: [N, K] in, zeroed */
  float* raw_b;         /* finish: [N] in, zeroed, or NULL */
  float* dweight;       /* finish: [N, K] accumulate */
  float* dbias;         /* finish: [N] accumulate or NULL */
  float* dgamma;        /* finish: [N] accumulate (atomic) or NULL */
  int32_t N, K;
} vlm_layerscale_job_t;
int vlm_layerscale_fold(const vlm_layerscale_job_t* jobs_host, int n_jobs, void* stream);
int vlm_layerscale_finish(const vlm_layerscale_job_t* jobs_host, int n_jobs, void* stream);
/* out[n] += sum_m a[m,n] (bf16 a; N % 8 == 0): bias gradients of qkv / fc1 / heads. */
int vlm_colsum_bf16(const void* a, int lda, int M, int N, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Cross-entropy of the MLM head (modules/objectives.py:88-143: F.cross_entropy(mlm_logits.view(-1, vocab), mlm_labels.view(-1),
 * ignore_index=-100)) on the bf16 logits [rows, ld] the decoder GEMM wrote (V valid columns, ld % 8 == 0), fp32 arithmetic:
 *   fwd: lse[r] = logsumexp(logits[r, :V]);  loss_rows[r] = lse[r] - logits[r, labels[r]], or 0 where labels[r] == ignore_index
 *        (the caller forms the mean over the rows that count);
 *   bwd: dlogits[r, c] = scale_dev[0] * (exp(logits[r, c] - lse[r]) - [c == labels[r]]) as bf16 for c < V, 0 for V <= c < ld_d
 *        and in ignored rows: written whole, padding included, so the buffer is the decoder dgrad / wgrad GEMMs' operand as it
 *        stands.  scale_dev: device scalar (upstream gradient / number of counted rows), no host round trip. */
int vlm_cross_entropy_fwd(const void* logits_bf16, int ld, int rows, int V, const int64_t* labels, int64_t ignore_index,
                          float* loss_rows, float* lse, void* stream);
int vlm_cross_entropy_bwd(const void* logits_bf16, int ld, int rows, int V, const int64_t* labels, int64_t ignore_index,
                          const float* lse, const float* scale_dev, void* dlogits_bf16, int ld_d, void* stream);

/* Round 6: the rest of the loss tail (the [B, *] algebra of modules/objectives.py that was ~120 torch-native launches per step).
 *   vlm_l2norm_fwd / _bwd: y = x / ||x||_2 per row in fp32 (the contrastive heads' feature normalisation, objectives.py:248-300,
 *        vilt_module.py:1329-1375 `/ norm(dim=-1, keepdim=True)`); backward dx = (g - y (g . y)) / ||x|| in x's dtype.
 *   vlm_contrastive: the symmetric cross-entropy of compute_ifm / compute_irtr (objectives.py:274-300, :393-445) on gathered,
 *        normalised features [n, D] (this rank's B rows first, as the reference re-inserts them, :277-286), s = exp(log_scale[0]):
 *        logits [n, n] = s img txt^T; out3 = (loss, d loss / d log_scale, s); d_img / d_txt [B, D] = gradients of the OWN rows.
 *        ws: vlm_contrastive_ws_floats(n) floats.
 *   vlm_small_cross_entropy: F.cross_entropy(logits [rows, V], labels) (mean) and dlogits [rows, V] fp32 = its gradient
 *        (compute_itm_hardneg, objectives.py:239-245: [3B, 2] logits).
 *   vlm_cross_entropy_reduce: out2 = (sum of the counted rows' losses / count, 1 / count) after vlm_cross_entropy_fwd.
 *   vlm_scale_by_scalar: out_k = in_k * scalar_dev[0] for up to 4 buffers in one launch (a scalar loss's upstream gradient). */
int vlm_l2norm_fwd(const void* x, int x_is_bf16, int ld, int rows, int D, float* y, float* inv_norm, void* stream);
int vlm_l2norm_bwd(const float* g, const float* y, const float* inv_norm, int rows, int D, void* dx, int dx_is_bf16, int ld, void* stream);
size_t vlm_contrastive_ws_floats(int n);
int vlm_contrastive(const float* all_img, const float* all_txt, int n, int B, int D, const float* log_scale, float* logits, float* out3,
                    float* d_img, float* d_txt, float* ws, void* stream);
int vlm_small_cross_entropy(const void* logits, int is_bf16, int ld, int rows, int V, const int64_t* labels, float* loss, float* dlogits,
                            void* stream);
int vlm_cross_entropy_reduce(const float* loss_rows, const int64_t* labels, int rows, int V, int64_t ignore_index, float* out2, void* stream);
int vlm_scale_by_scalar(const float* const* in, float* const* out, const int* n, int count, const float* scalar_dev, void* stream);

/* Round 6: the two ends of a pass as single launches (csrc/frontops.hip).
 *   vlm_text_rows_fwd: BertEmbeddings.forward + the modality type row (reference vilt_module.py:51-63, :1111-1113):
 *        out[r] = dropout(LayerNorm(word[ids[r]] + add0; gamma, beta, eps)) + add1, fp32 rows (leading dimension ld_out: the rows of
 *        the pass's token matrix), stats[r] = (mean, rstd).  u: optional fp32 [n, D]; an element is kept where u >= p and scaled by
 *        `scale` (nn.Dropout(p): scale = 1 / (1 - p)).  D % 4 == 0, D <= 1024.
 *   vlm_text_rows_bwd: g = gradient of those rows; adds into d_word[ids[r]] (rows with ids == padding_idx excepted: nn.Embedding's
 *        padding_idx), d_add1, d_beta, d_gamma, d_add0 (each may be null).  ws: vlm_text_rows_bwd_ws_floats(D) floats.
 *   vlm_image_rows_prep: out2[0] = conv_bias + type_row (the patch-embed GEMM's bias), out2[1] = cls + type_row (the lead row) --
 *        visual_embed's cls concat + token_type_embeddings add, vision_transformer.py:952-991, vilt_module.py:1114-1117.
 *   vlm_image_lead_rows: x[b * rows] = lead for b < B.
 *   vlm_image_rows_bwd: g fp32 [B * rows, D] -> g16 bf16 (lead rows zero: the wgrad GEMM's operand); d_bias += column sums over the
 *        patch rows, d_type_row += over all rows, d_cls += over the lead rows.  ws: vlm_image_rows_bwd_ws_floats(D) floats.
 *   vlm_tanh_fwd: y fp32 = tanh(x bf16) (Pooler, heads.py:8-19).
 *   vlm_act_bwd: dy bf16 [M, Np] (columns >= N zero) = g * act'(.): mode 0 exact GELU from the saved bf16 pre-activation (MLM
 *        transform, heads.py:36-46), mode 1 tanh from the saved fp32 output, mode 2 no activation (cast + padding; saved unused).
 *   vlm_colsum_small: out[c] += sum_r a[r][c], N <= 64 columns (ITMHead.fc bias gradient: N = 2).
 *   vlm_sample_negatives: idx[0][i] ~ softmax(sim_a[i, :]) without entry i, idx[1][i] likewise from sim_b (objectives.py:176-229:
 *        F.softmax, fill_diagonal_(0), torch.multinomial(., 1)), inverse CDF on the uniforms u[2][B]; strides in elements (logits_per_text
 *        is the transposed view of logits_per_image).
 *   vlm_weighted_sum: out[0] = sum_k weights[k] * terms[k][0] (device scalars, count <= 8): the loss sum of training_step. */
int vlm_text_rows_fwd(const int64_t* ids, int n, const float* word, int ld_word, const float* add0, const float* gamma, const float* beta,
                      float eps, const float* u, float p, float scale, const float* add1, float* out, int ld_out, float* stats, int D,
                      void* stream);
size_t vlm_text_rows_bwd_ws_floats(int D);
int vlm_text_rows_bwd(const float* g, int ld_g, const int64_t* ids, int n, const float* word, int ld_word, const float* add0,
                      const float* gamma, const float* stats, const float* u, float p, float scale, int D, float* d_word,
                      int64_t padding_idx, float* d_add1, float* d_beta, float* d_gamma, float* d_add0, float* ws, void* stream);
int vlm_image_rows_prep(const float* conv_bias, const float* type_row, const float* cls, int D, float* out2, void* stream);
int vlm_image_lead_rows(float* x, int ld_x, int B, int rows, int D, const float* lead, void* stream);
size_t vlm_image_rows_bwd_ws_floats(int D);
int vlm_image_rows_bwd(const float* g, int ld_g, int B, int rows, int D, void* g16, float* d_bias, float* d_type_row, float* d_cls, float* ws,
                       void* stream);
int vlm_tanh_fwd(const void* x_bf16, int ld_x, int M, int N, float* y, void* stream);
int vlm_act_bwd(const void* g, int g_is_f32, int ld_g, const void* saved, int ld_saved, int mode, int M, int N, int Np, void* dy_bf16,
                void* stream);
int vlm_colsum_small(const void* a_bf16, int lda, int M, int N, float* out, void* stream);
int vlm_sample_negatives(const float* sim_a, int a_row_stride, int a_col_stride, const float* sim_b, int b_row_stride, int b_col_stride, int B,
                         int n, const float* u, int64_t* idx, void* stream);
int vlm_weighted_sum(const float* const* terms, const float* weights, int count, float* out, void* stream);

/* vlm_scatter_rows: dx [R, D] (bf16 or fp32) = sum over the sources of g_k [count_k, D] placed at rows first_row + j * row_step, zero
 *        elsewhere -- the gradient of the row views a pass's result exposes (text_feats = x[:B T], the cls rows x[b T], a leading block
 *        of samples: vilt_module.py:1136-1156, objectives.py:97 `infer["text_feats"]`, heads.py:17 `hidden_states[:, 0]`) in one pass
 *        instead of autograd's zero fill + strided copy per view and an addition per extra view.  D % 4 == 0, at most 4 sources. */
typedef struct {
  const void* g; /* [count, D] rows, bf16 or fp32 */
  int g_is_f32;
  int ld;        /* row stride of g in elements */
  int first_row, row_step, count;
} vlm_scatter_src_t;
int vlm_scatter_rows(void* dx, int dx_is_f32, int ld_dx, int R, int D, const vlm_scatter_src_t* src, int n_src, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Flat-buffer elementwise kernels.
 * vlm_adamw_step: transformers-4.x AdamW as instantiated at modules/vilt_utils.py:314-317
 *   (betas=(0.9, beta_2), eps=1e-8, bias-corrected step_size computed by the caller, decoupled decay applied
 *   AFTER the Adam update); also writes the bf16 shadow of the parameters (GEMM operand) and may zero the grad.
 * vlm_patch_im2col: front end of PatchEmbed (vision_transformer.py:714-728): NCHW fp32 image -> bf16 patch rows
 *   [B*(lead_rows + patches), 3*P*P] (column = c*P*P + i*P + j); lead_rows zero rows per image hold the place of
 *   the cls token (:974-975).
 */
int vlm_adamw_step(float* p, float* g, float* m, float* v, void* p_bf16, uint64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, float step_size, float grad_scale, int zero_grad,
                   void* stream);
int vlm_cast_f32_bf16(const float* src, void* dst_bf16, uint64_t n, void* stream);
/* Backward of the word-embedding gather (HF BertEmbeddings.word_embeddings = nn.Embedding(vocab, D, padding_idx),
 * vilt_module.py:63, :1090; what torch's embedding_dense_backward + grad accumulation compute):
 * dW[ids[t], 0..D) += gy[t, 0..D) for every t in [0, n) with ids[t] != padding_idx (pass -1 for none).  ids are int64; a token
 * whose id lies outside [0, vocab) is skipped (nothing is written out of bounds); dW is ACCUMULATED (float atomics: the sum order of tokens sharing an id is not fixed). */
int vlm_embedding_bwd(const float* gy, int ld, const int64_t* ids, int64_t n, int D, int64_t padding_idx, float* dW, int ld_w,
                      int64_t vocab, void* stream);
/* Transposed bf16 weight shadows.  The backward of F.linear (dX = dY . W, modules/vision_transformer.py:291/:295/:335/
 * :360 through autograd) reads W [N_out, K_in] along its strided axis; with W^T [K_in, N_out] kept next to W the dgrad
 * is an ordinary K-contiguous GEMM (both operands by LDS-DMA).  One launch refreshes every listed matrix: the device
 * table holds one entry per 64x64 tile (src [rows, cols] row-major -> dst [cols, rows] row-major). */
typedef struct {
  uint64_t src;      /* bf16 device pointer */
  uint64_t dst;      /* bf16 device pointer */
  int32_t rows, cols;
  int32_t tile_row, tile_col;
} vlm_transpose_tile_t;
int vlm_transpose_bf16_tiles(const vlm_transpose_tile_t* tiles_dev, int n_tiles, void* stream);

/* DropPath (timm drop_path, modules/vision_transformer.py:446 used at :586/:603): out[row] = u[b] < keep ? 1/keep : 0
 * for every token row of sample b in the segment-major layout (text rows base0 + b*n0 + t, image rows
 * base1 + b*n1 + i); u = one uniform [0,1) draw per sample.  The result is the GEMM epilogue's row_scale. */
int vlm_droppath_rows(const float* u, float keep, int B, int n0, int n1, int base0, int base1, float* out, void* stream);
/* Every DropPath site of a pass at once: out[s][row] = u0[s][b] < keep[s] ? 1/keep[s] : 0 (image rows take u1[s][b] when u1
 * is given: two unimodal passes sharing one launch keep their own draws).  out is [n_sites, rows]. */
int vlm_droppath_sites(const float* u0, const float* u1, const float* keep, int n_sites, int B, int n0, int n1, int base0,
                       int base1, int rows, float* out, void* stream);
int vlm_patch_im2col(const float* image, void* patches_bf16, int B, int H, int W, int P, int lead_rows, void* stream);
/* ------------------------------------------------------------------------------------------------
 * float64 kernels of the merge side (csrc/f64ops.hip), all products on v_mfma_f64_16x16x4_f64.
 * Gram cache (K15, src/cache_gram_matrices.py:246-254: `G += X.to(float64)^T X` for the input X of every hooked
 * linear): vlm_gram_f64 adds X^T X of a bf16 or fp32 [M, D] activation matrix into the fp64 [D, D] accumulator on the
 * device (exact conversion, fp64 FMA chains, upper-triangular tiles mirrored on the way out).
 * RegMean (K14, src/vilt/modules/vilt_module.py:388-392, 407-434): vlm_scale_gram_f64 forms a*G + (1-a)*diag(G)
 * (optionally accumulating the sum over modalities); vlm_gemm_f64 is C = alpha*op(A) op(B) + beta*C in fp64 (A may be the
 * fp32 checkpoint weight); the reference's torch.inverse of the SPD sum is replaced by a blocked Cholesky
 * factorisation + two triangular solves built from vlm_potrf_block_f64 (in-place lower factor of one <= 64-wide
 * diagonal block; *status gets 1 + the index of a non-positive pivot), vlm_trsm_block_f64 (X <- X op(L)^-1 for a
 * row panel against one diagonal block; trans = 1: X L^T = B, trans = 0: X L = B) and vlm_gemm_f64 for the block
 * updates.  vlm_cholesky_f64 (in-place lower factor of a contiguous SPD [n, n] matrix; *status as above, checked by the
 * caller whenever it chooses to synchronise) and vlm_solve_spd_right_f64 (rhs [rows, ld] <- rhs (L L^T)^-1 in place) walk the
 * block columns inside the library: one call each per merged weight.  vlm_accumulate_f32_f64 (dst += src) is kept for
 * callers that already hold an fp32 product. */
int vlm_accumulate_f32_f64(const float* src, double* dst_f64, uint64_t n, void* stream);
int vlm_gram_f64(const void* x, int ldx, int M, int D, int x_is_f32, double* gram, void* stream);
int vlm_gemm_f64(int ta, int tb, int M, int N, int K, double alpha, const void* A, int lda, int a_is_f32, const double* B,
                 int ldb, double beta, double* C, int ldc, void* stream);
int vlm_scale_gram_f64(const double* src, double* dst, int n, double alpha, int accumulate, void* stream);
int vlm_potrf_block_f64(double* A, int lda, int j0, int nb, int* status, void* stream);
int vlm_trsm_block_f64(const double* L, int ldl, int l0, int nb, int trans, double* B, int ldb, int rows, int c0, void* stream);
int vlm_cholesky_f64(double* A, int n, int* status, void* stream);
int vlm_solve_spd_right_f64(const double* chol, int n, double* rhs, int ld, int rows, void* stream);
/* The same factorisation / solve for `count` (<= 64) matrices of ONE shape in lock step: every block step is ONE launch over all of
 * them (RegMean's 36 solves of 768^2 and 12 of 3072^2, vilt_module.py:432-434: ~420 launches instead of ~5 000).  A_list / chol_list /
 * rhs_list: HOST arrays of device pointers; status: device int[count], zero on entry, verdict per matrix as in vlm_cholesky_f64.
 * Per matrix bit-identical to the unbatched calls. */
int vlm_gemm_f64_batched(int ta, int tb, int M, int N, int K, double alpha, const void* const* A_list, int lda, int a_is_f32,
                         const double* const* B_list, int ldb, double beta, double* const* C_list, int ldc, int count, void* stream);
int vlm_cholesky_f64_batched(double* const* A_list, int count, int n, int* status, void* stream);
int vlm_solve_spd_right_f64_batched(double* const* chol_list, int n, double* const* rhs_list, int ld, int rows, int count, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused attention (K4 + K7 + K7b): softmax(scale*Q K^T + bias[h] + key mask) V, head_dim = 64.
 * Replaces Attention.forward's core, modules/vision_transformer.py:346-358, the bias materialisation
 * get_rel_pos_bias modules/vilt_module.py:1061-1064 (never formed here), and the text/image block-diagonal
 * split of separate_plain_forward / moe_forward (vision_transformer.py:567-584, :619-637).
 *   qkv       bf16 [rows, 3*H*64] = F.linear output of :335 as is (q | k | v thirds, head-major); the kernel
 *             applies `scale` (:346) to the scores.
 *   bias_t    f32 [n_cols, R]: TRANSPOSE of relative_position_bias_table [R, heads*layers]; row head_row0+h
 *             is head h of this layer.  NULL = no bias.
 *   rel_index / rel_index_t  int16 [index_rows, ld_index] and its transpose (BOTH are needed: each kernel reads the
 *             orientation whose fast axis runs along its lanes, so every index load is a coalesced 64-B row piece;
 *             the forward reads only rel_index_t): 4 x relative-position index (= byte offset into the fp32 table column,
 *             R <= 8191) in "index coordinates": text token t is position t, image token i is position pos1 + i
 *             (pos1 % 8 == 0, ld_index % 4 == 0).
 *   keep0/1   uint8 [B, n0] / [B, n1] key keep flags (text_masks; NULL = keep all), masked_fill(-inf) of :354.
 *   rows      segment-major: text (b,t) -> base0 + b*n0 + t ; image (b,i) -> base1 + b*n1 + i.
 *   mode      JOINT: every query sees text then image keys.  SEPARATE: queries see their own segment only.
 * Forward writes out bf16 [rows, H*64] (the layout :358 reshapes to) and lse f32 [H, total_rows] (log2 domain,
 * consumed by the backward).  Backward: delta = rowsum(dO*O), then dK/dV (+ the bias-table gradient, accumulated
 * into dbias_t f32 [n_cols, R]) and dQ, written into dqkv bf16 [rows, 3*H*64] (dq already carries `scale`).
 */
#define VLM_ATTN_JOINT 0
#define VLM_ATTN_SEPARATE 1

typedef struct {
  const void* qkv;
  int32_t ld_qkv;
  int32_t H;
  int32_t total_rows;
  int32_t R;
  const float* bias_t;
  const int16_t* rel_index;    /* [q position][key position] */
  const int16_t* rel_index_t;  /* its transpose [key position][q position], same leading-dimension rules */
  int32_t ld_index;
  int32_t index_rows;
  int32_t ld_index_t;
  int32_t index_t_rows;
  int32_t head_row0;
  int32_t mode;
  const uint8_t* keep0;
  const uint8_t* keep1;
  int32_t B, n0, n1, base0, base1, pos1;
  float scale;
  int32_t reserved;
  /* Dense bias (vlm_bias_dense), REQUIRED when bias_t is set: fp16 log2(e) * bias_t[c][rel_index[q][k]/4] (the
   * kernels' score accumulators are base-2 exponents) for every (layer, head) column c, stored as 4-KiB tiles in MFMA operand order (one tile
   * per 32 stationary x 64 streamed positions, vlm_bias_dense_bytes() per column); positions that are not valid
   * members of a tile's segment hold a large negative value, so ragged tiles, the text/image gap and -- in SEPARATE
   * mode -- the foreign segment mask themselves.  bias_dense: stationary = query (forward, dQ); bias_dense_t:
   * stationary = key (dK/dV).  The tables depend on (n0, n1, pos1, mode); the kernels add the slice
   * [head_row0 + h] to the score accumulators on the matrix pipe.  The bias is identical for every sample of the
   * batch, so a layer's slices stay cache-resident.  The bias-table GRADIENT still goes through rel_index. */
  const void* bias_dense;
  const void* bias_dense_t;
  int32_t dense_tiles; /* vlm_bias_dense_bytes(n0, n1, pos1, mode) / 4096 */
  int32_t reserved2;
} vlm_attn_desc_t;

/* Dense relative-position bias for all heads and layers at once (the reference's get_rel_pos_bias,
 * modules/vilt_module.py:1061-1064): out = n_cols columns of vlm_bias_dense_bytes() each.  index: int16 [pos1 + n1,
 * ld_index] relative-position index (4 x index, as in vlm_attn_desc_t.rel_index); k_major selects bias_dense_t. */
size_t vlm_bias_dense_bytes(int n0, int n1, int pos1, int mode);
int vlm_bias_dense(const float* bias_t, int n_cols, int R, const int16_t* index, int ld_index, int n0, int n1, int pos1,
                   int mode, int k_major, void* out_f16, void* stream);

int vlm_attention_fwd(const vlm_attn_desc_t* d, void* out_bf16, int ld_out, float* lse, void* stream);
/* Optional fused bias gradients: dq[s] / dv[s] (f32 [H*64], may be NULL) are ACCUMULATED with the column sums of dQ /
 * dV over the rows of segment s (0 = text rows, 1 = image rows; modality experts own different q_bias / v_bias,
 * vision_transformer.py:335), taken while the tiles are still in registers instead of re-reading dqkv. */
typedef struct {
  float* dq[2];
  float* dv[2];
} vlm_attn_colsum_t;

/* delta_ws: f32 scratch of ws_floats elements: at least H * total_rows (rowsum(dO*O)).  With a bias table,
 * vlm_attention_bwd_ws_floats(d, 0) = 3 * H * total_rows also holds the C operands (-lse / (scale log2 e), -delta) of the
 * 16-wave bias-table-gradient kernel (a smaller scratch selects the 8-wave kernel it replaced), and with
 * vlm_attention_bwd_ws_floats(d, 1) elements the per-workgroup histograms are summed by a second small launch instead of
 * global atomics (order-independent result).  dbias_t (f32 [n_cols, R], may be NULL) is ACCUMULATED.  colsum may be NULL. */
size_t vlm_attention_bwd_ws_floats(const vlm_attn_desc_t* d, int with_dbias);
int vlm_attention_bwd(const vlm_attn_desc_t* d, const void* out_bf16, int ld_out, const void* dout_bf16, int ld_dout,
                      const float* lse, float* delta_ws, size_t ws_floats, void* dqkv_bf16, int ld_dqkv, float* dbias_t,
                      const vlm_attn_colsum_t* colsum, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VLM_HIP_H */
