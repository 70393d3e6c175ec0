/*
 * merlin_hip.h -- C ABI of libmerlin_hip.so: the MI355X (gfx950) hot path behind the
 * Merlin Models `mm` Python surface.
 *
 * The reference (NVIDIA-Merlin/models) is pure Python over TensorFlow and has NO FFI of its
 * own: every entry point below replaces the TensorFlow op(s) a reference call site dispatches
 * to.  Each declaration cites that call site (paths relative to the reference checkout).
 *
 * Conventions (all entry points):
 *   - return int32 status: MH_OK (0) or a negative MH_ERR_*; mh_last_error() returns a
 *     thread-local human-readable message for the last failing call on this thread.
 *   - every data pointer is a DEVICE pointer owned by the caller (row-major, contiguous unless a
 *     leading dimension is given); arguments documented as "HOST array" are small host arrays
 *     (of device pointers / sizes) that are consumed before the call returns.
 *   - the library allocates no persistent device memory; workspaces are caller-provided and
 *     sized by the matching mh_*_workspace_bytes() query.
 *   - every launch is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and ordered with respect to that stream only.
 *   - no exceptions cross the ABI, no torch / C++ types appear in signatures.
 *   - all floating point data is IEEE fp32; ids are int32 or int64 (MH_I32 / MH_I64).
 */
#ifndef MERLIN_HIP_H
#define MERLIN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_OK 0
#define MH_ERR_INVALID_ARGUMENT (-1)
#define MH_ERR_UNSUPPORTED (-2)
#define MH_ERR_LAUNCH (-3)
#define MH_ERR_WORKSPACE (-4)

/* id dtypes */
#define MH_I32 0
#define MH_I64 1

/* activations, keras names: linear / relu / sigmoid (blocks/mlp.py:35-139) */
#define MH_ACT_NONE 0
#define MH_ACT_RELU 1
#define MH_ACT_SIGMOID 2
/* activations that run as their own element-wise layer (mh_activation), not in a GEMM epilogue */
#define MH_ACTX_TANH 10
#define MH_ACTX_ELU 11
#define MH_ACTX_SELU 12
#define MH_ACTX_SOFTPLUS 13
#define MH_ACTX_SWISH 14
#define MH_ACTX_GELU 15
#define MH_ACTX_LEAKY_RELU 16
#define MH_ACTX_RELU6 17

/* sequence combiners (inputs/embedding.py:432-441, 1545-1587) */
#define MH_COMBINER_SUM 0
#define MH_COMBINER_MEAN 1
#define MH_COMBINER_SQRTN 2
#define MH_COMBINER_MAX 3 /* dense lists only: process_str_sequence_combiner "max" (inputs/embedding.py:1579-1580) */

/* sparse optimizers for the embedding backward (models/base.py:1121-1174; blocks/optimizer.py) */
#define MH_OPT_SGD 0
#define MH_OPT_ADAGRAD 1
#define MH_OPT_ADAM 2 /* dense: keras Adam; sparse rows: LazyAdam (blocks/optimizer.py:342-437) */

/* max features per gather call (pointer tables travel as kernel arguments) */
#define MH_MAX_FEATURES 64

typedef void* mh_stream_t;

/* ---- library --------------------------------------------------------------------------- */
int32_t mh_version(void);
const char* mh_last_error(void);
/* number of compute units / XCDs of the current device, for grid sizing by callers */
int32_t mh_device_info(int32_t* cu_count, int64_t* hbm_bytes);

/* ---- a1/a3: categorical embedding lookup ------------------------------------------------
 * Replaces keras Embedding / tf.gather in EmbeddingTable._call_table
 * (merlin/models/tf/inputs/embedding.py:424-471, one-hot branch :458-461) and
 * EmbeddingFeatures.lookup_feature (:1126-1156), for ALL features of a batch in ONE launch.
 * Feature f reads ids[f][b] (b < B) from table tables[f] ([table_rows[f], D] fp32) and writes
 *   out[b * out_row_stride + out_offset[f] + d]   (d < D; offsets in floats, multiples of 4)
 * i.e. directly into the stacked [B, F_total, D] layout StackFeatures would produce
 * (core/aggregation.py:101-108) when out_row_stride = F_total*D and out_offset = rank*D of the
 * sorted feature name -- or into the ConcatFeatures layout (aggregation.py:54-66).
 * Ids outside [0, table_rows[f]) produce a zero row (TF-GPU gather semantics).
 * D must be a multiple of 4 and <= 1024; F <= MH_MAX_FEATURES. */
int32_t mh_embedding_gather_fwd(const float* const* tables /*HOST [F]*/,
                                const int64_t* table_rows /*HOST [F]*/,
                                const void* const* ids /*HOST [F] of device ptrs*/,
                                int32_t ids_dtype, int64_t B, int32_t F, int32_t D,
                                float* out, int64_t out_row_stride,
                                const int64_t* out_offset /*HOST [F]*/, mh_stream_t stream);

/* Multi-hot / ragged lookup with a string combiner: replaces
 * tf.nn.safe_embedding_lookup_sparse(W, sp_ids, None, combiner)
 * (inputs/embedding.py:432-441, :1131-1139).  CSR input: values[nnz], offsets[B+1] (int32 or
 * int64, same dtype as values).  Empty bags -> zero row; ids < 0 are pruned (safe_* semantics);
 * ids >= rows contribute zero but are counted (TF-GPU gather).  mean divides by the number of
 * kept ids, sqrtn by its sqrt.  nnz = number of entries of values (host-known shape).
 * out[b * out_row_stride + d]. */
int32_t mh_embedding_bag_fwd(const float* table, int64_t rows, const void* values, int64_t nnz,
                             const void* offsets, int32_t ids_dtype, int64_t B, int32_t D,
                             int32_t combiner, float* out, int64_t out_row_stride,
                             mh_stream_t stream);

/* Backward of the two list lookups (ragged CSR above, dense [B, L] below) with the optimizer fused in, as
 * in mh_embedding_gather_bwd: the gradient of `tf.nn.safe_embedding_lookup_sparse` /
 * `process_str_sequence_combiner` w.r.t. the table is an IndexedSlices over the nnz values with rows
 * grad[b] / div(b), div = 1 | kept | sqrt(kept) for sum | mean | sqrtn (kept = non-pruned ids of bag b; every
 * position for a dense list) -- a division, like the gradient of the forward's div_no_nan.  Pruned (< 0) and out-of-range ids receive no update.
 * offsets == NULL selects the dense list of length L (nnz must equal B*L).  nnz < 2^26.
 * MAX (dense lists only): the gradient of tf.reduce_max -- grad[b, d] goes to the positions whose row attains the maximum
 * of component d, split equally among ties (the rows are re-gathered from the not-yet-updated table). */
/* The expansion half on its own: gexp[nnz, D] (row j = the gradient row of value j: grad[bag(j)] / div(bag(j)), or the
 * max-combiner split; `table` is only read by MAX).  Callers that must fold SEVERAL lookups of one table into a single
 * dedup + optimizer step -- a one-hot feature and a list feature that share a table (item_id / item_id_history), the
 * IndexedSlices of which Keras sums before ONE apply (models/base.py:1121-1174) -- expand every list, concatenate values
 * and rows with the one-hot ids / gradient rows, and call mh_embedding_gather_bwd once.  scale_ws: B floats (unused by MAX). */
int32_t mh_embedding_bag_expand(const float* table, int64_t rows, const void* values, int64_t nnz, const void* offsets,
                                int64_t L, int32_t ids_dtype, int64_t B, int32_t D, int32_t combiner, const float* grad,
                                int64_t grad_row_stride, float* gexp, float* scale_ws, mh_stream_t stream);
int64_t mh_embedding_bag_bwd_workspace_bytes(int64_t B, int64_t nnz, int32_t D);
int32_t mh_embedding_bag_bwd(float* table, float* state, float* state2, int64_t rows, const void* values,
                             int64_t nnz, const void* offsets, int64_t L, int32_t ids_dtype, int64_t B, int32_t D,
                             int32_t combiner, const float* grad, int64_t grad_row_stride, int32_t optimizer,
                             float lr, float eps, float beta1, float beta2, const float* lr_device,
                             void* workspace, int64_t workspace_bytes, mh_stream_t stream);
/* F list features over F DISTINCT tables of one width in ONE update (the embedding variables of a model's multi-hot columns:
 * inputs/embedding.py:398-428 looks each one up, models/base.py:1121-1174 applies every IndexedSlices in one optimizer step):
 * the combiner divisors and the bag index of every value in two launches, then one sort / segmented reduce / fused optimizer over
 * all F x max(nnz) values that reads a value's gradient row THROUGH its bag index -- no expanded [nnz, D] gradient is written.
 * values[f] / offsets[f]: device pointers of feature f (offsets == NULL: every feature is a dense [B, L] list);
 * grad: [B, grad_row_stride] floats, feature f's [B, D] block starts grad_offset[f] floats into the row.  Same terms and optimizer
 * arithmetic as F calls of mh_embedding_bag_bwd; a run of equal ids may be cut into partial sums at other places (a few ulp).
 * Combiners SUM / MEAN / SQRTN. */
int64_t mh_embedding_bag_bwd_multi_workspace_bytes(int64_t B, int64_t max_nnz, int32_t F, int32_t D);
int32_t mh_embedding_bag_bwd_multi(float* const* tables, float* const* state, float* const* state2, const int64_t* table_rows,
                                   const void* const* values, const int64_t* nnz, const void* const* offsets, int64_t L,
                                   int32_t ids_dtype, int64_t B, int32_t F, int32_t D, int32_t combiner, const float* grad,
                                   int64_t grad_row_stride, const int64_t* grad_offset, int32_t optimizer, float lr, float eps,
                                   float beta1, float beta2, const float* lr_device, void* workspace, int64_t workspace_bytes,
                                   mh_stream_t stream);

/* Dense fixed-length list [B, L] with a string combiner over axis 1 (mean / sum):
 * process_str_sequence_combiner (inputs/embedding.py:1556-1587): every position counts
 * (id 0 is NOT padding-aware).  combiner: SUM, MEAN or MAX (elementwise maximum over the L rows). */
int32_t mh_embedding_dense_list_fwd(const float* table, int64_t rows, const void* ids /*[B,L]*/,
                                    int32_t ids_dtype, int64_t B, int32_t L, int32_t D,
                                    int32_t combiner, float* out, int64_t out_row_stride,
                                    mh_stream_t stream);

/* Backward of the one-hot lookup fused with the sparse optimizer step (the IndexedSlices
 * apply of BaseModel.train_step, models/base.py:1121-1174).  grad[b * grad_row_stride +
 * grad_offset[f] + d] is dL/d out (same layout as the forward).  Features whose table pointers
 * coincide share one table (shared embeddings): their gradients are summed before the update.  Duplicate ids in a batch are summed BEFORE the update
 * (Keras _deduplicate_indexed_slices semantics).
 *   SGD:     W[r] -= lr * g[r]
 *   ADAGRAD: acc[r] += g[r]^2 ; W[r] -= lr * g[r] / (sqrt(acc[r]) + eps)   (keras Adagrad)
 *   ADAM:    LazyAdam._resource_apply_sparse (blocks/optimizer.py:412-437), touched rows only:
 *            m[r] = b1 m[r] + (1-b1) g[r]; v[r] = b2 v[r] + (1-b2) g[r]^2; W[r] -= lr_t m[r]/(sqrt(v[r])+eps)
 *            with lr_t = lr sqrt(1-b2^t)/(1-b1^t) supplied by the caller (host value `lr`, or the device
 *            scalar `lr_device` maintained by mh_adam_tick so that a captured hipGraph replays correctly).
 * state[f] (Adagrad accumulator / Adam m) and state2[f] (Adam v) have the table's shape; NULL when unused.
 * workspace: mh_embedding_bwd_workspace_bytes(B, F, D) bytes.  Limits: F < 64, B < 2^26, B*F < 2^31, D % 4 == 0,
 * D <= 1024.  Runs whose gradients fit one 16-entry piece are summed in sorted (sample) order -> reproducible;
 * hot rows spanning several pieces are combined with float atomics (order-dependent in the last bits). */
/* Deterministic mode of the fused sparse update (process-wide; initial value from MERLIN_HIP_DETERMINISTIC=1 at load time):
 * crossing runs are walked in sample order instead of summed with float atomics -- bit-reproducible, slower on very hot rows. */
int32_t mh_set_deterministic(int32_t on);
/* Arithmetic of the in-batch scorer (mh_inbatch_softmax_fwd without logits, _fwd_dq, _bwd) at E = 128, mode 2 also at E = 64 and with the
 * logQ corrections (mh_scorer_split.hip):
 *   0 (the library's initial value) exact fp32 MFMA, every score one k-ascending fmaf chain;
 *   1 "bf16x3": every fp32 operand split into two bf16 values, every product of both GEMMs of a pass formed as hi hi + hi lo + lo hi
 *     on v_mfma_f32_32x32x16_bf16 with fp32 accumulators (error of a dot product <= ~2.3e-6 |q| |item| measured; 16 / 3 of the fp32
 *     MFMA rate).  NOT fp32-grade: opt-in (MERLIN_HIP_SCORER_ARITH=bf16x3 through the Python layer);
 *   2 "bf16x6": three bf16 pieces per operand (they hold the 24-bit significand exactly), six terms h h + h m + m h + h l + l h + m m
 *     per product, dropped terms <= 2^-25 of it: as close to the real dot product as the fp32 chain (16 / 6 of the fp32 MFMA rate).
 *     The host side (models_amd/ops.py) selects this mode unless MERLIN_HIP_SCORER_ARITH says f32 or bf16x3. */
int32_t mh_set_scorer_arith(int32_t mode);

/* ---- a9 on the split-bf16 GEMM: the three GEMMs of a full-rank DCN-v2 cross layer (Cross.call, blocks/cross.py:188-202) ----
 * mh_cross_layer_fwd_split = mh_cross_layer_fwd / _fwd_save (p_out may be NULL), mh_cross_layer_bwd_split = mh_cross_layer_bwd
 * (same phases, selected by the non-NULL outputs) with every fp32 product formed from bf16 pieces on the bf16 MFMA (fp32 accumulators;
 * operands split -- and, for dW, transposed -- once per call into the workspace).  d % 4 == 0 (zero-padded layer).
 * mh_set_gemm_arith(mode) selects WHICH split the *_split entry points compute in: 2 = six terms "bf16x6" (x = h + m + l, products
 * h h + h m + m h + h l + l h + m m; dropped terms <= 2^-25 of a product: fp32-grade -- what the Python layer sets by default);
 * 1 (and the initial 0) = three terms "bf16x3" (x = hi + lo, hi hi + hi lo + lo hi; 2^-17 per operand: MERLIN_HIP_GEMM_ARITH=bf16x3).
 * The exact fp32 kernels are the entry points WITHOUT the _split suffix (MERLIN_HIP_GEMM_ARITH=f32 through the Python layer). */
int32_t mh_set_gemm_arith(int32_t mode);
int64_t mh_cross_layer_split_workspace_bytes(int64_t M, int32_t d);
int32_t mh_cross_layer_fwd_split(const float* x0, const float* x, const float* W, const float* b, int64_t M, int32_t d,
                                 float* out, float* p_out, void* workspace, int64_t workspace_bytes, mh_stream_t stream);
int32_t mh_cross_layer_bwd_split(const float* x0, const float* x, const float* p, const float* dout, const float* W,
                                 int64_t M, int32_t d, float* g, float* dx0_acc, int32_t accumulate_dx0, float* dx,
                                 float* dW, float* db, void* workspace, int64_t workspace_bytes, mh_stream_t stream);
/* Dense layer (blocks/mlp.py:275-280) on the same split-bf16 GEMM: the contracts of mh_linear_bias_act_fwd / _bwd (same argument
 * meaning, dy overwritten with dz, dx masked by x_act, dW / db optional) with the three GEMMs in the arithmetic mh_set_gemm_arith selected; the workspace
 * (mh_linear_split_workspace_bytes) is needed by every phase.  Meant for wide layers (N >= 256: whole 256 x 256 output tiles). */
int64_t mh_linear_split_workspace_bytes(int64_t M, int32_t K, int32_t N);
int32_t mh_linear_bias_act_fwd_split(const float* x, int64_t ldx, const float* W, const float* b, int64_t M, int32_t K, int32_t N,
                                     int32_t act, float* y, int64_t ldy, void* workspace, int64_t workspace_bytes, mh_stream_t stream);
int32_t mh_linear_bias_act_bwd_split(const float* x, int64_t ldx, const float* W, const float* y, int64_t ldy, float* dy, int64_t lddy,
                                     int64_t M, int32_t K, int32_t N, int32_t act, int32_t x_act, float* dx, int64_t lddx, float* dW,
                                     float* db, void* workspace, int64_t workspace_bytes, mh_stream_t stream);
/* Tower layers -- Dense layers with N = 128 outputs, 32 <= K <= 1024, at batch >= 4096 (tf/blocks/mlp.py:275-280: the DLRM's 415 -> 128, the
 * two-tower's 256 -> 128) -- on the bf16 matrix pipe in the SIX-term split "bf16x6" (mh_tower_split.hip): x = h + m + l (three bf16 pieces of
 * the 24-bit significand), products h h + h m + m h + h l + l h + m m with fp32 accumulators; dropped terms <= 2^-25 |x y|, i.e. below the
 * rounding of an fp32 product: fp32-grade accuracy (NOT the fmaf chain's bits), 16 / 6 of the fp32 MFMA rate.  Same arguments as
 * mh_linear_bias_act_fwd / _bwd plus a workspace (mh_tower_workspace_bytes: the operand images of W).  mh_tower_supported: 1 when the shape
 * is covered.  Replaces the same Keras Dense call sites as mh_linear_bias_act_fwd / _bwd (tf/blocks/mlp.py:270-330). */
int32_t mh_tower_supported(int64_t M, int32_t K, int32_t N);
int64_t mh_tower_workspace_bytes(int64_t M, int32_t K, int32_t N);
int32_t mh_tower_linear_fwd(const float* x, int64_t ldx, const float* W, const float* b, int64_t M, int32_t K, int32_t N, int32_t act,
                            float* y, int64_t ldy, void* workspace, int64_t workspace_bytes, mh_stream_t stream);
int32_t mh_tower_linear_dx(const float* dz, int64_t lddz, const float* W, int64_t M, int32_t K, int32_t N, float* dx, int64_t lddx,
                           void* workspace, int64_t workspace_bytes, mh_stream_t stream);
int64_t mh_embedding_bwd_workspace_bytes(int64_t B, int32_t F, int32_t D);
int32_t mh_embedding_gather_bwd(float* const* tables /*HOST [F]*/, float* const* state /*HOST [F]*/,
                                const int64_t* table_rows /*HOST [F]*/,
                                const void* const* ids /*HOST [F]*/, int32_t ids_dtype,
                                int64_t B, int32_t F, int32_t D, const float* grad,
                                int64_t grad_row_stride, const int64_t* grad_offset /*HOST [F]*/,
                                int32_t optimizer, float lr, float eps, float* const* state2 /*HOST [F]*/,
                                float beta1, float beta2, const float* lr_device, void* workspace,
                                int64_t workspace_bytes, mh_stream_t stream);

/* The two halves of mh_embedding_gather_bwd as separate calls.  _prepare needs the ids only (segmented sort + piece list,
 * left in the workspace): a caller issues it at the START of the step on a second stream, beside the forward pass, and the
 * sort leaves the critical path (142 us of the 430 us launch at the C2 shapes).  _apply -- same tables / table_rows / ids /
 * B / F / D and the SAME, untouched workspace, after the gradient exists -- does the segmented reduce + fused optimizer and
 * the carried runs.  mh_embedding_gather_bwd == _prepare followed by _apply. */
int32_t mh_embedding_gather_bwd_prepare(float* const* tables /*HOST [F]: identity of shared tables only*/,
                                        const int64_t* table_rows, const void* const* ids, int32_t ids_dtype, int64_t B,
                                        int32_t F, int32_t D, void* workspace, int64_t workspace_bytes, mh_stream_t stream);
int32_t mh_embedding_gather_bwd_apply(float* const* tables, float* const* state, const int64_t* table_rows,
                                      const void* const* ids, int32_t ids_dtype, int64_t B, int32_t F, int32_t D,
                                      const float* grad, int64_t grad_row_stride, const int64_t* grad_offset,
                                      int32_t optimizer, float lr, float eps, float* const* state2, float beta1,
                                      float beta2, const float* lr_device, void* workspace, int64_t workspace_bytes,
                                      mh_stream_t stream);

/* l2_batch_regularization_factor of EmbeddingTable (inputs/embedding.py:463-464): the layer adds
 * factor * sum(out^2) over the batch of looked-up rows to the loss.  out[B, D] / grad[B, D] are views with row strides
 * ld_out / ld_grad (e.g. one slot of the stacked [B, F, D] buffers): grad += 2 factor out (skipped if grad == NULL),
 * loss_accum[0] += factor * sum(out^2) (fixed summation order).  workspace: 256 floats. */
int32_t mh_l2_batch_reg(const float* out, int64_t ld_out, float* grad, int64_t ld_grad, int64_t B, int32_t D,
                        float factor, float* loss_accum, float* workspace, mh_stream_t stream);

/* ---- a6: Dense layer  y = act(x W + b) -------------------------------------------------
 * Replaces keras Dense inside _Dense.call (blocks/mlp.py:275-280).  x[M, K] (leading dim ldx),
 * W[K, N] row-major (the Keras kernel layout), b[N] or NULL, y[M, N] (leading dim ldy).
 * fp32 in / fp32 accumulate on the f32 MFMA pipe; for N > 4 each output is one k-ascending fmaf
 * chain (N <= 4 heads use 16 interleaved partial chains + a shuffle tree). */
int32_t mh_linear_bias_act_fwd(const float* x, int64_t ldx, const float* W, const float* b,
                               int64_t M, int32_t K, int32_t N, int32_t act, float* y,
                               int64_t ldy, mh_stream_t stream);

/* Backward: given dy (leading dim lddy) and the forward OUTPUT y (for the activation
 * derivative), computes dz = dy * act'(y) in place of dy (skipped when act == NONE, e.g. when the
 * caller already holds dz), then
 *   dx[M,K] = (dz W^T) * x_act'(x)   (skipped if dx == NULL).  x_act names the activation whose
 *             OUTPUT is x (NONE for raw inputs): folding the producer's derivative into the dX
 *             epilogue hands the previous layer its dz directly, so it can be called with act = NONE;
 *   dW[K,N] = x^T dz,   db[N] = colsum(dz) (if db != NULL; fused into the dW kernel).
 * All reductions have a fixed order (deterministic).  workspace: mh_linear_bwd_workspace_bytes.
 * The three parts can be issued separately (dx and dW are independent once dz exists, so a caller may run
 * the dW pass on a second stream): dx == NULL skips dX, dW == NULL (then db == NULL, no workspace) skips
 * dW/db, and with both NULL only the in-place activation gradient is applied. */
int64_t mh_linear_bwd_workspace_bytes(int64_t M, int32_t K, int32_t N);
int32_t mh_linear_bias_act_bwd(const float* x, int64_t ldx, const float* W, const float* y,
                               int64_t ldy, float* dy, int64_t lddy, int64_t M, int32_t K,
                               int32_t N, int32_t act, int32_t x_act, float* dx, int64_t lddx,
                               float* dW, float* db, void* workspace, int64_t workspace_bytes,
                               mh_stream_t stream);

/* ---- a6 (chains): consecutive SMALL Dense layers in one launch ---------------------------
 * Replaces a run of _Dense.call (blocks/mlp.py:275-280) inside an MLPBlock (blocks/mlp.py:35-139) -- e.g. the DLRM
 * bottom MLP 13 -> 128 -> 64 and the tail of the top MLP with the BinaryOutput head 128 -> 64 -> 32 -> 1
 * (outputs/classification.py:114) -- when every width is <= 128:  y_l = act_l(y_{l-1} W_l + b_l), l = 1..L, y_0 = x.
 * dims[L+1] = {K_0, N_1, .., N_L}; W[l] is [dims[l], dims[l+1]] row-major (Keras kernel layout), b[l] is [dims[l+1]]
 * or NULL (b itself may be NULL); act[l] in MH_ACT_*; y[l] ([M, dims[l+1]], leading dim ldy[l]) receives every layer's
 * output (the backward needs them).  W / b / act / y / ldy / dims are HOST arrays of L entries.  Every output is one
 * k-ascending fmaf chain, bit-identical to mh_linear_bias_act_fwd layer by layer (which uses 16 partial chains for
 * N <= 4 heads: there the chain differs in the last bits).
 * mh_mlp_chain_supported: 1 when a fused kernel covers (L, dims) (L = 2 or 3, widths <= 128 within the compiled
 * signatures), else 0 -- _fwd / _bwd then return MH_ERR_UNSUPPORTED and the caller goes layer by layer. */
int32_t mh_mlp_chain_supported(int32_t L, const int32_t* dims);
int32_t mh_mlp_chain_fwd(const float* x, int64_t ldx, int64_t M, int32_t L, const int32_t* dims,
                         const float* const* W, const float* const* b, const int32_t* act, float* const* y,
                         const int64_t* ldy, mh_stream_t stream);
/* Backward of the chain (GradientTape through the same layers, models/base.py:1121-1174): g[M, dims[L]] (leading dim
 * ldg) is the gradient w.r.t. y_L, or already dz_L when pre_masked != 0 (e.g. the BCE-with-sigmoid gradient).  Per
 * layer, last to first: dz_l = g * act_l'(y_l), dW[l] = y_{l-1}^T dz_l, db[l] = colsum(dz_l) (db or db[l] may be NULL),
 * g = dz_l W_l^T.  dx (NULL = not needed) receives (dz_1 W_1^T) * x_act'(x) where x_act names the activation that
 * produced x (as in mh_linear_bias_act_bwd).  dz never touches HBM; dW / db are reduced in a fixed order
 * (deterministic).  dx is bit-identical to the layer-by-layer path; dW / db sum the batch in a different order. */
int64_t mh_mlp_chain_bwd_workspace_bytes(int64_t M, int32_t L, const int32_t* dims);
int32_t mh_mlp_chain_bwd(const float* x, int64_t ldx, int64_t M, int32_t L, const int32_t* dims,
                         const float* const* W, const int32_t* act, const float* const* y, const int64_t* ldy,
                         const float* g, int64_t ldg, int32_t pre_masked, int32_t x_act, float* dx, int64_t lddx,
                         float* const* dW, float* const* db, void* workspace, int64_t workspace_bytes,
                         mh_stream_t stream);
/* The same backward in two launches: _partial runs the strip kernel (dx -- what the previous layer's backward waits for -- and one
 * partial dW / db slab per workgroup, left in the workspace); _reduce sums the slabs into dW / db in the fixed order of
 * mh_mlp_chain_bwd (same bits) and can be issued later on the same stream (the weight gradients are first needed by the optimizer
 * step: base.py:1164-1174): a step issues it at its tail.  The workspace must not be reused in between. */
int32_t mh_mlp_chain_bwd_partial(const float* x, int64_t ldx, int64_t M, int32_t L, const int32_t* dims,
                                 const float* const* W, const int32_t* act, const float* const* y, const int64_t* ldy,
                                 const float* g, int64_t ldg, int32_t pre_masked, int32_t x_act, float* dx, int64_t lddx,
                                 void* workspace, int64_t workspace_bytes, mh_stream_t stream);
int32_t mh_mlp_chain_bwd_reduce(int64_t M, int32_t L, const int32_t* dims, float* const* dW, float* const* db, void* workspace,
                                int64_t workspace_bytes, mh_stream_t stream);

/* ---- a7: DLRM pairwise dot interaction --------------------------------------------------
 * Replaces tf.matmul(x, x, transpose_b=True) + strict-upper-triangle boolean_mask in
 * DotProductInteraction.call (blocks/interaction.py:86-116): x[B, F, D] ->
 * pairs[b, p(i,j)] = <x[b,i,:], x[b,j,:]> for i < j in row-major order, p = i(2F-i-1)/2 + j-i-1.
 * When tail != NULL, tail[b, 0:T] is the shortcut branch of the "concat" aggregation
 * (core/combinators.py:669-693 + aggregation.py:54-66) and lands in the same output row:
 *   tail_first != 0:  out[b] = [tail (T) | pairs (P)]   <- the reference's DLRM order: the shortcut is
 *       Filter("bottom_block") (blocks/dlrm.py:126-130), whose dict output keeps its key "bottom_block"
 *       (core/tabular.py:552-576); ParallelBlock.call merges dict-valued branches by `update`
 *       (core/combinators.py:564-569), the interaction branch is keyed by its Keras name "sequential_block_<n>", and
 *       ConcatFeatures concatenates in sorted-key order: "bottom_block" < "sequential_block...".  The torch twin agrees:
 *       cat((continuous, interactions)) (torch/blocks/dlrm.py:102-104);
 *   tail_first == 0:  out[b] = [pairs (P) | tail (T)]   (a shortcut whose key sorts after the block's name).
 * out has leading dimension ldo >= F(F-1)/2 + T.   F <= 32, D % 4 == 0, D <= 256. */
int32_t mh_dot_interaction_fwd(const float* x, int64_t B, int32_t F, int32_t D,
                               const float* tail, int64_t ld_tail, int32_t T, int32_t tail_first,
                               float* out, int64_t ldo, mh_stream_t stream);
/* Backward: dx[B,F,D] = (G + G^T) x with G the strict-upper-triangular matrix scattered from the P pair
 * columns of dout.  If tail_slot >= 0, the gradient of the shortcut copy (the T tail columns of dout, T <= D;
 * column layout as in the forward, selected by tail_first) is added to dx[:, tail_slot, :T] in the same pass. */
int32_t mh_dot_interaction_bwd(const float* x, const float* dout, int64_t ldo, int64_t B,
                               int32_t F, int32_t D, float* dx, int32_t tail_slot, int32_t T,
                               int32_t tail_first, mh_stream_t stream);

/* ---- a1 + a5 + a7 fused: DLRM gather -> interaction without the stacked [B, F, D] round trip ---
 * Replaces the whole `ParallelBlock{embeddings, bottom_block} -> StackFeatures -> DotProductInteraction
 * -> concat shortcut` segment of DLRMBlock (blocks/dlrm.py:110-130).  Slot s (s < F, sorted feature
 * order, core/aggregation.py:101-108) is either a categorical feature -- slot_tables[s] ([slot_rows[s], D]),
 * slot_ids[s] ([B] int32/int64) -- or the ONE dense slot (slot_tables[s] == NULL) whose row b is
 * dense[b * ld_dense .. + D] (the bottom-MLP output).  With append_dense the dense row is the shortcut part of the
 * output row: out[b] = [dense row (D) | pairwise dots (row-major i<j)] when tail_first (the reference's DLRM order,
 * see mh_dot_interaction_fwd), [pairs | dense row] otherwise; without it out[b] = pairs.
 * Needs D % 16 == 0 and F*D <= 2048.  HOST arrays as in mh_embedding_gather_fwd. */
int32_t mh_dlrm_interaction_fused_fwd(const float* const* slot_tables, const int64_t* slot_rows,
                                      const void* const* slot_ids, int32_t ids_dtype, const float* dense,
                                      int64_t ld_dense, int64_t B, int32_t F, int32_t D,
                                      int32_t append_dense, int32_t tail_first, float* out, int64_t ldo,
                                      mh_stream_t stream);
/* Backward of the fused segment: re-gathers the rows (tables are not yet updated), dx[B, F, D] = (G+G^T) X;
 * with tail_to_dense the gradient of the dense copy (the D shortcut columns of dout, placed as in the forward)
 * is added to the dense slot.  D in {16, 32, 64, 128}. */
int32_t mh_dlrm_interaction_fused_bwd(const float* const* slot_tables, const int64_t* slot_rows,
                                      const void* const* slot_ids, int32_t ids_dtype, const float* dense,
                                      int64_t ld_dense, const float* dout, int64_t ldo, int64_t B,
                                      int32_t F, int32_t D, int32_t tail_to_dense, int32_t tail_first,
                                      float* dx, mh_stream_t stream);

/* ---- a9: DCN-v2 cross layer  out = x0 * (x W + b) + x ----------------------------------
 * Replaces Cross.call (blocks/cross.py:188-202) with a full-rank kernel W[d, d]
 * (DenseMaybeLowRank, blocks/mlp.py:304-396, low_rank_dim=None). x0, x, out: [M, d]. */
int32_t mh_cross_layer_fwd(const float* x0, const float* x, const float* W, const float* b,
                           int64_t M, int32_t d, float* out, mh_stream_t stream);
/* The same layer under a gradient tape (BaseModel.train_step, models/base.py:1121-1174): additionally stores
 * p = x W + b into p_out [M, d]; the backward needs it (d loss / d x0 = dout * p) and would otherwise recompute the
 * whole d x d product (11 ms per layer at d = 3344, B = 64 K). */
int32_t mh_cross_layer_fwd_save(const float* x0, const float* x, const float* W, const float* b, int64_t M, int32_t d,
                                float* out, float* p_out, mh_stream_t stream);

/* Low-rank form W = U V (DCN-v2 Eq. 2; DenseMaybeLowRank with low_rank_dim = r, blocks/mlp.py:365-396):
 * h[M, r] = x U comes from mh_linear_bias_act_fwd (no bias, no activation); this call finishes the layer,
 * out = x0 * (h V + b) + x with V[r, d].  All matrices contiguous. */
int32_t mh_cross_layer_lowrank_fwd(const float* x0, const float* x, const float* h, const float* V, const float* b,
                                   int64_t M, int32_t d, int32_t r, float* out, mh_stream_t stream);

/* ---- a10: L2 row normalisation (transforms/regularization.py:26-80) --------------------
 * y = x / max(||x||_2, eps) per row  == tf.linalg.l2_normalize(x, axis=-1, epsilon=eps^2). */
int32_t mh_l2norm_rows(const float* x, int64_t M, int32_t N, float eps, float* y,
                       mh_stream_t stream);
/* Backward of the same (x is the forward INPUT): dx = (dy - y (y . dy)) / max(||x||, eps), and dx = dy / eps
 * for rows below the clamp.  x, dy, dx contiguous [M, N]. */
int32_t mh_l2norm_rows_bwd(const float* x, const float* dy, int64_t M, int32_t N, float eps, float* dx,
                           mh_stream_t stream);

/* tf.keras.layers.Activation for the Keras activation names MLPBlock accepts (blocks/mlp.py:35-139) beyond relu / sigmoid /
 * linear, which are fused into the GEMM epilogues: tanh, elu, selu, softplus, swish (silu), gelu (exact, erf), leaky_relu
 * (slope 0.2, tf.nn.leaky_relu's default), relu6.  dy == NULL: out = f(x).  dy != NULL: out = dy * f'(x) with x the layer's
 * SAVED INPUT (the pre-activation).  Row-major [M, N] operands, each with its own leading dimension; out may alias x or dy. */
int32_t mh_activation(int32_t act, const float* x, int64_t ldx, const float* dy, int64_t lddy, float* out, int64_t ldo,
                      int64_t M, int32_t N, mh_stream_t stream);

/* buf[m, col0 : col0 + ncols] = value for the M rows of a row-major buffer of pitch ld: the pad columns between a row's width
 * and its 16-byte aligned pitch (the reference has no such op: its tensors are dense, tf.concat at
 * merlin/models/tf/core/aggregation.py:54-66; the padded pitch is this library's layout).  A library launch rather than a
 * tensor-library fill so that a recorded step (mh_record_*) holds every launch of the step. */
int32_t mh_fill_columns(float* buf, int64_t M, int64_t ld, int32_t col0, int32_t ncols, float value, mh_stream_t stream);

/* Mean of n floats into mean[0] (Keras' SUM_OVER_BATCH_SIZE reduction of a per-sample loss, e.g. the softmax-CE rows of
 * the retrieval step, losses/listwise.py:38-52): two launches, fixed summation order (deterministic).  workspace: 256 floats. */
int32_t mh_mean(const float* x, int64_t n, float* mean, float* workspace, mh_stream_t stream);

/* DotProduct.call (outputs/base.py:307-310): out[m] = sum_n a[m,n] * b[m,n]  (positive scores;
 * the inference branch of ContrastiveOutput, outputs/contrastive.py:221). */
int32_t mh_rowwise_dot(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t M,
                       int32_t N, float* out, mh_stream_t stream);

/* ---- a11-a13: in-batch sampled-softmax scorer ------------------------------------------
 * Replaces ItemRetrievalScorer.call_outputs (blocks/retrieval/base.py:283-429) /
 * ContrastiveOutput.outputs (outputs/contrastive.py:276-344) + rescore_false_negatives
 * (utils/tf_utils.py:126-154) + LogitsTemperatureScaler (transforms/bias.py:65-68) +
 * CategoricalCrossEntropy(from_logits=True) (losses/listwise.py:38-52).
 *   pos[b]    = <q[b], item[b]>
 *   neg[b, j] = <q[b], neg_item[j]>, replaced by false_neg_score where pos_ids[b]==neg_ids[j]
 *               (skipped when pos_ids == NULL: downscore_false_negatives=False)
 *   logits[b] = [pos[b], neg[b, 0..Nn)] / temperature                     (fp32, [B, 1+Nn])
 *   loss[b]   = logsumexp(logits[b]) - logits[b, 0];  lse[b] = logsumexp(logits[b])
 * logits may be NULL (fused mode: nothing of size B*Nn is written); loss / lse may be NULL.
 * q, item: [B, E]; neg_item: [Nn, E] (pass item and Nn = B for in-batch negatives).
 * ids int32 or int64.  E % 4 == 0, E <= 1024.
 * logQ sampling correction (pos_logq[B], neg_logq[Nn]; both NULL = off): the reference subtracts
 * log(sampling probability + 1e-16) of each candidate from its score -- ContrastiveOutput.outputs with
 * logq_sampling_correction=True (outputs/contrastive.py:309-319: BEFORE the false-negative rescoring, so rescored
 * entries stay at false_neg_score; logq_after_mask = 0) or the PopularityLogitsCorrection `post` block
 * (transforms/bias.py:238-254: on every column AFTER the rescoring, scaled by reg_factor; logq_after_mask = 1).  The
 * caller passes the (scaled) log-probabilities; pos[b] -= pos_logq[b], neg[b, j] -= neg_logq[j].
 * workspace: mh_inbatch_softmax_workspace_bytes(B, Nn, E, pass) with pass 0 = _fwd, 1 = _bwd, 2 = _fwd_dq.
 * E <= 128 runs on the row-stationary streaming kernels (a workgroup keeps 256 rows of one matrix in registers and
 * streams the other through LDS by direct-to-LDS DMA); other E are zero-padded to 32 / 64 / 128 inside the
 * workspace (scores unchanged bit for bit).  E > 128 takes tiled kernels whose backward materialises ds[B, Nn]. */
int64_t mh_inbatch_softmax_workspace_bytes(int64_t B, int64_t Nn, int32_t E, int32_t pass);
int32_t mh_inbatch_softmax_fwd(const float* q, const float* item, const float* neg_item,
                               const void* pos_ids, const void* neg_ids, int32_t ids_dtype,
                               int64_t B, int64_t Nn, int32_t E, float temperature,
                               float false_neg_score, const float* pos_logq,
                               const float* neg_logq, int32_t logq_after_mask, float* logits, int64_t ld_logits,
                               float* loss, float* lse, void* workspace, int64_t workspace_bytes,
                               mh_stream_t stream);

/* Training-mode forward (the forward half of BaseModel.train_step, models/base.py:1121-1174, for the in-batch
 * softmax): loss / lse as above (fused mode, no logits) AND, from the same pass over the score tiles,
 *   dq[B,E]    = d (grad_scale * sum_b loss[b]) / d q     (complete, positive column included)
 *   ditem[B,E] = the positive-role part of the item gradient (may be NULL)
 * -- dq is a softmax-weighted sum of item rows, i.e. exactly an attention output with the items as values, so it
 * is accumulated flash-style while the log-sum-exp is still running (lazy rescaling against a reference max that
 * starts at the positive logit).  The caller then needs only the column pass of mh_inbatch_softmax_bwd
 * (dq = NULL).  E <= 128 only (MH_ERR_UNSUPPORTED otherwise: call _fwd then _bwd). */
int32_t mh_inbatch_softmax_fwd_dq(const float* q, const float* item, const float* neg_item,
                                  const void* pos_ids, const void* neg_ids, int32_t ids_dtype,
                                  int64_t B, int64_t Nn, int32_t E, float temperature,
                                  float false_neg_score, const float* pos_logq,
                               const float* neg_logq, int32_t logq_after_mask, float grad_scale, float* loss, float* lse,
                                  float* dq, float* ditem, void* workspace, int64_t workspace_bytes,
                                  mh_stream_t stream);

/* Backward of sum_b(loss[b]) * grad_scale (pass grad_scale = 1/B for the Keras mean) w.r.t.
 * q (dq[B,E]), item in its positive role (ditem[B,E], may be NULL) and neg_item (dneg_item[Nn,E]);
 * all are overwritten.  For in-batch negatives the caller adds ditem + dneg_item.
 * Flash-style for E <= 128: the probability tiles are recomputed from (q, neg_item, lse) inside the MFMA kernels
 * and contracted on the spot -- a row pass (q stationary) for dq / ditem and a column pass (neg_item stationary)
 * for dneg_item; nothing of size B x Nn is written and every output is written once, in a fixed order
 * (deterministic).  dq == NULL (then ditem == NULL) skips the row pass: use it after mh_inbatch_softmax_fwd_dq. */
int32_t mh_inbatch_softmax_bwd(const float* q, const float* item, const float* neg_item,
                               const void* pos_ids, const void* neg_ids, int32_t ids_dtype,
                               int64_t B, int64_t Nn, int32_t E, float temperature,
                               float false_neg_score, const float* pos_logq,
                               const float* neg_logq, int32_t logq_after_mask, const float* lse, float grad_scale,
                               float* dq, float* ditem, float* dneg_item, void* workspace,
                               int64_t workspace_bytes, mh_stream_t stream);

/* ---- a14: brute-force top-k retrieval ----------------------------------------------------
 * Replaces BruteForce.call (outputs/topk.py:182-237): scores = q C^T (:113-115),
 * tf.math.top_k(scores, k) (values descending, ties -> lower candidate index first),
 * tf.gather(ids, idx).  q[Bq, E], cand[N, E] fp32, cand_ids[N] int32 (NULL -> index itself).
 * Each score is a k-ascending fp32 fmaf chain (bit-reproducible by oracle/oracle_c.c).
 * out_scores[Bq, k] fp32 and out_idx[Bq, k] int32 (candidate row indices) are required (they hold the
 * running lists between candidate chunks); out_ids[Bq, k] int32 may be NULL.
 * k <= 1024, k <= N.  workspace: mh_topk_workspace_bytes(Bq, N, k).
 * Large catalogues (N > 4 * bootstrap) take the fused-filter path: after a dense bootstrap the remaining
 * candidates are scored by an MFMA kernel whose epilogue keeps only scores >= the row's current k-th
 * best; the call then synchronises the stream ONCE to read an overflow flag (adversarially ordered data
 * falls back to the dense path), so it cannot be captured into a hipGraph. */
int64_t mh_topk_workspace_bytes(int64_t Bq, int64_t N, int32_t k);
int32_t mh_topk_dot(const float* q, const float* cand, const int32_t* cand_ids, int64_t Bq,
                    int64_t N, int32_t E, int32_t k, float* out_scores, int32_t* out_ids,
                    int32_t* out_idx, void* workspace, int64_t workspace_bytes,
                    mh_stream_t stream);

/* ---- a14 on the bf16 matrix pipe, same contract: BruteForce.index (outputs/topk.py:124-180) + BruteForce.call (:182-237) ----
 * mh_topk_split is what `BruteForce.index(candidates)` runs ONCE: every fp32 value x is split into two bf16 values,
 * hi = bf16(x) and lo = bf16(x - hi) (hi[n, E], lo[n, E] uint16 bit patterns); norm2[n] (may be NULL) receives |x_row|^2 and
 * *norm2_max (device float, may be NULL, zeroed by the caller) is raised to the largest of them.  E in {32, 64, 128, 256}.
 * mh_topk_dot_split returns EXACTLY what mh_topk_dot returns -- scores are the k-ascending fp32 fmaf chains, indices follow
 * tf.math.top_k's tie rule, bit for bit -- but runs the threshold filter over the catalogue as a 3-term split product
 * (hi hi + hi lo + lo hi) on v_mfma_f32_32x32x16_bf16, keeps the best k' = k + slack approximate scores per row, proves from the
 * error bound 2^-13 |q| max|c| that they contain the exact top-k, re-scores those k' in exact fp32 and orders them; rows for
 * which the proof fails (more than `slack` candidates within the error band of the k-th score) are recomputed exactly on the
 * device.  The filter forms hi hi for every 32 x 32 block and the two small terms only where a score can still reach its threshold
 * (bound 1.05 * 2^-8 |q| max|c|): same survivors.  Catalogues too small for the filter path and widths other than 128 / 64 run
 * mh_topk_dot itself.  No host synchronisation: capturable.
 * workspace: mh_topk_split_workspace_bytes(Bq, N, k, E) (>= mh_topk_workspace_bytes). */
int32_t mh_topk_split(const float* x, int64_t n, int32_t E, uint16_t* hi, uint16_t* lo, float* norm2,
                      float* norm2_max, mh_stream_t stream);
int64_t mh_topk_split_workspace_bytes(int64_t Bq, int64_t N, int32_t k, int32_t E);
int32_t mh_topk_dot_split(const float* q, const float* cand, const uint16_t* cand_hi,
                          const uint16_t* cand_lo, const float* cand_norm2_max, const int32_t* cand_ids,
                          int64_t Bq, int64_t N, int32_t E, int32_t k, float* out_scores,
                          int32_t* out_ids, int32_t* out_idx, void* workspace,
                          int64_t workspace_bytes, mh_stream_t stream);

/* ---- top-k ranking metrics (metrics/topk.py:48-195) on pre-sorted labels ----------------------------
 * labels_sorted[b, j] = relevance of the j-th best candidate of query b (what BruteForce.call returns as
 * targets in testing mode, outputs/topk.py:224-236); relevant_counts[b] = total relevant items (NULL -> 1).
 * out[b, 0..5] = recall, precision, average precision, dcg, ndcg, mrr @k. */
int32_t mh_topk_metrics(const float* labels_sorted, int64_t ld, const float* relevant_counts, int64_t B,
                        int32_t k, float* out, mh_stream_t stream);

/* ---- optional layers of MLPBlock (blocks/mlp.py:108-137): tf.keras.layers.Dropout and BatchNormalization behind a Dense layer ----
 * mh_dropout: y[i] = keep(i) ? x[i] / (1 - rate) : 0 (tf.nn.dropout).  keep(i) of call c is a pure function of (seed, c, i):
 *   word (i % 4) of Philox4x32-10(counter = (i / 4, c), key = seed) >= rate * 2^32.  rng_state: DEVICE uint64[3] =
 *   {seed, calls, last}.  backward == 0: call = calls; afterwards last = calls, calls += 1 (on the stream).  backward != 0:
 *   the same mask as the last forward (call = last) applied to the gradient -- no mask is stored.  x == y is allowed.
 * mh_batchnorm_fwd (axis -1, Keras non-fused 2-D semantics): training != 0: batch mean / biased variance per column ->
 *   save_mean, save_invstd = 1 / sqrt(var + eps); y = (x - mean) invstd gamma + beta; moving = moving * momentum +
 *   batch * (1 - momentum) for mean and (biased) variance.  training == 0: the moving statistics normalise
 *   (save_invstd receives 1 / sqrt(moving_var + eps)).  gamma / beta may be NULL (scale / center off).
 * mh_batchnorm_bwd: dbeta = sum dy, dgamma = sum dy xhat; training: dx = gamma invstd (dy - dbeta / M - xhat dgamma / M);
 *   inference statistics: dx = dy gamma invstd.  Column sums are two-stage with a fixed order (deterministic). */
int32_t mh_dropout(const float* x, float* y, int64_t n, float rate, uint64_t* rng_state, int32_t backward, mh_stream_t stream);
int64_t mh_batchnorm_workspace_bytes(int64_t M, int32_t N);
int32_t mh_batchnorm_fwd(const float* x, int64_t ldx, int64_t M, int32_t N, const float* gamma, const float* beta, float eps,
                         float momentum, int32_t training, float* moving_mean, float* moving_var, float* save_mean,
                         float* save_invstd, float* y, int64_t ldy, void* workspace, int64_t workspace_bytes, mh_stream_t stream);
int32_t mh_batchnorm_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t M, int32_t N, const float* gamma,
                         const float* save_mean, const float* save_invstd, int32_t training, float* dx, int64_t lddx,
                         float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes, mh_stream_t stream);

/* ---- input staging of a replayed step: `count` small buffers (the columns of a batch: HOST arrays of device pointers and
 * byte counts, whole 4-byte words, 4-byte aligned) copied by ONE launch into the static inputs a captured graph reads. */
int32_t mh_copy_many(const void* const* src, void* const* dst, const int64_t* bytes, int32_t count, mh_stream_t stream);

/* ---- ConcatFeatures of narrow fp32 columns (merlin/models/tf/core/aggregation.py:38-66 over the Continuous block's columns,
 * tf/inputs/continuous.py:73-138): out[b, off_i + c] = src_i[b, c], off_i = width_0 + ... + width_{i-1}; `src`, `ld`, `width` are
 * HOST arrays of `count` <= MH_MAX_FEATURES entries (device pointers of [B, width_i] row-major sources with row stride ld_i).
 * Columns [sum width, pad_to) are written as zeros (pad_to = 0: none), so that a consumer may read 16-byte aligned rows;
 * at most 150 output columns, ldo >= max(sum width, pad_to). */
int32_t mh_concat_columns(const float* const* src, const int64_t* ld, const int32_t* width, int32_t count, int64_t B, float* out,
                          int64_t ldo, int32_t pad_to, mh_stream_t stream);

/* ---- measurement probe ("record a stream-copy peak on the box") -----------------------------------
 * dst[0 .. bytes) = src[0 .. bytes) with a float4 grid-stride kernel (16-byte aligned, bytes % 16 == 0): the streaming
 * rate a hand-written kernel reaches on this GPU, reported by bench.py beside the 8 TB/s spec peak. */
int32_t mh_stream_copy(const void* src, void* dst, int64_t bytes, mh_stream_t stream);

/* ---- log-uniform (Zipfian) candidate sampler (outputs/sampling/popularity.py:118-137 ->
 * tf.random.log_uniform_candidate_sampler(range_max, num_sampled, unique)) -------------------------------------
 * out_ids[i] = min_id + k_i with P(k) = (log(k + 2) - log(k + 1)) / log(range_max + 1), k in [0, range_max).
 * Draw j of call c is a pure function of (seed, c, j) (Philox4x32-10; u = 53 bits; k = floor(exp(u log(range_max + 1))) - 1):
 * unique = 0: out_ids[i] = draw i; unique != 0: the first n DISTINCT draws in draw order (what the rejection loop of the
 * TF sampler returns; needs n <= range_max).  rng_state: DEVICE uint64[2] = {seed, calls}; the kernel bumps `calls`, so a
 * replayed hipGraph samples anew each time.  Asynchronous, no host synchronisation. */
int64_t mh_log_uniform_sample_workspace_bytes(int64_t n, int32_t unique);
int32_t mh_log_uniform_sample(int64_t range_max, int64_t min_id, int64_t n, int32_t unique, uint64_t* rng_state,
                              int64_t* out_ids, void* workspace, int64_t workspace_bytes, mh_stream_t stream);

/* ---- BinaryOutput head loss (outputs/classification.py:72-123): BCE on probabilities ----
 * p[M] = sigmoid output of the head, label[M] in {0,1}: loss[m] = keras binary_crossentropy
 * (clip p to [1e-7, 1-1e-7]); dlogit[m] = (p - label) * grad_scale (gradient w.r.t. the
 * pre-sigmoid logit).  Either output may be NULL. */
int32_t mh_bce_fwd_bwd(const float* p, const float* label, int64_t M, float grad_scale,
                       float* loss, float* dlogit, mh_stream_t stream);
/* The same with the batch mean (the scalar `compute_loss` returns, models/base.py:1137-1151) formed on the device
 * in a fixed order; workspace: 256 floats.  loss_mean[1]. */
int32_t mh_bce_mean_fwd_bwd(const float* p, const float* label, int64_t M, float grad_scale, float* loss_mean,
                            float* dlogit, float* workspace, mh_stream_t stream);
/* The two launches of mh_bce_mean_fwd_bwd as separate calls: _partial writes dlogit (what the backward waits for) and one partial
 * sum per workgroup; _finish adds the partials into loss_mean in the same fixed order (same bits) and can be issued any time later
 * on the same stream -- the scalar is only reported (`train_step` returns it, models/base.py:1174), so a step issues it at its tail,
 * where the launch stream idles behind the side stream.  The workspace must not be reused in between. */
int32_t mh_bce_mean_partial(const float* p, const float* label, int64_t M, float grad_scale, float* dlogit, float* workspace,
                            mh_stream_t stream);
int32_t mh_bce_mean_finish(const float* workspace, int64_t M, float* loss_mean, mh_stream_t stream);

/* ---- dense optimizer step for MLP / cross / head weights (models/base.py:1161) -----------
 * SGD: w -= lr*g.  ADAGRAD (keras): state += g^2; w -= lr * g / (sqrt(state) + eps).
 * ADAM (keras): state = b1 state + (1-b1) g; state2 = b2 state2 + (1-b2) g^2; w -= lr_t state/(sqrt(state2)+eps). */
int32_t mh_dense_optimizer_step(float* w, const float* grad, float* state, int64_t n,
                                int32_t optimizer, float lr, float eps, float* state2, float beta1,
                                float beta2, const float* lr_device, mh_stream_t stream);
/* step[0] += 1; lr_t[0] = lr * sqrt(1 - beta2^step) / (1 - beta1^step): the Adam bias correction kept on
 * the device (two fp32 scalars) so that a hipGraph-captured train step advances it on every replay. */
int32_t mh_adam_tick(float* step, float lr, float beta1, float beta2, float* lr_t, mh_stream_t stream);

/* The same update for up to MH_MAX_FEATURES tensors in ONE launch (HOST arrays of device pointers /
 * element counts): the dense parameters of a DLRM are 12 small tensors, one launch instead of 12. */
int32_t mh_dense_optimizer_step_multi(float* const* w, const float* const* grad, float* const* state,
                                      const int64_t* n, int32_t count, int32_t optimizer, float lr, float eps,
                                      float* const* state2, float beta1, float beta2, const float* lr_device,
                                      mh_stream_t stream);

/* Backward of a cross layer  out = x0 * p + x,  p = x W + b  (Cross.call, blocks/cross.py:188-202, differentiated by the
 * GradientTape of BaseModel.train_step, models/base.py:1121-1174):
 *     g = d loss / d p = dout * x0;   d loss / d x0 (this layer's share) = dout * p;   d loss / d x = g W^T + dout;
 *     dW = x^T g;   db = column sums of g.
 * All operands are contiguous [M, d] with d % 4 == 0 (zero-padded layers) and 16-byte aligned.  The non-NULL outputs select
 * the phases, so that the caller can run dX on its launch stream and dW / db beside it on another:
 *   dx0_acc != NULL : ONE element-wise pass: g = dout * x0 (written to the caller's [M, d] buffer `g`) and
 *                     dx0_acc = (accumulate_dx0 ? dx0_acc : 0) + dout * p  (the sum over the layers of a CrossBlock);
 *   dx      != NULL : dx [M, d] = g W^T + dout on the MFMA GEMM, the residual add in its epilogue.  W is [d, r] row-major,
 *                     g [M, r]: r = d for a full-rank layer; for a low-rank layer (DenseMaybeLowRank, blocks/mlp.py:304-396)
 *                     the caller passes g = d loss / d h [M, r] and W = U [d, r];
 *   dW      != NULL : dW [d, d] = x^T g and db [d] (full-rank layer; workspace mh_linear_bwd_workspace_bytes(M, d, d)). */
int32_t mh_cross_layer_bwd(const float* x0, const float* x, const float* p, const float* dout, const float* W, int64_t M,
                           int32_t d, int32_t r, float* g, float* dx0_acc, int32_t accumulate_dx0, float* dx, float* dW,
                           float* db, void* workspace, int64_t workspace_bytes, mh_stream_t stream);

/* Elementwise helper of the cross-layer backward (blocks/cross.py:188-202 under GradientTape):
 * op 0: out = a*b;  op 1: out = a+b;  op 2: out = a*b + c.  n contiguous floats (float4 path when n % 4 == 0 and the
 * pointers are 16-byte aligned).  The full-rank cross backward no longer uses it (mh_cross_layer_bwd). */
int32_t mh_eltwise(int32_t op, const float* a, const float* b, const float* c, float* out, int64_t n,
                   mh_stream_t stream);

/* ---- multi-GPU: row-sharded embedding exchange --------------------------------------
 * The reference scales the embedding lookup by handing the table to SparseOperationKit behind
 * `merlin/models/tf/distributed/embedding.py:16-150` (SOKEmbedding: the table is spread over the Horovod
 * workers, `sok.lookup_sparse` exchanges ids and rows; SOK itself is a closed third-party dependency).  Here the shard rule is owner = row % W, local row = row / W, and the send order is built
 * on the device by a stable counting sort over the owners of the F id columns:
 *   entry e = f*B + b;  send_keys[p] = f << 40 | ids[f][b] / W, grouped by owner, input order within an owner;
 *   pos_of[e] = p (inverse permutation);  src_row[p] = b*F_total + slots[f] (row of a [B, F_total, D] gradient
 *   stack that request p back-propagates into);  counts[w] = requests for owner w (int64, device).
 * ids: HOST array of F device pointers; slots: HOST array. */
int64_t mh_route_workspace_bytes(int64_t n, int32_t W);
/* capacity == 0: dense send order (owner w's requests start at the sum of the counts before it; the exchange then needs
 * the host-side counts).  capacity > 0: FIXED windows -- owner w's requests occupy send_keys[w*capacity ..), padded with
 * key -1 / source row -1, so the all-to-all has equal, host-known splits: no host sync, the step replays from a hipGraph.
 * A request beyond its owner's window is dropped (pos_of = -1) and overflow[0] |= 1 (device flag, may be NULL). */
int32_t mh_route_build(const void* const* ids, int32_t ids_dtype, int32_t F, int64_t B, int32_t W,
                       const int32_t* slots, int32_t F_total, int64_t capacity, int64_t* send_keys, int64_t* pos_of,
                       int64_t* src_row, int64_t* counts, int32_t* overflow, void* workspace, int64_t workspace_bytes,
                       mh_stream_t stream);
/* The same route with per-(sender, owner) DE-DUPLICATION: a row many samples of the batch ask for is requested -- and its
 * gradient returned -- once (what SparseOperationKit does inside `sok.lookup_sparse`, reached from
 * merlin/models/tf/distributed/embedding.py:144-148).  send_keys holds each distinct (feature, id) once, grouped by owner, in
 * the order of FIRST occurrence in entry order (a pure function of the ids); counts[w] = distinct keys for owner w;
 * pos_of[e] = the send slot of entry e's key (entries with equal keys share it; -1: negative id, or the key fell outside a
 * fixed window, overflow[0] |= 1).  The rows come back once per slot: the forward is the gather out[e] = back[pos_of[e]], the
 * backward the segment sum send[p] = sum over {e: pos_of[e] == p} of grad[e] (mh_embedding_gather_bwd, SGD with lr = -1 onto a
 * zero [n_send, D] buffer, does exactly that).  capacity as in mh_route_build; there is no src_row. */
int64_t mh_route_dedup_workspace_bytes(int64_t n, int32_t W);
int32_t mh_route_build_dedup(const void* const* ids, int32_t ids_dtype, int32_t F, int64_t B, int32_t W, int64_t capacity,
                             int64_t* send_keys, int64_t* pos_of, int64_t* counts, int32_t* overflow, void* workspace,
                             int64_t workspace_bytes, mh_stream_t stream);
/* Owner side: rows[i] = base[key >> 40] + (key & (2^40 - 1)): row of the rank's concatenated local shards; -1 (read as a
 * zero row by the gather, skipped by the fused update) for padding keys (< 0) and for local rows >= shard_rows[f]
 * (an id beyond the table's cardinality; shard_rows may be NULL = unchecked). */
int32_t mh_route_local_rows(const int64_t* recv_keys, int64_t n, const int64_t* base, const int64_t* shard_rows,
                            int32_t F, int64_t* rows, mh_stream_t stream);

/* ---- collectives behind the C ABI: RCCL over xGMI, one communicator per process ---------------
 * Reference role: SparseOperationKit's distributed lookup behind `tf/distributed/embedding.py:117-149` and Horovod's
 * gradient all-reduce (`tf/models/base.py:476-508`).  Every call is asynchronous on `stream` and has host-known sizes
 * only: a step built from them is a fixed launch sequence (hipGraph-capturable).  RCCL is resolved with dlopen at the
 * first call (the copy PyTorch already loaded wins: one RCCL per process).
 *   mh_comm_unique_id: rank 0 obtains the 128-byte id; the caller broadcasts it over its own channel (MPI, a TCP store,
 *                      torch.distributed's store) and every rank calls mh_comm_init.  world == 1 needs no id: without one
 *                      the collectives of a one-rank communicator are device copies / no-ops; WITH one it owns a real
 *                      one-rank RCCL communicator and every call goes through RCCL (the single-GPU execution of the N-rank code). */
typedef void* mh_comm_t;
int32_t mh_comm_unique_id(void* id128);
int32_t mh_comm_init(int32_t rank, int32_t world, const void* unique_id128, mh_comm_t* comm_out);
int32_t mh_comm_destroy(mh_comm_t comm);
/* equal windows: bytes_per_peer bytes to / from every rank (send / recv hold world * bytes_per_peer bytes) */
int32_t mh_comm_alltoall(mh_comm_t comm, const void* send, void* recv, int64_t bytes_per_peer, mh_stream_t stream);
/* in-place SUM of n floats across the ranks: reduce-scatter + all-gather when n % world == 0 (every xGMI link carries
 * 1/world of the bucket per phase), plain all-reduce otherwise */
int32_t mh_allreduce_dense(mh_comm_t comm, float* buf, int64_t n, mh_stream_t stream);
/* Row-sharded lookup of F one-hot features in ONE call: route (fixed windows of `capacity` requests per owner, see
 * mh_route_build) -> all-to-all(keys) -> local rows -> gather of the rank's concatenated shards -> all-to-all(rows) ->
 * feature f of sample b lands at out[b * out_row_stride + out_offset[f] ..+D] (HOST offsets, like
 * mh_embedding_gather_fwd).  base / shard_rows: device [F] (first row / row count of feature f's shard in
 * local_shards).  The workspace (mh_sharded_lookup_workspace_bytes) keeps the route for the matching _bwd call.
 * _bwd: grad_stack[B, F, D] = d loss / d (looked-up rows); the rows travel back to their owners, which apply the fused
 * dedup + optimizer update (mh_embedding_gather_bwd semantics) to local_shards (+ state, state2 of the same shape). */
int64_t mh_sharded_lookup_workspace_bytes(int64_t B, int32_t F, int32_t W, int64_t capacity, int32_t D);
int32_t mh_sharded_lookup_fwd(mh_comm_t comm, const void* const* ids /*HOST [F]*/, int32_t ids_dtype, int32_t F, int64_t B,
                              int64_t capacity, const float* local_shards, const int64_t* base,
                              const int64_t* shard_rows, int32_t D, float* out, int64_t out_row_stride,
                              const int64_t* out_offset /*HOST [F]*/, int32_t* overflow, void* workspace,
                              int64_t workspace_bytes, mh_stream_t stream);
int32_t mh_sharded_lookup_bwd(mh_comm_t comm, int32_t F, int64_t B, int64_t capacity, int32_t D, const float* grad_stack,
                              float* local_shards, float* state, float* state2, int64_t local_rows_total,
                              int32_t optimizer, float lr, float eps, float beta1, float beta2, const float* lr_device,
                              void* workspace, int64_t workspace_bytes, mh_stream_t stream);

/* ---- launch recorder: a train step traced once, replayed without re-entering the host language ----------------------------
 * Replaces the role of Keras' traced `train_function` (BaseModel.fit -> make_train_function, models/base.py:1361-1421): between
 * mh_record_begin and mh_record_end every kernel launch of this library is executed AND kept (kernel, geometry, stream, arguments by
 * value), together with the event hand-offs between streams the host issues through mh_record_event / mh_record_wait_event;
 * mh_record_replay re-issues the sequence in order on the same streams.  Contract of a replayed step = contract of a captured
 * graph: fixed shapes, every buffer the launches address stays alive and in place (the host mirror stages each batch into static
 * inputs and keeps the step's intermediates allocated).  Unlike a hipGraph the replay keeps the step's multi-stream overlap, and
 * unlike per-stream graph segments it pays no graph-launch latency: a few microseconds of host time per launch.
 * One recording at a time per process; the host side must be single-threaded while a recording is open. */
int32_t mh_record_begin(void);
int32_t mh_record_end(void** handle_out);
int32_t mh_record_abort(void);
int32_t mh_record_event(mh_stream_t stream, int64_t* event_id_out);   /* record an event on `stream` (recording only) */
int32_t mh_record_wait_event(mh_stream_t stream, int64_t event_id);   /* `stream` waits for that event */
int32_t mh_record_replay(void* handle);
int32_t mh_record_info(void* handle, int64_t* launches, int64_t* hand_offs);
int32_t mh_record_free(void* handle);

#ifdef __cplusplus
}
#endif
#endif /* MERLIN_HIP_H */
