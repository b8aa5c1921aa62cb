/*
 * recman_hip.h - C ABI of librecman_hip.so: the MI355X (gfx950) kernels behind
 * recman's CTR forward+backward path (DeepFM / DCN / xDeepFM).
 *
 * The reference (dev-wei/recman) has no native code, no operator registry and no
 * FFI: its seam is the Python layer-callable protocol `Layer(variables, ...)(x)`
 * of recman/tf/core/layers.py, composed by xDeepFM._out (recman/tf/core/xDeepFM.py:47-104),
 * DeepFM._init_graph (DeepFM.py:107-163) and DCN._init_graph (DCN.py:99-149).
 * Each entry point below cites the span of that Python it replaces.  The Python
 * host side that binds these (ctypes) is recman_amd/_lib.py; INTEGRATION.md shows
 * the stub a maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless named host_*; the caller owns all
 *    buffers, the library allocates nothing and keeps no global mutable state;
 *  - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *    default stream) and return immediately: asynchronous, graph-capturable;
 *  - return value: RM_OK (0) or a negative RM_E* code; rm_last_error() gives a
 *    thread-local message for the last failing call.  No exceptions cross the ABI;
 *  - floats are IEEE fp32, indices are int64 (recman/tf/inputs.py:158,199);
 *  - layouts are row-major; E is [B, F, D] exactly as tf.concat(axis=1) of the
 *    per-field [B,1,D] lookups produces it (layers.py:248-253).
 */
#ifndef RECMAN_HIP_H
#define RECMAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RM_OK 0
#define RM_EINVAL (-1)  /* bad argument (null pointer, size, alignment)      */
#define RM_ELAUNCH (-2) /* hipLaunch / runtime error, see rm_last_error()    */
#define RM_EUNSUPPORTED (-3)

/* activation ids (layers.py:601,738; tf.nn.leaky_relu alpha = 0.2) */
#define RM_ACT_IDENTITY 0
#define RM_ACT_RELU 1
#define RM_ACT_LEAKY_RELU 2

typedef void *rm_stream_t;

int rm_version(void);
const char *rm_last_error(void);
/* number of compute units of the current device (grid sizing for callers) */
int rm_device_cus(void);
/* Enqueues an empty one-wave kernel named rm_profile_marker_kernel: brackets a stretch of launches so
 * that a rocprofv3 kernel / counter trace can be cut at it (bench.py's PMC child passes).  No memory
 * is touched.  (New: the reference has no profiler hooks beyond tf.summary.trace_on,
 * recman/tf/core/TensorBoardLogger.py:58-69.) */
int rm_profile_marker(int tag, rm_stream_t stream);

/* ------------------------------------------------------------------------
 * Embedding gather (+ FM, + sparse/dense linear term), forward.
 * Replaces: FeatEmbedding.__call__ SparseFeat branch + FeatEmbeddingLayer.__call__
 *   (layers.py:117-128, 238-261), FMLayer.__call__ (layers.py:457-478) and
 *   LinearCombiner/LinearLayer resp. SparseLinearCombiner/SparseLinearLayer
 *   (layers.py:281-347, 368-439; one_hot utils.py:51-67) in gather form.
 *
 *   idx        [B,F]  per-field row index (0 = null/unknown, inputs.py:166)
 *   table      [R,D]  all field tables concatenated, row r at table + r*table_ld
 *                     (table_ld >= D floats); field f owns rows
 *                     field_off[f] .. field_off[f]+V_f-1
 *   bias_table [R]    first-order FM bias tables, same row numbering, element r at
 *                     bias_table + r*bias_ld, or NULL
 *   lin_w             sparse part of linear_w (one one-hot block per feature), or
 *                     NULL; element (f, i) at lin_w + (lin_off[f]+i)*lin_ld.
 *                     (the *_ld strides let bias and linear weight live inside the
 *                     embedding row - one HBM sector per lookup instead of three)
 *   lin_w_dense [Dn]  linear weights of the dense columns, lin_w0 [1] the bias
 *   dense      [B,Dn] scaled dense features or NULL (Dn = 0)
 *   mask_b [B,F], mask_e [B,F,D]: FMLayer dropout multipliers (0 or 1/keep,
 *                     layers.py:461,466) or NULL = keep-probability 1
 * outputs (each may be NULL = not wanted):
 *   E        [B,F,D]  gathered rows (the un-dropped embeddings the DNN/CIN/cross use)
 *   fm_sum   [B,D]    S = sum_f mask_e*E  (saved for the backward)
 *   fm_logit [B]      sum_f mask_b*bias + 0.5*sum_k(S_k^2 - sum_f (mask_e*E)_fk^2)
 *   lin_logit[B]      sum_f lin_w[lin_off[f]+idx] + dense . lin_w[dense block] + lin_w0
 * D must be a multiple of 4 and <= 256; table/E 16-byte aligned.
 */
/* flags: RM_EMBED_STREAM_ROWS = load the table rows non-temporally (fused-row layout only).  For ids that
 * touch a row about once per batch (uniformly hashed ids over a table far larger than the caches) the
 * row lines then do not evict E from L2 / the Infinity Cache, which the next kernels read; ids with
 * heavy reuse (Zipf) want the plain, cached loads. */
enum { RM_EMBED_STREAM_ROWS = 1 };
int rm_embed_fwd(const int64_t *idx, const float *table, int64_t table_ld,
                 const int64_t *field_off, const float *bias_table, int64_t bias_ld,
                 const float *lin_w, int64_t lin_ld, const int64_t *lin_off,
                 const float *lin_w_dense, const float *lin_w0, const float *dense, int Dn,
                 const float *mask_b, const float *mask_e, int64_t B, int F, int D,
                 float *E, float *fm_sum, float *fm_logit, float *lin_logit,
                 int flags /* RM_EMBED_* */, rm_stream_t stream);

/* The linear term on its own: (Sparse)LinearCombiner + (Sparse)LinearLayer.__call__ (layers.py:281-347,
 * 368-439; one_hot utils.py:51-67) in gather form, for callers that compose the reference's layer
 * callables one by one (recman_amd/th/layers.py; the engines take it from rm_embed_fwd):
 *   out[b] = sum_f w[lin_off[f] + idx[b,f]] + sum_j dense[b,j] * w_dense[j] + w0[0]
 * w0 may be NULL.  Its backward is rm_scatter_add_rows (g_row form) + rm_linear_dense_bwd. */
int rm_linear_fwd(const int64_t *idx, const int64_t *lin_off, const float *w, const float *dense,
                  const float *w_dense, const float *w0, int64_t B, int F, int Dn, float *out,
                  rm_stream_t stream);

/* Backward of the embedding + FM block w.r.t. the gathered rows: the IndexedSlices
 * values TF's autodiff produces for tf.nn.embedding_lookup (one row per (b,f)
 * occurrence, duplicates NOT merged; the indices are `idx` itself).
 *   d_rows[b,f,:] = dE_up[b,f,:] + g_fm[b] * mask_e[b,f,:] * (S[b,:] - mask_e*E[b,f,:])
 *   d_bias[b,f]   = g_fm[b] * mask_b[b,f]
 * dE_up (upstream gradient from DNN / CIN / cross w.r.t. E) may be NULL; g_fm may
 * be NULL (no FM term: d_rows = dE_up).  d_rows may alias dE_up.  d_bias may be NULL.
 */
int rm_embed_bwd(const float *E, const float *fm_sum, const float *dE_up, const float *g_fm,
                 const float *mask_b, const float *mask_e, int64_t B, int F, int D,
                 float *d_rows, float *d_bias, rm_stream_t stream);

/* Scatter-add of occurrence rows into a dense table gradient (what TF's
 * IndexedSlices densify to once the l2 term touches every row, layers.py:188-193):
 *   d_table[(field_off[f]+idx[b,f])*ld + k] += rows[b,f,k], k < width   (float atomics)
 * `width` = D for embedding tables, 1 for bias tables / linear weights (then
 * field_off = lin_off).  When g_row != NULL, rows is ignored and the added value
 * is g_row[b] (width must be 1): the D=1 linear / bias gradient. */
int rm_scatter_add_rows(const int64_t *idx, const int64_t *field_off, const float *rows,
                        const float *g_row, int64_t B, int F, int width, int64_t ld,
                        float *d_table, rm_stream_t stream);

/* Weighted column sums: d_w_dense[j] = sum_b g[b]*dense[b,j] (j < Dn <= 1023),
 * d_w0 = sum_b g[b].  The dense part of the linear-layer backward, and the gradient of
 * every [*,1] output projection (d cin_w = pooled^T g, d dnn_w = a^T g).  Deterministic
 * two-stage reduction; workspace >= 262144 floats; either output may be NULL. */
int rm_linear_dense_bwd(const float *g, const float *dense, int64_t B, int Dn,
                        float *d_w_dense, float *d_w0, float *workspace, rm_stream_t stream);

/* ------------------------------------------------------------------------
 * PredictionLayer + create_loss, forward and backward in one pass.
 * Replaces: tf.add_n of the branch logits (xDeepFM.py:99-102, DeepFM.py:149-158,
 *   DCN.py:140-144), PredictionLayer.__call__ (layers.py:796-808), create_loss
 *   (utils.py:192-198: Keras binary_crossentropy on probabilities, clip 1e-7) and
 *   their gradient.
 *   logit_a..d [B]  branch logits (NULL = absent); coef_* multiply them (DCN's
 *                   double-counted dnn logit uses coef 2 under strict_reference)
 *   y   [B] int64 labels (classification) or NULL with y_f [B] float (regression)
 *   task 0 = classification (sigmoid + BCE), 1 = regression (MSE)
 * outputs: logit [B] (sum), pred [B], dlogit [B] = d(mean loss)/d(logit) (NULL ok),
 *   loss [1] (NULL ok; needs workspace >= 1024 floats).
 */
int rm_logit_loss(const float *logit_a, float coef_a, const float *logit_b, float coef_b,
                  const float *logit_c, float coef_c, const float *logit_d, float coef_d,
                  const int64_t *y, const float *y_f, int task, int64_t B, float *logit,
                  float *pred, float *dlogit, float *loss, float *workspace,
                  rm_stream_t stream);

/* ------------------------------------------------------------------------
 * Skinny DNN (every hidden width <= 32, 1..3 hidden layers, FD+Dn <= 448), fused on the
 * f32 MFMA.  Replaces DNNCombiner + DNN.__call__ (layers.py:494-501, 576-609) and their
 * gradient for the reference-default deep_hidden_units (32, 32) (DeepFM.py:37,
 * hparams/xDeepFM.py:27); wider MLPs (DCN's [400,400]) go to rocBLAS/hipBLASLt.
 *   x = [xe | xd] (never concatenated); W[0] [FD+Dn, H0], W[l] [H_{l-1}, H_l], bias[l] [H_l];
 *   w_out [H_last], w0_out [1];  host arrays H / W / bias / h_out / dh / dW have NL entries.
 * rm_mlp_fwd: h_out[l] [B,32] (post-activation, columns >= H_l zero), logit [B].
 * rm_mlp_bwd: g [B] = dLoss/dlogit; writes dLoss/dxe into d_rows [B,FD]; when fm_sum
 *   (S [B,D] from rm_embed_fwd) is given the FM second-order gradient g*(S - E) is added
 *   (xe is E: the whole DeepFM row gradient in one pass, no rm_embed_bwd launch);
 *   dh[l] [B,32] = dLoss/d(pre-activation of layer l); dW[l] / db[l] = weight / bias
 *   gradients, d_w_out [H_last], d_w0_out [1] = gradients of the output projection
 *   (db, d_w_out, d_w0_out may be NULL); d_xd_wsum [Dn] (NULL or Dn <= 32) = sum_b g[b] * xd[b,:],
 *   and d_g_sum [1] (NULL ok) = sum_b g[b]: the gradients of the linear term's dense weights and
 *   bias when the same g drives it (layers.py:330-347; saves the rm_linear_dense_bwd pass).
 *   Deterministic (no float atomics).
 *   workspace: rm_mlp_bwd_workspace(FD, Dn) floats.
 *   flags: RM_MLP_STREAM_DROWS = store d_rows non-temporally - only when NOTHING re-reads it soon (a
 *   bare fwd+bwd); in training the optimizer step gathers it right after: leave it 0.
 *
 * Fused training head (rm_mlp_tail, optional - NULL gives the plain calls above).  When the DNN is
 *   the LAST branch of the model's forward (DeepFM, xDeepFM), everything between its output and its
 *   backward is elementwise per example: final logit = coef_mlp * dnn + coef_a * logit_a + coef_b *
 *   logit_b (xDeepFM.py:99-102), PredictionLayer + loss (layers.py:796-808, utils.py:192-198, the
 *   arithmetic of rm_logit_loss) and the dh chain of the hidden layers.  rm_mlp_fwd with a tail does
 *   all of it in the forward kernel's epilogue, where h_l are still in registers: it writes logit /
 *   pred / dlogit (= dLoss/dlogit * grad_scale, the `g` of the backward), the per-tile loss sums
 *   loss_partial [ceil(B/32)] and dh[l] [B,32].  rm_mlp_bwd with the SAME tail skips its own dh-chain
 *   launch (g = tail->dlogit) and its finishing kernel also reduces loss_partial into loss [1]
 *   (mean over B; no l2 terms).  Saves three launches per step (rm_logit_loss's two kernels and the
 *   chain kernel) - 20 of 237 us on the DeepFM benchmark step. */
typedef struct rm_mlp_tail {
  const float *logit_a; /* other branch logits [B], NULL = absent */
  float coef_a;
  const float *logit_b;
  float coef_b;
  float coef_mlp;       /* coefficient of this MLP's logit in the sum */
  const int64_t *y;     /* labels: int64 (classification) ... */
  const float *y_f;     /* ... or float (regression); exactly one of them */
  int task;             /* 0 = classification (sigmoid + binary cross-entropy), 1 = regression (MSE) */
  float grad_scale;     /* dlogit is multiplied by this (micro-batches / ranks); 1 = plain mean over B */
  float *logit, *pred, *dlogit; /* [B] outputs (logit / pred may be NULL) */
  float *loss_partial;  /* [ceil(B/32)] workspace: per-tile loss sums */
  float *loss;          /* [1], written by rm_mlp_bwd's finishing kernel; NULL = not wanted */
  float *dh[3];         /* dh[l] [B,32] for l < NL (the same buffers rm_mlp_bwd gets as dh) */
} rm_mlp_tail;

#define RM_MLP_STREAM_DROWS 1
int rm_mlp_supported(int FD, int Dn, int NL, const int *H);
int rm_mlp_fwd(const float *xe, const float *xd, int FD, int Dn, int NL, const int *H,
               const float *const *W, const float *const *bias, const float *w_out,
               const float *w0_out, int act, int64_t B, float *const *h_out, float *logit,
               const rm_mlp_tail *tail, rm_stream_t stream);
/* Gather + FM + linear term + skinny MLP (+ training head) in ONE kernel: rm_embed_fwd followed by rm_mlp_fwd
 * on E without the round trip of x through HBM (FeatEmbeddingLayer + FMLayer + LinearLayer + DNNCombiner + DNN of
 * DeepFM._init_graph, recman/tf/core/DeepFM.py:96-150 over layers.py:238-261,457-478,330-347,494-501,576-609).
 * Fused-row layout only: table rows [16 embedding | bias | linear weight | ...] with table_ld = 32 (or 20: the
 * rows a row-sharded table exchanges, idx = positions in the received buffer, field_off = 0), D = 16;
 * Dn <= 16, 16 F + Dn <= 448, hidden widths <= 32 (rm_embed_mlp_fwd_supported).  want_bias / want_lin: the FM
 * bias sum / the sparse part of the linear term are wanted; lin_w_dense [Dn] (NULL: the linear term has no
 * dense part), lin_w0 [1] or NULL; xd [B,Dn] feeds the MLP (and the linear term).  Outputs as the two calls':
 * E [B,F,16] (always written: the backward reads it), fm_sum [B,16], fm_logit [B], lin_logit [B] (NULL = not
 * wanted), h_out / logit / tail as rm_mlp_fwd - tail->logit_a / logit_b must be this call's lin_logit /
 * fm_logit buffers (or NULL).  flags: RM_EMBED_STREAM_ROWS.  No dropout masks (the callers fall back to the
 * two separate entry points). */
int rm_embed_mlp_fwd_supported(int F, int D, int64_t table_ld, int Dn, int NL, const int *H);
int rm_embed_mlp_fwd(const int64_t *idx, const float *table, int64_t table_ld, const int64_t *field_off,
                     int want_bias, int want_lin, const float *lin_w_dense, const float *lin_w0,
                     const float *xd, int Dn, int64_t B, int F, int D, float *E, float *fm_sum,
                     float *fm_logit, float *lin_logit, int flags, int NL, const int *H,
                     const float *const *W, const float *const *bias, const float *w_out,
                     const float *w0_out, int act, float *const *h_out, float *logit,
                     const rm_mlp_tail *tail, rm_stream_t stream);
int64_t rm_mlp_bwd_workspace(int FD, int Dn);
int rm_mlp_bwd(const float *xe, const float *xd, int FD, int Dn, int NL, const int *H,
               const float *const *W, const float *w_out, int act, int64_t B, const float *g,
               const float *const *h, const float *fm_sum, int D, float *d_rows,
               float *const *dh, float *const *dW, float *const *db, float *d_w_out,
               float *d_w0_out, float *d_xd_wsum, float *d_g_sum, float *workspace,
               const rm_mlp_tail *tail, int flags, rm_stream_t stream);

/* DeepFM's WHOLE training step in one kernel + one finishing launch: DeepFM._init_graph forward and the gradient
 * of every variable (recman/tf/core/DeepFM.py:107-180; FeatEmbeddingLayer layers.py:238-261, LinearLayer :330-347,
 * FMLayer :457-478, DNN :576-609, PredictionLayer :796-808, create_loss utils.py:192-198) - what rm_embed_mlp_fwd
 * followed by rm_mlp_bwd compute, without E, S, h_l or dh_l ever reaching HBM (every DeepFM gradient except the
 * parameter reductions is local to the example).  Shapes: fused-row table [R, table_ld] (rows [16 embedding | bias
 * entry | linear weight | ...], table_ld a multiple of 4, >= 20), D = 16, F <= 26, Dn <= 16, NL = 2 hidden layers
 * of width <= 32 (rm_deepfm_step_supported); FM, bias tables, linear term and DNN all on, no dropout masks - the
 * callers use the separate entry points for anything else.
 * In: idx [B,F] int64, field_off [F], dense [B,Dn] (NULL when Dn = 0), labels y (int64) or y_f (float),
 *   W[l] / bias[l] (l < 2), w_out [H1], w0_out [1], lin_w_dense [Dn], lin_w0 [1], act, task (0 classification,
 *   1 regression), grad_scale (dlogit multiplier: micro-batches / ranks; 1 = plain mean over B).
 * Out: d_rows [B,F,16] = dLoss/dE (IndexedSlices form, duplicates not merged), logit / pred / dlogit [B] (dlogit =
 *   dLoss/dlogit: the per-occurrence gradient of the bias-table and sparse linear entries), loss [1] (mean over B,
 *   no l2 terms), dW[l] / db[l] (db may be NULL), d_w_out [H1], d_w0_out [1], d_lin_w_dense [Dn], d_lin_w0 [1].
 * packed_rows > 0 (row-sharded table, recman_amd/dist.py): `table` is the buffer of RECEIVED rows [packed_rows,
 *   20] ([16 embedding | bias | linear | 0 0], table_ld = 20), idx [B,F] = the position of every occurrence in it
 *   (field_off zeros), and d_rows is the send buffer of the backward exchange, [packed_rows, 20]: occurrence (b, f)
 *   writes [dE | dlogit | dlogit * lin_field_mask[f] | 0 0] to row idx[b, f] - what rm_pack_grad_rows builds from
 *   the plain layout (lin_field_mask [F] or NULL: the linear_features subset).  packed_rows = 0: the plain layout.
 * workspace: rm_deepfm_step_workspace(F, Dn) floats.  flags: bit 0 = non-temporal row loads
 *   (RM_EMBED_STREAM_ROWS), bit 1 = non-temporal d_rows stores (only when nothing re-reads them soon), bit 2 = skip
 *   the finishing launch (measurement only: the parameter gradients and the loss are then NOT written).
 * Deterministic (fixed summation orders, no float atomics). */
int rm_deepfm_step_supported(int F, int D, int64_t table_ld, int Dn, int NL, const int *H);
int64_t rm_deepfm_step_workspace(int F, int Dn);
int rm_deepfm_step(const int64_t *idx, const float *table, int64_t table_ld, const int64_t *field_off,
                   const float *dense, int Dn, const int64_t *y, const float *y_f, int64_t B, int F, int D,
                   int NL, const int *H, const float *const *W, const float *const *bias, const float *w_out,
                   const float *w0_out, const float *lin_w_dense, const float *lin_w0, int act, int task,
                   float grad_scale, float *d_rows, float *logit, float *pred, float *dlogit, float *loss,
                   float *const *dW, float *const *db, float *d_w_out, float *d_w0_out, float *d_lin_w_dense,
                   float *d_lin_w0, float *workspace, int64_t packed_rows, const float *lin_field_mask, int flags,
                   rm_stream_t stream);

/* Epilogues of the library-GEMM DNN path (wide hidden layers, layers.py:593-601):
 * rm_bias_act: x[b,j] = act(x[b,j] + bias[j]) in place (bias may be NULL);
 * rm_act_bwd:  da[b,j] *= act'(a[b,j]) in place, act' read off the post-activation a. */
int rm_bias_act(float *x, const float *bias, int64_t B, int N, int act, rm_stream_t stream);
int rm_act_bwd(float *da, const float *a, int64_t B, int N, int act, rm_stream_t stream);
/* da[b,j] = g[b] * w[j] * act'(a[b,j]) (a = post-activation of the last hidden layer, or NULL for no
 * activation factor): the backward of the [H,1] output projection (layers.py:606-609) and of the last
 * activation in one pass.  N % 4 == 0. */
int rm_outer_actgrad(const float *g, const float *w, const float *a, int64_t B, int N, int act,
                     float *da, rm_stream_t stream);
/* The same pass also reduces the columns of the two arrays it touches: d_w[j] = sum_b g[b] a[b,j]
 * (gradient of dnn_w), d_w0 = sum_b g[b] (dnn_w0), db[j] = sum_b da[b,j] (bias of the last hidden
 * layer) - three reads of a [B,N] array less.  a is required; d_w / d_w0 / db may be NULL;
 * workspace: rm_outer_actgrad_sums_workspace(B, N) floats.  Deterministic. */
int64_t rm_outer_actgrad_sums_workspace(int64_t B, int N);
int rm_outer_actgrad_sums(const float *g, const float *w, const float *a, int64_t B, int N, int act,
                          float *da, float *d_w, float *d_w0, float *db, float *workspace,
                          rm_stream_t stream);

/* out[b] = sum_j X[b,j]*w[j] + w0[0]: the [*,1] output projections (dnn_w/dnn_w0,
 * layers.py:606-609; cin_w/cin_w0, layers.py:757-760).  w0 may be NULL. */
int rm_rowdot(const float *X, const float *w, const float *w0, int64_t B, int P, float *out,
              rm_stream_t stream);

/* ------------------------------------------------------------------------
 * CrossNet (DCN v1, vector form), all L layers fused.
 * Replaces: CrossNet(layer_num, l2_reg)(dnn_input) -> logit [B,1] used at
 *   DCN.py:134-137 - the class itself is ABSENT from the reference
 *   (DCN.py:7 comments the import out); arithmetic per arXiv 1708.05123 eq. (3):
 *   x_{l+1} = x0 * (x_l . w_l) + b_l + x_l,  logit = x_L . w_out
 * x0 = [xe | xd]: xe [B,FD] is the flattened embedding block E, xd [B,Dn] the dense
 * columns (DNNCombiner, layers.py:494-501) - never concatenated in memory.
 *   w, b [L,d], w_out [d], d = FD + Dn, FD % 4 == 0, 0 < FD <= 512, L <= 8.
 * Computed in closed form (exact algebra): x_l = c_l x0 + Bp_l with Bp_l = sum_{j<l} b_j, so
 *   s_l = x_l.w_l = c_l p_l + beta_l,  p_l = x0.w_l,  beta_l = Bp_l.w_l,  c_{l+1} = c_l + s_l,
 *   logit = c_L p_out + beta_out  (p_out = x0.w_out): L+1 dot products per row + a scalar recurrence.
 *   p_out [B, p_ld] (p_ld >= L+1; NULL ok): the row's dot products (p_0..p_{L-1}, x0.w_out), saved
 *   for the backward, which needs nothing else of x0.
 */
int rm_cross_fwd(const float *xe, const float *xd, int FD, int Dn, const float *w,
                 const float *b, const float *w_out, int L, int64_t B, float *logit,
                 float *p_out, int p_ld, rm_stream_t stream);

/* Backward: given g [B] = dLoss/dlogit and p [B, p_ld] from the forward,
 *   d_xe [B,FD] (+ dx_in_e when given): gradient w.r.t. the embedding block of x0
 *       = g c_L w_out + sum_l t_l c_l w_l, optionally summed with another branch's gradient
 *       (t_l = g p_out + sum_{j>l} t_j p_j).  x0 itself is not read; the dense columns are input
 *       data and get no gradient (the reference differentiates variables only, xDeepFM.py:126).
 *   coef [B, 2L+2]: per-example scalars from which the parameter gradients follow
 *       by one skinny GEMM (see recman_amd/engine.py and DESIGN.md):
 *       columns 0..L-1 = t_l*c_l, L = g*c_L, L+1..2L = t_l, 2L+1 = g
 */
int rm_cross_bwd(int FD, int Dn, const float *w, const float *b, const float *w_out, int L,
                 int64_t B, const float *g, const float *p, int p_ld, const float *dx_in_e,
                 float *d_xe, float *coef, rm_stream_t stream);

/* Finishes the CrossNet parameter gradients from P = x0^T @ coef[:, :L+1] ([d, L+1],
 * row-major) and colsum = sum_b coef[:, L+1:] ([L+1]):
 *   d_w[l] = P[:,l] + T_l * Bp_l,  d_w_out = P[:,L] + G * Bp_L,
 *   d_b[l] = G * w_out + sum_{j>l} T_j * w_j,   Bp_l = sum_{j<l} b_j */
int rm_cross_param_grads(const float *P, const float *colsum, const float *w, const float *b,
                         const float *w_out, int L, int d, float *d_w, float *d_b,
                         float *d_w_out, rm_stream_t stream);

/* ------------------------------------------------------------------------
 * CIN (xDeepFM), one layer, on the f32-input MFMA.
 * Replaces the loop body of CIN.__call__ (layers.py:714-752):
 *   Z[b,d,i*H+j] = X0[b,i,d] * Xk[b,j,d];  M = Z @ W + bias;  out = act(M)
 *   out laid out [B,N,D] (after the transpose of layers.py:739); Z is never formed.
 *   X0 [B,m,D]; Xk: rows j < H of a [B,*,D] tensor with xk_bstride floats between
 *   examples (the "next hidden" half of the previous layer's map, layers.py:744-746);
 *   W [m*H, N] (cin_filter_k[0]), bias [N].
 *   pooled [B, pool_stride]: when not NULL, sum_d out[b,n,d] for n in [pool_from, N)
 *   is written to pooled[b, pool_col0 + n - pool_from] (the direct-connect half and
 *   the reduce_sum of layers.py:751-755).
 *   filter_ws: scratch of rm_cin_filter_workspace(m,H,N) floats (the filter
 *   re-laid out as MFMA B-operand; rewritten by every call).
 * N <= 128; D a multiple of 4 dividing 256; (m+1+H)*1 KiB + 32 KiB of LDS <= 160 KiB.
 */
int64_t rm_cin_filter_workspace(int m, int H, int N);
int rm_cin_layer_fwd(const float *X0, const float *Xk, int64_t xk_bstride, const float *W,
                     const float *bias, int act, int64_t B, int m, int H, int N, int D,
                     float *out, float *pooled, int pool_stride, int pool_col0, int pool_from,
                     float *filter_ws, rm_stream_t stream);
/* rm_cin_layer_fwd6: the same layer on the bf16 matrix pipe with split fp32 operands (csrc/cin6.hip; the scheme of
 * rm_dense_fwd6): Z = fl(X0 * Xk) is formed in fp32 as the reference does, then split into three bf16 pieces; six
 * exact piece products per k-step, fp32 accumulate - fp32-level error.  Covers H <= 64 (padded to a multiple of 32),
 * N <= 128, D in {16, 32, 64}; RM_EUNSUPPORTED otherwise (run rm_cin_layer_fwd, which is also the better choice
 * for the first layer: its symmetric k' ordering does half the work).
 * filter_ws: rm_cin_filter_workspace6(m, H, N, D) floats (0 = not covered), 16-byte aligned. */
int64_t rm_cin_filter_workspace6(int m, int H, int N, int D);
int rm_cin_layer_fwd6(const float *X0, const float *Xk, int64_t xk_bstride, const float *W, const float *bias,
                      int act, int64_t B, int m, int H, int N, int D, float *out, float *pooled, int pool_stride,
                      int pool_col0, int pool_from, float *filter_ws, rm_stream_t stream);

/* Backward of one CIN layer (three MFMA passes over the same GEMM shape).
 *   out [B,N,D]: the layer's post-activation map (act' is read off its sign);
 *   the gradient w.r.t. out is never materialised by the caller:
 *     n <  pool_from: d_hidden[b,n,d]  (the next layer's dXk, dh_bstride floats per example)
 *     n >= pool_from: g[b] * cin_w_direct[n - pool_from]   (reduce_sum + cin_w, layers.py:754-758)
 *   dX0 [B,m,D] (+= when accumulate_dx0 & 1) = sum_j (dM @ W^T)[.,(i,j)] * Xk[b,j,d]
 *        accumulate_dx0 & 2: the dX pass runs on the bf16 matrix pipe with split fp32 operands (csrc/cin6.hip, the
 *        scheme of rm_cin_layer_fwd6), and the dW pass too, where csrc/cin6.hip covers the layer (H <= 64, 64 < N <=
 *        128, D in {16, 32, 64}; not the first layer, whose symmetric f32 kernels do half the work - & 4: its dX too)
 *   dXk [B,H,D] (dxk_bstride floats per example) = sum_i (dM @ W^T)[.,(i,j)] * X0[b,i,d];
 *        with xk_is_x0 (first layer: Xk is X0 itself) it is added into dX0 instead
 *   dW [m*H,N] = Z^T @ dM,  dbias [N] = colsum(dM)   (dM = d_out * act'(out))
 *   workspace: rm_cin_bwd_workspace(B,m,H,N,D) floats.  D must divide 64. */
int64_t rm_cin_bwd_workspace(int64_t B, int m, int H, int N, int D);
int rm_cin_layer_bwd(const float *X0, const float *Xk, int64_t xk_bstride, int xk_is_x0,
                     const float *W, int act, const float *out, const float *d_hidden,
                     int64_t dh_bstride, const float *g, const float *cin_w_direct, int pool_from,
                     int64_t B, int m, int H, int N, int D, float *dX0, int accumulate_dx0,
                     float *dXk, int64_t dxk_bstride, float *dW, float *dbias, float *workspace,
                     int64_t workspace_floats, rm_stream_t stream);

/* ------------------------------------------------------------------------
 * Wide dense layers on the f32 MFMA (hidden widths > 32: DCN's deep_hidden_units (400, 400),
 * DCN.py:33; and the matrix form of the cross layer).  Replaces tf.matmul(x, W) + bias +
 * activation of DNN.__call__ (layers.py:594-602) and the GEMMs of its gradient, with the
 * dense-input piece, bias, activation and activation gradient fused (csrc/gemm.hip).
 *
 * rm_dense_fwd:  C[M,N] = epilogue( [A1 | A2][M, K1+K2] . op(W) )
 *   A1 [M,K1] (float4 loads when its rows are 16-byte aligned: lda1 % 4 == 0; per-element loads
 *   otherwise), A2 [M,K2] or NULL (the 13 dense inputs);
 *   W row-major: [K,N] (w_transposed = 0) or [N,K] (w_transposed = 1, i.e. op(W) = W^T);
 *   epilogue (acc = the GEMM result, b = bias[col] or 0 when bias is NULL):
 *     RM_DENSE_BIAS_ACT     C = act(acc + b)
 *     RM_DENSE_MUL_ACTGRAD  C = (acc + b) * act'(.) taken from the POST-activation values aux1
 *     RM_DENSE_ADD          C = acc + b (+ aux1 when not NULL)
 *     RM_DENSE_CROSS        u = acc + b ; C = aux1 * u + aux2 ; C2 = u when not NULL
 *                           (x_{l+1} = x0 o (W x_l + b) + x_l with aux1 = x0, aux2 = x_l)
 *   filter_ws: rm_dense_filter_workspace(K1+K2, N) floats (the weights re-laid for the kernel).
 * rm_dense_wgrad: dW[K1+K2, N] (+)= [A1 | A2]^T . G[M,N]  (reduction over the batch, split
 *   into slabs + a deterministic second-stage sum); db [N] (NULL ok) = the column sums of G, i.e. the
 *   bias gradient of the same layer, from the G tiles the kernel stages anyway;
 *   workspace: rm_dense_wgrad_workspace floats. */
enum { RM_DENSE_BIAS_ACT = 0, RM_DENSE_MUL_ACTGRAD = 1, RM_DENSE_ADD = 2, RM_DENSE_CROSS = 3 };
int64_t rm_dense_filter_workspace(int K, int N);
int rm_dense_fwd(const float *A1, int64_t lda1, int K1, const float *A2, int64_t lda2, int K2,
                 const float *W, int64_t ldw, int w_transposed, int N, const float *bias, int epilogue,
                 int act, const float *aux1, int64_t ld_aux1, const float *aux2, int64_t ld_aux2,
                 int64_t M, float *C, int64_t ldc, float *C2, int64_t ldc2, float *filter_ws,
                 rm_stream_t stream);
/* rm_dense_fwd6: rm_dense_fwd's GEMM with every fp32 operand SPLIT into three bf16 pieces (x = h + m + l, 8 + 8 + 8
 * significant bits) and six of the nine piece products formed on the bf16 matrix pipe, fp32 accumulate
 * (csrc/gemm6.hip): the kept products are exact in fp32, the dropped ones are below 3 * 2^-24 of |x y| - the
 * result carries fp32-level error (measured against float64 it is SMALLER than the f32-MFMA kernel's, whose
 * k-chunked sums are longer chains).  gfx950 has no tf32; its f32 MFMA runs at 1/16 of the bf16 rate.
 * Arguments as rm_dense_fwd; epilogues RM_DENSE_BIAS_ACT / MUL_ACTGRAD / ADD; A1 rows 16-byte aligned (lda1 % 4 ==
 * 0); K1 % 32 + K2 <= 32 (the ragged end of [A1 | A2] is copied into one padded slab), otherwise
 * RM_EUNSUPPORTED.  dot_w [N] + dot_out [M] (or both NULL; dot_w0 [1] or NULL): also dot_out[b] = C[b, :] . dot_w
 * + dot_w0 from the epilogue's registers - the [*, 1] output projection of the DNN (rm_rowdot) without another
 * pass over C; fixed summation order.  workspace: rm_dense6_workspace(K, N, M) floats, 16-byte aligned. */
int64_t rm_dense6_workspace(int K, int N, int64_t M);
/* rm_dense_wgrad6: rm_dense_wgrad (dW[K,N] (+)= [A1 | A2]^T . G, db[N] = column sums of G or NULL) on the same
 * scheme: both activations split into bf16 pieces on the fly, once per block and 32-row slab, through transposed
 * LDS planes; partial tiles per batch split in the workspace, added in split order (deterministic).  Any K / N /
 * strides.  G2 [M, N2] + dW2 [K, N2] (or NULL / 0): a second piece of gradient columns against the SAME activations in
 * the same pass (DCN: the cross net's coefficient columns beside the first dense layer's dA - x0 read once).
 * workspace: rm_dense_wgrad6_workspace(K, N + N2, M) floats, 16-byte aligned. */
int64_t rm_dense_wgrad6_workspace(int K, int N, int64_t M);
int rm_dense_wgrad6(const float *A1, int64_t lda1, int K1, const float *A2, int64_t lda2, int K2, const float *G,
                    int64_t ldg, int N, const float *G2, int64_t ldg2, int N2, int64_t M, float *dW, int64_t lddw,
                    float *dW2, int64_t lddw2, int accumulate, float *db, float *workspace,
                    int64_t workspace_floats, rm_stream_t stream);
int rm_dense_fwd6(const float *A1, int64_t lda1, int K1, const float *A2, int64_t lda2, int K2, const float *W,
                  int64_t ldw, int w_transposed, int N, const float *bias, int epilogue, int act,
                  const float *aux1, int64_t ld_aux1, int64_t M, float *C, int64_t ldc, const float *dot_w,
                  const float *dot_w0, float *dot_out, float *workspace, rm_stream_t stream);

int64_t rm_dense_wgrad_workspace(int K, int N, int64_t M);
int rm_dense_wgrad(const float *A1, int64_t lda1, int K1, const float *A2, int64_t lda2, int K2,
                   const float *G, int64_t ldg, int N, int64_t M, float *dW, int64_t lddw,
                   int accumulate, float *db, float *workspace, int64_t workspace_floats,
                   rm_stream_t stream);

/* ------------------------------------------------------------------------
 * Multi-valued tag-list features (MultiValCsvFeat, inputs.py:380-425): the sqrtn-pooled
 * lookup tf.nn.embedding_lookup_sparse(..., combiner="sqrtn") of layers.py:144-169 and the
 * multi-hot linear input of utils.py:86-108.  CSR input: example b owns tag ids
 * ids[offsets[b] .. offsets[b+1]) (0 = unknown tag).  rm_pool_rows writes one pooled FUSED
 * row per example: out[b, 0..D) = sum emb / sqrt(n), out[b, D] = sum bias / sqrt(n),
 * out[b, D+1] = sum over known tags (id >= 1) of the linear weight; the gather kernel then
 * reads that row like any other.  rm_pool_rows_bwd scatters a row gradient (d_rows rows of
 * dr_stride floats, g_bias / g_lin [B] or NULL) back to the tag rows of dense gradient
 * buffers d_table [R,D], d_bias [R], d_lin [R] (float atomics; NULL = skip).
 *
 * vals (float [nnz] or NULL): per-id weights.  With vals the pair implements the
 * value-weighted lookup of SparseValueFeat (inputs.py:213-278; `feat_embeds * value`,
 * layers.py:129-142; `one_hot * value`, utils.py:70-71): out[b, 0..D) = sum v*emb,
 * out[b, D] = sum bias (NOT scaled, as layers.py:136-140), out[b, D+1] = sum v*lin; no sqrtn
 * factor and slot 0 is kept.  A SparseValueFeat is the CSR with one id per example. */
int rm_pool_rows(const float *rows, int64_t row0, int LD, int D, const int64_t *offsets,
                 const int64_t *ids, const float *vals, int64_t B, float *out, rm_stream_t stream);
int rm_pool_rows_bwd(const float *d_rows, int64_t dr_stride, const float *g_bias, const float *g_lin,
                     int D, const int64_t *offsets, const int64_t *ids, const float *vals, int64_t B,
                     int64_t row0, float *d_table, float *d_bias, float *d_lin, rm_stream_t stream);
/* The same pooling in PADDED form, for the row-sharded table's fixed-capacity exchange (recman_amd/dist.py): the
 * tags of a feature are T columns of the occurrence matrix - ids[b * ids_ld + t], -1 = no tag - and the row of tag
 * (b, t) is rows[pos[b * pos_ld + t]] (a position among the rows received from their owners; rows are LD floats).
 * vals[b * vals_ld + t] or NULL as above.  rm_pool_rows_padded writes the pooled rows out [B, LD] (same sums,
 * same order as rm_pool_rows); rm_pack_pooled_grad_rows writes each tag's gradient row
 * [d_rows[b] * we | g_bias[b] * wb | g_lin[b] * wl | 0 ..] (width floats; the factors of rm_pool_rows_bwd) at
 * out[pos[b, t]]: the send buffer of the backward all_to_all, every tag its own slot, no atomics. */
int rm_pool_rows_padded(const float *rows, int LD, int D, const int64_t *pos, int64_t pos_ld, const int64_t *ids,
                        int64_t ids_ld, const float *vals, int64_t vals_ld, int64_t B, int T, float *out,
                        rm_stream_t stream);
int rm_pack_pooled_grad_rows(const float *d_rows, int64_t dr_stride, const float *g_bias, const float *g_lin,
                             int D, const int64_t *pos, int64_t pos_ld, const int64_t *ids, int64_t ids_ld,
                             const float *vals, int64_t vals_ld, int64_t B, int T, int width, float *out,
                             rm_stream_t stream);

/* ------------------------------------------------------------------------
 * Optimizer steps.  Replace optimizer.minimize(...) of xDeepFM.py:121-126 / create_optimizer
 * (utils.py:201-213): Keras Adam (kind 0: beta1/beta2, epsilon outside the sqrt), Adagrad (kind 1,
 * accumulator starts at 0.1) or SGD (kind 2).  reset != 0 ignores the stored moments (the reference
 * builds a new optimizer for every batch); step >= 1 is the Adam bias-correction step.
 *
 * rm_sparse_optimizer_step: ROW-WISE and LAZY step on table rows, straight from the IndexedSlices
 * form the backward produces (LazyAdam - Keras' sparse Adam decays the moments of EVERY row; the two
 * coincide under `reset` and when every row is touched each step; see csrc/optim.hip).  Only rows
 * occurring in idx are touched; duplicate occurrences are summed in occurrence order (a stable sort,
 * no float atomics: bit-reproducible).  Occurrences with idx < 0 are skipped.
 *   rows [R, ld]: parameter rows [D embedding | bias | lin | m_bias | m_lin | v_bias | v_lin | pad pad ..],
 *                 ld >= D + 8, ld % 4 == 0 (the four moment entries live in the row's padding), 8 <= D <= 64
 *   mom  [R, 2D]: moments of the embedding entries, interleaved per float4 slice
 *                 [m[0:4] v[0:4] | m[4:8] v[4:8] | ..] (Adagrad uses the v halves; SGD: may be NULL)
 *   d_rows [B,F,D], g_bias / g_lin [B] (per-example gradient of column D / D+1, or NULL),
 *   lin_field_mask [F] or NULL (linear_features subsets)
 *   l2_embedding / l2_linear: LAZY l2 (0 = none) - the gradient of FeatEmbedding.l2 / LinearLayer.l2
 *                 (reg * 1/2 |.|^2, layers.py:188-193, 349-354) is added as reg * row (resp. reg * linear entry)
 *                 for the rows this batch touches, once per distinct row and step, inside the same pass (one FMA
 *                 per element, no extra traffic).  The reference's dense term touches EVERY row every step (the
 *                 whole table, 1.7 - 25.6 GB at the BASELINE configs): the two coincide when every row is touched
 *                 each step; DESIGN.md section 6.
 *   workspace: rm_sparse_optimizer_workspace(B * F) BYTES.
 *   max_field_rows: 0, or a promise that lets the sort work per field: field f owns the rows
 *                 [field_off[f], field_off[f+1]) (the last one up to R), ascending and disjoint, none with more
 *                 than max_field_rows rows, F <= 64.  The ids are then sorted as F independent lists of B keys of
 *                 log2(max_field_rows) bits (csrc/optim.hip, "field-segmented sort": 2 passes for < 2^20 rows per
 *                 field instead of 4 over log2(R) bits); an id beyond its own field's rows is skipped like a
 *                 negative one.  0: one sort of all B * F (row, occurrence) pairs, any field_off.
 * rm_sparse_optimizer_prepare: the id-only part of a step (keys + stable sort by table row) on its own,
 *   so that it can be issued before / beside the forward+backward pass; the following step call on the
 *   same ids passes prepared = 1 and the same workspace (ids: idx [n/F, F] + field_off, or row_ids [n]).
 * rm_sparse_optimizer_step_rows: the same step from gradient rows that already carry their table row
 *   (the row-sharded table's owner side): row_ids [n] (< 0: skip), grad_rows [n, gw] with columns
 *   [0, D) embedding gradient, D bias gradient, D+1 linear gradient.
 * rm_dense_optimizer_step: the same update rule on a flat parameter buffer (dense parameters). */
int64_t rm_sparse_optimizer_workspace(int64_t n);
int rm_sparse_optimizer_prepare(const int64_t *idx, const int64_t *field_off, const int64_t *row_ids,
                                int64_t n, int F, int64_t R, int64_t max_field_rows, void *workspace,
                                int64_t ws_bytes, rm_stream_t stream);
int rm_sparse_optimizer_step(const int64_t *idx, const int64_t *field_off, const float *d_rows,
                             const float *g_bias, const float *g_lin, int64_t B, int F, int D,
                             int64_t R, float *rows, int64_t ld, float *mom, int step, int kind,
                             float lr, float beta1, float beta2, float eps, int reset, float l2_embedding,
                             float l2_linear, const float *lin_field_mask, int64_t max_field_rows, int prepared,
                             void *workspace, int64_t ws_bytes, rm_stream_t stream);
int rm_sparse_optimizer_step_rows(const int64_t *row_ids, const float *grad_rows, int64_t gw, int64_t n,
                                  int D, int64_t R, float *rows, int64_t ld, float *mom, int step, int kind,
                                  float lr, float beta1, float beta2, float eps, int reset, float l2_embedding,
                                  float l2_linear, int prepared, void *workspace, int64_t ws_bytes,
                                  rm_stream_t stream);
int rm_dense_optimizer_step(float *p, const float *g, float *m, float *v, int64_t n, int step, int kind,
                            float lr, float beta1, float beta2, float eps, int reset, rm_stream_t stream);

/* ------------------------------------------------------------------------
 * Row helpers (owner-side gather and re-ordering for the row-sharded table).
 */
/* Buckets the n = B*F occurrences (b,f) by the owner rank of their global row
 * g = field_off[f] + idx[b,f] (owner g % world, local row g / world) - a deterministic
 * counting sort:  pos[o] = position of occurrence o in the bucketed order,
 * send_ids[pos[o]] = g / world, counts[w] = occurrences owned by rank w.  An EMPTY occurrence (idx < 0: the
 * padding of a multi-valued feature's tag columns, see rm_pool_rows_padded) takes no slot: pos[o] = -1, counted
 * nowhere.  workspace: rm_shard_route_workspace(world) int32 words.  world <= 16. */
int64_t rm_shard_route_workspace(int world);
int rm_shard_route(const int64_t *idx, const int64_t *field_off, int64_t B, int F, int world,
                   int64_t *pos, int64_t *send_ids, int64_t *counts, int32_t *workspace,
                   rm_stream_t stream);
/* Fixed-capacity variant: bucket w occupies slots [w*cap, (w+1)*cap) of pos / send_ids
 * (send_ids has world*cap entries; unused slots hold -1, which rm_gather_rows answers with a
 * zero row).  Every rank then exchanges equal splits: no count exchange, no host sync, and the
 * whole step can be captured in one hipGraph.  A bucket that needs more than cap slots sets
 * *overflow = 1 (sticky; the caller clears it) and the step's result must be discarded. */
int rm_shard_route_padded(const int64_t *idx, const int64_t *field_off, int64_t B, int F, int world,
                          int64_t cap, int64_t *pos, int64_t *send_ids, int64_t *counts,
                          int32_t *overflow, int32_t *workspace, rm_stream_t stream);

/* out[pos[o], :] = [d_rows[o, 0..D) | g_bias[b] | g_lin[b] * lin_field_mask[f] | 0 ..] (width floats per
 * row): the per-occurrence gradient rows of the fused table rows, written straight in bucketed order
 * (the send buffer of the backward all_to_all).  g_bias / g_lin [B] may be NULL (0); lin_field_mask
 * [F] or NULL: the linear_features subset (get_linear_features, utils.py:27-30). */
int rm_pack_grad_rows(const float *d_rows, const float *g_bias, const float *g_lin,
                      const float *lin_field_mask, const int64_t *pos, int64_t B, int F, int D, int width,
                      float *out, rm_stream_t stream);

/* rows_out[i, 0..width) = table[rows[i], 0..width) for i < n (owner-side gather of the requested
 * local rows; table rows are table_ld floats apart - the shard keeps optimizer state behind the
 * exchanged columns -, width % 4 == 0, table_ld % 4 == 0); rows[i] < 0 gives a zero row. */
int rm_gather_rows(const float *table, int64_t table_ld, const int64_t *rows, int64_t n, int width,
                   float *rows_out, rm_stream_t stream);

/* dst[i,:] = src[slot[i],:] (un-route: received rows back into (b,f) order) when
 * inverse == 0;  dst[slot[i],:] = src[i,:] when inverse != 0 (route gradient rows). */
int rm_permute_rows(const float *src, const int64_t *slot, int64_t n, int width, int inverse,
                    float *dst, rm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RECMAN_HIP_H */
