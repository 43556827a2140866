/* pcrcg_train.h -- C ABI of the rows SURVEY.md 8f marks "next": the training-side neighbours of the hot
 * path (loss labels, ground-truth correspondences, backward kernels).  Same conventions as pcrcg.h:
 * extern "C", plain device pointers and sizes, row-major, `stream` = hipStream_t as void*, return
 * PCRCG_OK or a negative PCRCG_E* code (pcrcg_last_error() has the text).
 *
 * gfx950 (MI355X) only; there is no CPU implementation behind these entry points. */
#ifndef PCRCG_TRAIN_H
#define PCRCG_TRAIN_H
#include "pcrcg.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Saliency labels of MetricLoss.forward (ref:lib/loss.py:209-213): for every row i of a [n,c] the index
 *   arg[i] = argmax_j <a_i, b_j>  over the rows of b [m,c]
 * i.e. `torch.matmul(a, b.T).max(1)` without materialising the n x m score matrix (n, m = points in the
 * overlap region, up to tens of thousands).  First index wins ties.  best (optional) receives the maximum. */
size_t pcrcg_feature_argmax_ws_bytes(int n);
int pcrcg_feature_argmax(const float* a, int lda, int n, const float* b, int ldb, int m, int c, int64_t* arg,
                         float* best, void* ws, size_t ws_bytes, void* stream);

/* Ground-truth correspondences (replaces get_correspondences, ref:lib/benchmark_utils.py:121-134: an open3d KD-tree
 * radius search per source point in a Python loop).  `grid` = pcrcg_cellgrid_build over the m target points as ONE
 * cloud with a radius slightly above `radius` (the fp32 candidate search must not lose a pair that is inside in
 * float64; pcrcg_amd/correspondences.py uses radius * (1 + 1e-4)).  trans: HOST pointer to the 4x4 row-major float64
 * transform (open3d holds points and transform in float64; so does this kernel: p = R * src_i + t and every candidate
 * distance are evaluated in float64, hit <=> |tgt_j - p| < radius).
 *   _rows : per source point the hits ordered by (distance, target index), the first `keep` of them if keep > 0, into
 *           stage [n, cols] (int32), their number into counts [n]; max_count (device int, zeroed by the caller) receives
 *           the longest untruncated list: if it exceeds `cols` (or the 1024 hits a row can stage) the rows are incomplete
 *           and the caller re-runs with more columns (or gives up).
 *   _emit : out [sum(counts), 2] int64 = (source index, target index), source-major; offsets [n] = exclusive scan of
 *           counts (int64). */
int pcrcg_correspondences_rows(const float* src, int n, const double* trans, double radius, int keep, int m,
                               const void* grid, int cols, int* stage, int* counts, int* max_count, void* stream);
int pcrcg_correspondences_emit(const int* stage, int cols, const int* counts, const int64_t* offsets, int n, int64_t* out,
                               void* stream);

/* MetricLoss's dense parts with their gradients (ref:lib/loss.py:71-135; csrc/lossops.hip).
 * pcrcg_circle_loss: n <= 512 matched descriptor pairs a, b [n, c] (c <= 64) and their coordinate distances
 *   coords_dist [n, n]: out2[0] = get_circle_loss (:71-104, NaN when no row or no column holds both a positive and a
 *   negative, as the reference's mean over nothing), out2[1] = get_recall (:106-116); da, db [n, c] dense (both or neither)
 *   receive d circle_loss / d a, d b.  a, b 16-byte aligned (lda, ldb multiples of 4 when c is); ws:
 *   pcrcg_circle_loss_ws_bytes(n).  Two launches of 2n workgroups.
 * pcrcg_weighted_bce: get_weighted_bce_loss (:118-135) over n predictions in (0, 1) and labels gt: out3 = (loss,
 *   precision, recall) with sklearn's binary definition (0/0 -> 0); grad [n] (may be NULL) = d loss / d prediction as
 *   torch's binary_cross_entropy backward defines it.  ws: pcrcg_weighted_bce_ws_bytes(). */
int pcrcg_circle_loss(const float* a, int lda, const float* b, int ldb, const float* coords_dist, int ldc, int n, int c,
                      float pos_radius, float safe_radius, float pos_optimal, float neg_optimal, float pos_margin,
                      float neg_margin, float log_scale, float* out2, float* da, float* db, void* ws, size_t ws_bytes,
                      void* stream);
size_t pcrcg_circle_loss_ws_bytes(int n);
size_t pcrcg_weighted_bce_ws_bytes(void);
/* The optimiser step of the train loop over FLAT buffers of n floats (parameters, their gradients, the momentum buffer, all
 * 16-byte aligned): torch.optim.SGD(lr, momentum, weight_decay) with dampening 0 and no Nesterov momentum (ref:main.py:59-66;
 * a zeroed momentum buffer gives torch's first step), d = g + wd p; m = mu m + d; p -= lr m; zero_grads != 0 also clears
 * the gradients for the next accumulation.  One launch. */
int pcrcg_sgd_step(float* params, float* grads, float* momentum_buf, long n, float lr, float momentum, float weight_decay,
                   int zero_grads, void* stream);
int pcrcg_weighted_bce(const float* prediction, const float* gt, int n, float* out3, float* grad, void* ws, size_t ws_bytes,
                       void* stream);
/* validate_gradient (ref:lib/utils.py:100-111: no NaN, no Inf in any parameter gradient) over a flat buffer of n floats
 * (16-byte aligned) in one pass: flag[0] (device) = 1.0f when any value is not finite, else 0.0f. */
int pcrcg_nonfinite_flag(const float* x, long n, float* flag, void* stream);
/* The train step's re-packed weight layouts in ONE launch, and the way back for their gradients in one more (the host side
 * of pcrcg_kpfcnn_train_forward keeps K-contiguous KPConv weights, the DGCNN edge convolutions' [Wa - Wb ; Wb] split
 * (ref:models/gcn.py:31-60 applied to cat(x_i, x_j - x_i)), head-major attention projections (ref:models/gcn.py:139-160) and
 * row-padded decoder weights beside the parameters as the reference stores them; every step they follow the parameters).
 * jobs: n_jobs records IN DEVICE MEMORY; job j computes, for i < n,  dst[i] = src[m1[i]] + s2 * src[m2[i]]   (an index of -1
 * contributes zero, m2 may be NULL), or adds that to dst[i] when accumulate != 0.  The maps make every dst element the
 * target of exactly one i, so the result is a function of the inputs alone.  A job with m1 == NULL is a plain transpose:
 * src [n / cols][cols] -> dst [cols][n / cols].  max_n = the largest n of the table. */
typedef struct pcrcg_gather_job {
    const float* src;
    float* dst;
    const int* m1;
    const int* m2;
    int n;
    float s2;
    int accumulate;
    int cols; /* transpose jobs only */
} pcrcg_gather_job;
int pcrcg_gather_jobs(const void* jobs, int n_jobs, int max_n, void* stream);

/* pcrcg_gemm_f32 with an optionally transposed A:  C = (Aop * Bop) * row_scale[m] + bias[n],
 * Aop = A ([M,K] row-major, lda >= K) or A^T (A stored [K,M] row-major, lda >= M) when trans_a.
 * The weight gradients dW = X^T * dY of nn.Linear / the 1x1 convolutions / the KPConv contraction
 * (autograd of ref:models/blocks.py:361-372,487) reduce over the points: K = N_points, split over up to
 * 256 blocks with fp32 atomics. */
int pcrcg_gemm_f32_ex(const float* a, int lda, int trans_a, const float* b, int ldb, int trans_b, float* c, int ldc,
                      int m, int n, int k, const float* row_scale, const float* bias, void* stream);
/* The same product with the caller saying which operand holds GRADIENTS (grad_operand: 1 = A, 2 = B, 0 = neither = 
 * pcrcg_gemm_f32_ex).  Gradients of this network live far below fp16's normal range (1e-4 ... 1e-9): in the default
 * arithmetic (pcrcg_gemm_set_mode 1, include/pcrcg.h) that operand is multiplied by 2^16 -- exactly -- before the two-term
 * fp16 split and the factor is taken out of the sums again: rows whose gradients reach 2^-30 (9.3e-10) split as normal
 * fp16 values; rows entirely below that, and values beyond 1, send their tile to the exact bf16 form (the kernel's two range
 * checks, include/pcrcg.h) -- the result is fp32-class at every gradient scale (tests/test_gemm_range_gpu.py: 1e-6 ... 1e-20),
 * naming the operand only decides how often the cheaper loop suffices.  The train-step runner does this for every backward product;
 * the op-by-op autograd mirror (pcrcg_amd/autograd.py) calls this entry point. */
int pcrcg_gemm_f32_grad(const float* a, int lda, int trans_a, const float* b, int ldb, int trans_b, float* c, int ldc,
                        int m, int n, int k, const float* row_scale, const float* bias, int grad_operand, void* stream);

/* ---- backward kernels (autograd of the hot path) -------------------------------
 * Gradients flow to features and parameters only: geometry (points, influence weights) and the neighbour-count
 * normaliser carry none (ref:models/blocks.py:264-372).  Scatter kernels ACCUMULATE into dx with hardware fp32
 * atomics; the caller provides dx zero-initialised ([ns, c] row-major, shadow row not included). */

/* KPConv: dx[idx[q,h], c] += sum_k w[q,h,k] * d_wf[q,k,c]; d_wf [nq, 15*cin] = (dy * 1/n_q) @ W^T comes from
 * pcrcg_gemm_f32 (row_scale = inv_n), the weight gradient dW = wf^T @ (dy * 1/n_q) from pcrcg_gemm_f32_ex. */
int pcrcg_kpconv_backward_dx(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx, int h,
                             int ld_idx, const float* d_wf, int cin, const float* kp, float extent, float* dx,
                             void* stream);
/* Whole-op KPConv for a C caller (KPConv.forward and its autograd, ref:models/blocks.py:229-374):
 *   forward : out [nq,cout] = ((aggregate of x) @ weights [15*cin, cout]) / n_q; `ws` (pcrcg_kpconv_forward_ws_bytes)
 *             keeps the aggregated features and 1/n_q and must be handed unchanged to the backward call;
 *   backward: dweights [15*cin, cout] (overwritten) and dx [ns,cin] (ACCUMULATED into, zero it first); either may be
 *             NULL; `ws` = pcrcg_kpconv_backward_ws_bytes scratch. */
size_t pcrcg_kpconv_forward_ws_bytes(int nq, int ns, int cin);
int pcrcg_kpconv_forward(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx, int h, int ld_idx,
                         const float* x, int cin, const float* kp, float extent, const float* weights, int cout,
                         float* out, int ld_out, void* ws, size_t ws_bytes, void* stream);
size_t pcrcg_kpconv_backward_ws_bytes(int nq, int cin, int cout);
int pcrcg_kpconv_backward(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx, int h, int ld_idx,
                          int cin, const float* kp, float extent, const float* weights, int cout, const float* dy,
                          int ld_dy, const void* fwd_ws, size_t fwd_ws_bytes, float* dx, float* dweights, void* ws,
                          size_t ws_bytes, void* stream);
/* max_pool (ref:models/blocks.py:86-102): dx[idx[q,h*], c] += dy[q,c], h* = first neighbour attaining y[q,c]. */
int pcrcg_gather_max_backward(const float* x, int ns, int c, const int64_t* idx, int nq, int h, int ld_idx,
                              const float* y, const float* dy, float* dx, void* stream);
/* closest_pool (ref:models/blocks.py:71-83): dx[idx[q,0], :] += dy[q, :]. */
int pcrcg_gather_first_backward(const float* dy, int ld_dy, int c, const int64_t* idx, int nq, int ld_idx, int ns,
                                float* dx, void* stream);
/* InstanceNorm over the rows + LeakyReLU(slope) (ref:models/blocks.py:448-462): with xhat = (x-mean)*rstd
 * (stats = pcrcg_instnorm_stats layout) and g = dy * (xhat > 0 ? 1 : slope):
 *   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)). */
size_t pcrcg_instnorm_backward_ws_bytes(int c);
int pcrcg_instnorm_backward(const float* x, int n, int c, int ldx, const float* stats, const float* dy, int ld_dy,
                            float slope, float* dx, int ld_dx, void* ws, size_t ws_bytes, void* stream);
/* Multi-head attention backward in ONE launch (round 5; forward: pcrcg_attention, include/pcrcg.h; ref:models/gcn.py:151-155,
 * out[:, h d:(h+1) d] = softmax(scale q_h k_h^T) v_h with head-major column blocks).  Given q [n, heads d], k / v [ms, heads d],
 * the forward's out and d_out = dL/d out, ADDS dL/dq, dL/dk, dL/dv to dq / dk / dv (float atomics: the caller zeroes them
 * or holds other contributions there).  pcrcg_attention_backward_supported: head widths 32 / 64 / 128, ms <= 1216 (the score
 * tile of a 32-query workgroup in LDS), leading dimensions multiples of 4, not under deterministic=1 -- the train tape keeps
 * the per-head product path (pcrcg_gemm_f32_ex, pcrcg_softmax_rows_backward) for everything else. */
int pcrcg_attention_backward_supported(int n, int ms, int d, int ldq, int ldk, int ldv, int ld_do);
int pcrcg_attention_backward(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* out, int ldo,
                             const float* d_out, int ld_do, float* dq, int ld_dq, float* dk, int ld_dk, float* dv, int ld_dv,
                             int n, int ms, int heads, int d, float scale, void* stream);
/* Row softmax p = softmax(s * scale): ds = scale * p * (dp - sum_j p*dp) (ref:models/gcn.py:151-155). */
int pcrcg_softmax_rows_backward(const float* p, int ld_p, const float* dp, int ld_dp, int rows, int cols, float scale,
                                float* ds, int ld_ds, void* stream);

/* DGCNN edge conv of the GNN head (ref:models/gcn.py:37-64,121-129; forward = pcrcg_edgeconv_reduce +
 * pcrcg_instnorm_apply): y[i,c] = lrelu(max_j IN2d(ctr[i,c] + nbr[idx[i,j],c])), statistics over all n*k edges.
 * ctr, nbr, dy, dctr, dnbr are contiguous [n,c]; idx int32 [n,k]; dnbr must be zeroed by the caller. */
size_t pcrcg_edgeconv_backward_ws_bytes(int c);
int pcrcg_edgeconv_backward(const float* ctr, const float* nbr, const int* idx, int n, int k, int c, const float* stats,
                            const float* dy, float slope, float* dctr, float* dnbr, void* ws, size_t ws_bytes,
                            void* stream);

/* ---- the train step's network part in two calls (csrc/train_runner.hip) ---------------------------------------
 * KPFCNN.forward with a tape, and its backward (ref:lib/trainer.py:216-265 runs ref:models/architectures.py:181-191,
 * 516-610 under torch.autograd).  `model` as for pcrcg_kpfcnn_forward, with these differences in what the fields hold:
 *   kp_w [15*cin, cout] is used as stored (kp_wt, kp_w_pad, mlp_skip are ignored; cin must be 1 or a multiple of 4);
 *   gnn[].edge1 / edge2 are the packed [2*cout, cin] = [Wa - Wb ; Wb] form and the attention weights head-major,
 *   exactly as the inference descriptor holds them.
 * `grads` is a second pcrcg_model whose POINTER fields hold, for every weight of `model`, the buffer its gradient is
 * ACCUMULATED into (same shape and leading dimension as the weight as handed over; NULL = frozen).  Its integer / float
 * fields are ignored.  The caller maps the gradients of derived layouts (packed edge convolutions, permuted attention
 * weights, padded decoder weights) back to its parameters.
 *   _ws_bytes : sizes of the three regions of the workspace [values | gradients | backward scratch];
 *   _forward  : enqueues the forward, returns device pointers to the three outputs (inside `ws`) and the tape;
 *   _backward : d_* = gradients of the loss wrt the three outputs (device, dense; NULL = zero): clears the gradient
 *               region, enqueues the whole backward; parameter gradients are accumulated into `grads`' buffers and
 *               out->d_inv_temperature holds dL/d(1/temperature) (temperature = exp(epsilon) + 0.03 is the caller's);
 *   _free     : releases the tape (host memory only).
 * `ws` must stay untouched between _forward and _backward; both calls enqueue on `stream` and return without waiting. */
typedef struct pcrcg_train_outputs {
    float* feats_f;            /* [n_points, final_dim] */
    float* scores_overlap;     /* [n_points] */
    float* scores_saliency;    /* [n_points] */
    float* d_inv_temperature;  /* [1], valid after _backward */
    int n_points, final_dim;
} pcrcg_train_outputs;
int pcrcg_kpfcnn_train_ws_bytes(const pcrcg_model* model, const pcrcg_model* grads, const pcrcg_batch* batch,
                                size_t* value_bytes, size_t* grad_bytes, size_t* scratch_bytes);
int pcrcg_kpfcnn_train_forward(const pcrcg_model* model, const pcrcg_model* grads, const pcrcg_batch* batch, void* ws,
                               size_t value_bytes, size_t grad_bytes, size_t scratch_bytes, pcrcg_train_outputs* out,
                               void** tape, void* stream);
int pcrcg_kpfcnn_train_backward(void* tape, const float* d_feats_f, const float* d_scores_overlap,
                                const float* d_scores_saliency, void* stream);
void pcrcg_kpfcnn_train_free(void* tape);

#ifdef __cplusplus
}
#endif
#endif /* PCRCG_TRAIN_H */
