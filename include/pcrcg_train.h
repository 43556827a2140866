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
int pcrcg_feature_argmax(const float* a, int lda, int n, const float* b, int ldb, int m, int c, int64_t* arg,
                         float* best, void* stream);

/* pcrcg_gemm_f32 with an optionally transposed A:  C = (Aop * Bop) * row_scale[m] + bias[n],
 * Aop = A ([M,K] row-major, lda >= K) or A^T (A stored [K,M] row-major, lda >= M) when trans_a.
 * The weight gradients dW = X^T * dY of nn.Linear / the 1x1 convolutions / the KPConv contraction
 * (autograd of ref:models/blocks.py:361-372,487) reduce over the points: K = N_points, split over up to
 * 256 blocks with fp32 atomics. */
int pcrcg_gemm_f32_ex(const float* a, int lda, int trans_a, const float* b, int ldb, int trans_b, float* c, int ldc,
                      int m, int n, int k, const float* row_scale, const float* bias, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PCRCG_TRAIN_H */
