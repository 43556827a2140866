// lossops.hip -- MetricLoss's dense parts as kernels (ref:lib/loss.py:71-135): the circle loss with
// the feature-match recall, and the class-weighted BCE with precision / recall -- each with its gradient, so that the
// autograd wrappers of pcrcg_amd/loss.py enqueue two or three launches where the torch formulation took ~130 small ops
// (3.2 ms of interpreter time per train step).  The data-dependent selections around them (which points lie in the
// overlap, the max_points draw from the host generator) stay with the caller.
//
// Circle loss (ref:lib/loss.py:71-104) on n matched pairs (n <= 512, the `max_points` cap), descriptors a, b [n, c]:
//   fd_ij = sqrt(max(2 - 2 <a_i, b_j>, 1e-12))                      (ref:lib/utils.py:78-97, normalised = True)
//   pos = cd < pos_radius, neg = cd > safe_radius                    (cd = coordinate distances, given)
//   pw = max(0, fd - 1e5 * !pos - pos_optimal), nw = max(0, neg_optimal - (fd + 1e5 * !neg))     (constants for autograd)
//   row i: softplus(logsumexp_j(ls (fd - pos_margin) pw) + logsumexp_j(ls (neg_margin - fd) nw)) / ls, same per column;
//   loss = (mean over rows holding a pos and a neg + mean over such columns) / 2
// and recall (:106-116) = share of rows holding a positive whose nearest descriptor is one.  One workgroup: thread i < n
// owns row i, thread n + j column j (online logsumexp, 2n <= 1024 threads); the second phase turns the row / column
// terms into d loss / d a_i and d loss / d b_j the same way.
#include "common.h"
#include "pcrcg_train.h"

namespace pcrcg {
namespace {

struct CircleCfg { float pos_radius, safe_radius, pos_optimal, neg_optimal, pos_margin, neg_margin, log_scale; };

__device__ __forceinline__ float fdist(const float* __restrict__ x, const float* __restrict__ y, int c) {
    float s = 0.f;
    for (int k = 0; k < c; ++k) s += x[k] * y[k];
    return sqrtf(fmaxf(-2.0f * s + 2.0f, 1e-12f));
}
__device__ __forceinline__ void lse_push(float z, float& mx, float& sum) {      // online logsumexp
    if (z > mx) { sum = sum * expf(mx - z) + 1.0f; mx = z; }
    else sum += expf(z - mx);
}

// LDS: per line (row or column): lse_pos, lse_neg, coefficient g (d loss / d softplus-argument, 0 when the line is not
// selected); out[0] = loss, out[1] = recall
__global__ void __launch_bounds__(1024) k_circle_loss(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                       const float* __restrict__ cd, int ldc, int n, int c, CircleCfg cfg,
                                                       float* __restrict__ out, float* __restrict__ da, float* __restrict__ db) {
    extern __shared__ float sm[];
    float* lse_p = sm;                 // [2n]: rows then columns
    float* lse_n = sm + 2 * n;
    float* coef = sm + 4 * n;
    float* red = sm + 6 * n;           // [4]: selected rows, selected columns, rows with a positive, recalled rows
    __shared__ float s_loss[2];
    const int t = threadIdx.x;
    if (t < 4) red[t] = 0.f;
    if (t < 2) s_loss[t] = 0.f;
    __syncthreads();
    const bool is_row = t < n, is_col = t >= n && t < 2 * n;
    const int me = is_row ? t : t - n;
    const float ls = cfg.log_scale;
    float line_loss = 0.f;
    bool sel = false;
    if (is_row || is_col) {
        const float* mine = is_row ? a + (long)me * lda : b + (long)me * ldb;
        float mp = -INFINITY, sp = 0.f, mn = -INFINITY, sn = 0.f, best = INFINITY;
        int npos = 0, nneg = 0, arg = 0;
        for (int o = 0; o < n; ++o) {
            const float* other = is_row ? b + (long)o * ldb : a + (long)o * lda;
            const float d = is_row ? cd[(long)me * ldc + o] : cd[(long)o * ldc + me];
            const float fd = fdist(mine, other, c);
            const bool pos = d < cfg.pos_radius, neg = d > cfg.safe_radius;
            npos += pos;
            nneg += neg;
            const float pw = fmaxf(0.f, fd - (pos ? 0.f : 1e5f) - cfg.pos_optimal);
            const float nw = fmaxf(0.f, cfg.neg_optimal - (fd + (neg ? 0.f : 1e5f)));
            lse_push(ls * (fd - cfg.pos_margin) * pw, mp, sp);
            lse_push(ls * (cfg.neg_margin - fd) * nw, mn, sn);
            if (fd < best) { best = fd; arg = o; }          // torch.min: first index attaining the minimum
        }
        const float lp = mp + logf(sp), ln = mn + logf(sn);
        lse_p[t] = lp;
        lse_n[t] = ln;
        sel = npos > 0 && nneg > 0;
        const float z = lp + ln;
        line_loss = (z > 20.f ? z : log1pf(expf(z))) / ls;   // F.softplus (threshold 20)
        coef[t] = sel ? 1.0f / (1.0f + expf(-z)) / ls : 0.f;
        if (sel) atomicAdd(&red[is_row ? 0 : 1], 1.0f);
        if (is_row && npos > 0) {                            // recall (:106-116)
            atomicAdd(&red[2], 1.0f);
            if (cd[(long)me * ldc + arg] < cfg.pos_radius) atomicAdd(&red[3], 1.0f);
        }
    }
    __syncthreads();
    const float nrow = red[0], ncol = red[1];
    if (sel) atomicAdd(&s_loss[is_row ? 0 : 1], line_loss / (is_row ? nrow : ncol));
    // d loss / d z of a selected line: mean over the selected lines, half weight for rows and columns each
    if (is_row || is_col) coef[t] = sel ? coef[t] * 0.5f / (is_row ? nrow : ncol) : 0.f;
    __syncthreads();
    if (t == 0) {
        out[0] = 0.5f * (s_loss[0] + s_loss[1]);             // (an empty selection gives nan in the reference: mean of nothing)
        if (nrow == 0.f || ncol == 0.f) out[0] = NAN;
        out[1] = red[3] / (red[2] + 1e-12f);
    }
    // gradients: d loss / d fd_ij = g_row_i * (P_row_ij ls pw - N_row_ij ls nw) + g_col_j * (same with the column terms),
    // P = exp(z_pos - lse_pos) etc.; d fd / d <a_i, b_j> = -1 / fd (0 where the clamp is active)
    if ((is_row || is_col) && da && db) {
        const float* mine = is_row ? a + (long)me * lda : b + (long)me * ldb;
        float acc[64];
        for (int k = 0; k < c; ++k) acc[k] = 0.f;
        for (int o = 0; o < n; ++o) {
            const float* other = is_row ? b + (long)o * ldb : a + (long)o * lda;
            const float d = is_row ? cd[(long)me * ldc + o] : cd[(long)o * ldc + me];
            float s = 0.f;
            for (int k = 0; k < c; ++k) s += mine[k] * other[k];
            const float q = -2.0f * s + 2.0f;
            if (q <= 1e-12f) continue;                       // clamped: no gradient
            const float fd = sqrtf(q);
            const bool pos = d < cfg.pos_radius, neg = d > cfg.safe_radius;
            const float pw = fmaxf(0.f, fd - (pos ? 0.f : 1e5f) - cfg.pos_optimal);
            const float nw = fmaxf(0.f, cfg.neg_optimal - (fd + (neg ? 0.f : 1e5f)));
            const float zp = ls * (fd - cfg.pos_margin) * pw, zn = ls * (cfg.neg_margin - fd) * nw;
            const int ro = is_row ? t : o, co = is_row ? n + o : t;       // the row line and the column line of this entry
            const float g = coef[ro] * (expf(zp - lse_p[ro]) * ls * pw - expf(zn - lse_n[ro]) * ls * nw) +
                            coef[co] * (expf(zp - lse_p[co]) * ls * pw - expf(zn - lse_n[co]) * ls * nw);
            const float gs = -g / fd;                        // d loss / d <a_i, b_j>
            for (int k = 0; k < c; ++k) acc[k] += gs * other[k];
        }
        float* dst = is_row ? da + (long)me * c : db + (long)me * c;
        for (int k = 0; k < c; ++k) dst[k] = acc[k];
    }
}

// ---- class-weighted BCE (ref:lib/loss.py:118-135) -------------------------------------------------------------------
// sums[0..3] = sum gt, true positives, predicted positives, actual positives (doubles, zeroed by the caller)
__global__ void __launch_bounds__(256) k_bce_sums(const float* __restrict__ p, const float* __restrict__ gt, int n,
                                                   double* __restrict__ sums) {
    __shared__ double s[4][256];
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float g = gt[i], v = p[i];
        const bool pred = rintf(v) > 0.5f, tru = rintf(g) > 0.5f;      // torch.round: half to even
        a0 += g;
        a1 += pred && tru;
        a2 += pred;
        a3 += tru;
    }
    s[0][threadIdx.x] = a0; s[1][threadIdx.x] = a1; s[2][threadIdx.x] = a2; s[3][threadIdx.x] = a3;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d)
            for (int k = 0; k < 4; ++k) s[k][threadIdx.x] += s[k][threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x < 4) atomicAdd(&sums[threadIdx.x], s[threadIdx.x][0]);
}
// loss sum into sums[4]; grad[i] = weight_i * (p - g) / max((1 - p) p, 1e-12) / n   (torch's binary_cross_entropy backward)
__global__ void __launch_bounds__(256) k_bce_loss(const float* __restrict__ p, const float* __restrict__ gt, int n,
                                                   double* __restrict__ sums, float* __restrict__ grad) {
    __shared__ double s[256];
    const float w_neg = (float)(sums[0] / (double)n), w_pos = 1.0f - w_neg;
    double acc = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float g = gt[i], v = p[i];
        const float w = g >= 0.5f ? w_pos : w_neg;
        const float lp = fmaxf(logf(v), -100.0f), lq = fmaxf(log1pf(-v), -100.0f);      // F.binary_cross_entropy clamps the logs
        acc += (double)(w * -(g * lp + (1.0f - g) * lq));
        if (grad) grad[i] = w * (v - g) / fmaxf((1.0f - v) * v, 1e-12f) / (float)n;
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(&sums[4], s[0]);
}
__global__ void k_bce_final(const double* __restrict__ sums, int n, float* __restrict__ out) {
    out[0] = (float)(sums[4] / (double)n);
    out[1] = sums[2] > 0 ? (float)(sums[1] / sums[2]) : 0.f;       // precision (0/0 -> 0, as sklearn reports it)
    out[2] = sums[3] > 0 ? (float)(sums[1] / sums[3]) : 0.f;       // recall
}

}  // namespace
}  // namespace pcrcg

using namespace pcrcg;

extern "C" {

int pcrcg_circle_loss(const float* a, int lda, const float* b, int ldb, const float* coords_dist, int ldc, int n, int c,
                      float pos_radius, float safe_radius, float pos_optimal, float neg_optimal, float pos_margin,
                      float neg_margin, float log_scale, float* out2, float* da, float* db, void* stream) {
    PCRCG_CHECK_ARG(n >= 1 && n <= 512 && c >= 1 && c <= 64 && lda >= c && ldb >= c && ldc >= n);
    PCRCG_CHECK_ARG(a && b && coords_dist && out2 && (!da == !db));
    CircleCfg cfg = {pos_radius, safe_radius, pos_optimal, neg_optimal, pos_margin, neg_margin, log_scale};
    hipLaunchKernelGGL(k_circle_loss, dim3(1), dim3(1024), sizeof(float) * (6 * (size_t)n + 8), as_stream(stream), a, lda, b, ldb,
                       coords_dist, ldc, n, c, cfg, out2, da, db);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

size_t pcrcg_weighted_bce_ws_bytes(void) { return 256; }

int pcrcg_weighted_bce(const float* prediction, const float* gt, int n, float* out3, float* grad, void* ws, size_t ws_bytes,
                       void* stream) {
    PCRCG_CHECK_ARG(n >= 1 && prediction && gt && out3 && ws && ws_bytes >= 64);
    hipStream_t st = as_stream(stream);
    double* sums = static_cast<double*>(ws);
    PCRCG_CHECK_HIP(hipMemsetAsync(sums, 0, 64, st));
    int blocks = (n + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(k_bce_sums, dim3(blocks), dim3(256), 0, st, prediction, gt, n, sums);
    hipLaunchKernelGGL(k_bce_loss, dim3(blocks), dim3(256), 0, st, prediction, gt, n, sums, grad);
    hipLaunchKernelGGL(k_bce_final, dim3(1), dim3(1), 0, st, sums, n, out3);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
}
