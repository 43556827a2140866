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
// and recall (:106-116) = share of rows holding a positive whose nearest descriptor is one.  Two launches of 2n
// workgroups, one per LINE (row i = line i, column j = line n + j): k_circle_lines reduces each line (one wave, online
// logsumexp merged across lanes); k_circle_grads counts the selected lines, writes loss and recall (workgroup 0) and turns
// the line terms into d loss / d a_i resp. d loss / d b_j.  The n x n matrices are never stored: a line's <a_i, b_j>
// are recomputed where needed (n * c multiply-adds per line, the descriptors stay in L2).
#include "common.h"
#include "pcrcg_train.h"

namespace pcrcg {
namespace {

struct CircleCfg { float pos_radius, safe_radius, pos_optimal, neg_optimal, pos_margin, neg_margin, log_scale; };

__device__ __forceinline__ float dotc(const float* __restrict__ x, const float* __restrict__ y, int c) {
    float s = 0.f;
    if ((c & 3) == 0) {
        for (int k = 0; k < c; k += 4) {
            const float4 u = *reinterpret_cast<const float4*>(x + k), v = *reinterpret_cast<const float4*>(y + k);
            s += u.x * v.x; s += u.y * v.y; s += u.z * v.z; s += u.w * v.w;
        }
    } else {
        for (int k = 0; k < c; ++k) s += x[k] * y[k];
    }
    return s;
}
__device__ __forceinline__ void lse_push(float z, float& mx, float& sum) {      // online logsumexp
    if (z > mx) { sum = sum * expf(mx - z) + 1.0f; mx = z; }
    else sum += expf(z - mx);
}
__device__ __forceinline__ void lse_merge(float& mx, float& sum, float m2, float s2) {
    const float m = fmaxf(mx, m2);
    if (m == -INFINITY) return;                                // two empty lanes
    sum = sum * expf(mx - m) + s2 * expf(m2 - m);
    mx = m;
}

// per-line record in the workspace, 8 floats: lse_pos, lse_neg, sigmoid(z) / ls, line loss, selected, has a positive,
// recalled, -
constexpr int kLineRec = 8;

__global__ void __launch_bounds__(64) k_circle_lines(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                      const float* __restrict__ cd, int ldc, int n, int c, CircleCfg cfg,
                                                      float* __restrict__ lines) {
    const int line = blockIdx.x, lane = threadIdx.x;
    const bool is_row = line < n;
    const int me = is_row ? line : line - n;
    const float* mine = is_row ? a + (long)me * lda : b + (long)me * ldb;
    const float ls = cfg.log_scale;
    float mp = -INFINITY, sp = 0.f, mn = -INFINITY, sn = 0.f, best = INFINITY;
    int npos = 0, nneg = 0, arg = 0x7fffffff;
    for (int o = lane; o < n; o += 64) {
        const float* other = is_row ? b + (long)o * ldb : a + (long)o * lda;
        const float d = is_row ? cd[(long)me * ldc + o] : cd[(long)o * ldc + me];
        const float q = -2.0f * dotc(mine, other, c) + 2.0f;
        const float fd = sqrtf(q != q ? q : fmaxf(q, 1e-12f));   // torch.clamp propagates NaN (fmaxf would return 1e-12)
        const bool pos = d < cfg.pos_radius, neg = d > cfg.safe_radius;
        npos += pos;
        nneg += neg;
        const float pw = fmaxf(0.f, fd - (pos ? 0.f : 1e5f) - cfg.pos_optimal);
        const float nw = fmaxf(0.f, cfg.neg_optimal - (fd + (neg ? 0.f : 1e5f)));
        lse_push(ls * (fd - cfg.pos_margin) * pw, mp, sp);
        lse_push(ls * (cfg.neg_margin - fd) * nw, mn, sn);
        // torch.min: the first index attaining the minimum, and NaN counts as the minimum (the first NaN of a row wins).
        // The first entry a lane sees is always taken, so `arg` is a valid column whenever the lane saw one.
        if (arg == 0x7fffffff || fd < best || (fd != fd && best == best)) { best = fd; arg = o; }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        lse_merge(mp, sp, __shfl_xor(mp, off), __shfl_xor(sp, off));
        lse_merge(mn, sn, __shfl_xor(mn, off), __shfl_xor(sn, off));
        npos += __shfl_xor(npos, off);
        nneg += __shfl_xor(nneg, off);
        const float b2 = __shfl_xor(best, off);
        const int a2 = __shfl_xor(arg, off);
        const bool nan1 = best != best, nan2 = b2 != b2;
        const bool take = a2 != 0x7fffffff &&
                          (arg == 0x7fffffff || (nan2 && (!nan1 || a2 < arg)) || (!nan1 && !nan2 && (b2 < best || (b2 == best && a2 < arg))));
        if (take) { best = b2; arg = a2; }
    }
    if (lane == 0) {
        const float lp = mp + logf(sp), ln = mn + logf(sn), z = lp + ln;
        const bool sel = npos > 0 && nneg > 0;
        float* r = lines + (long)line * kLineRec;
        r[0] = lp;
        r[1] = ln;
        r[2] = sel ? 1.0f / (1.0f + expf(-z)) / ls : 0.f;                       // d line loss / d z
        r[3] = sel ? (z > 20.f ? z : log1pf(expf(z))) / ls : 0.f;                // F.softplus (threshold 20)
        r[4] = sel ? 1.f : 0.f;
        r[5] = is_row && npos > 0 ? 1.f : 0.f;
        r[6] = is_row && npos > 0 && cd[(long)me * ldc + min(arg, n - 1)] < cfg.pos_radius ? 1.f : 0.f;      // recall (:106-116)
        r[7] = 0.f;
    }
}

// d loss / d fd_ij = g_row_i * (P_row_ij ls pw - N_row_ij ls nw) + g_col_j * (same with the column terms),
// P = exp(z_pos - lse_pos) etc., g = sigmoid(z) / ls * 0.5 / (selected lines of that kind);
// d fd / d <a_i, b_j> = -1 / fd (0 where the clamp is active)
__global__ void __launch_bounds__(256) k_circle_grads(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                       const float* __restrict__ cd, int ldc, int n, int c, CircleCfg cfg,
                                                       const float* __restrict__ lines, float* __restrict__ out,
                                                       float* __restrict__ da, float* __restrict__ db) {
    __shared__ float red[6];                  // selected rows, selected columns, row loss, column loss, rows with a positive, recalled
    __shared__ float gs[512];
    __shared__ float part[256];
    const int line = blockIdx.x, t = threadIdx.x;
    if (t < 6) red[t] = 0.f;
    __syncthreads();
    {
        float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int l = t; l < 2 * n; l += 256) {
            const float* r = lines + (long)l * kLineRec;
            const int col = l >= n;
            v[col] += r[4];
            v[2 + col] += r[3];
            v[4] += r[5];
            v[5] += r[6];
        }
        // the four wavefronts' sums meet in a FIXED order (an LDS atomic here made the loss value depend on which wavefront
        // arrived first: run-to-run differences in its last bits)
        for (int q = 0; q < 6; ++q) {
            float x = v[q];
            for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
            if ((t & 63) == 0) part[(t >> 6) * 6 + q] = x;
        }
        __syncthreads();
        if (t < 6) red[t] = (part[t] + part[6 + t]) + (part[12 + t] + part[18 + t]);
    }
    __syncthreads();
    const float nrow = red[0], ncol = red[1];
    if (line == 0 && t == 0) {
        // (an empty selection gives nan in the reference: mean of nothing)
        out[0] = nrow == 0.f || ncol == 0.f ? NAN : 0.5f * (red[2] / nrow + red[3] / ncol);
        out[1] = red[5] / (red[4] + 1e-12f);
    }
    if (!da) return;
    const bool is_row = line < n;
    const int me = is_row ? line : line - n;
    const float* mine = is_row ? a + (long)me * lda : b + (long)me * ldb;
    const float ls = cfg.log_scale;
    const float* rm = lines + (long)line * kLineRec;
    const float lp_m = rm[0], ln_m = rm[1], g_m = rm[4] != 0.f ? rm[2] * 0.5f / (is_row ? nrow : ncol) : 0.f;
    const float inv_other = 0.5f / (is_row ? ncol : nrow);
    for (int o = t; o < n; o += 256) {
        const float* other = is_row ? b + (long)o * ldb : a + (long)o * lda;
        const float d = is_row ? cd[(long)me * ldc + o] : cd[(long)o * ldc + me];
        const float q = -2.0f * dotc(mine, other, c) + 2.0f;
        float g = 0.f;
        if (q > 1e-12f || q != q) {                            // clamped entries pass no gradient; NaN flows on (as in torch)
            const float fd = sqrtf(q);
            const bool pos = d < cfg.pos_radius, neg = d > cfg.safe_radius;
            const float pw = fmaxf(0.f, fd - (pos ? 0.f : 1e5f) - cfg.pos_optimal);
            const float nw = fmaxf(0.f, cfg.neg_optimal - (fd + (neg ? 0.f : 1e5f)));
            const float zp = ls * (fd - cfg.pos_margin) * pw, zn = ls * (cfg.neg_margin - fd) * nw;
            const float* ro = lines + (long)(is_row ? n + o : o) * kLineRec;      // the crossing line of this entry
            const float g_o = ro[4] != 0.f ? ro[2] * inv_other : 0.f;
            const float gfd = g_m * (expf(zp - lp_m) * ls * pw - expf(zn - ln_m) * ls * nw) +
                              g_o * (expf(zp - ro[0]) * ls * pw - expf(zn - ro[1]) * ls * nw);
            g = -gfd / fd;                                     // d loss / d <a_i, b_j>
        }
        gs[o] = g;
    }
    __syncthreads();
    // d mine[k] = sum_o gs[o] other[o][k]: thread (part, k) over every parts-th o, then across the parts
    const int parts = 256 / c, k = t % c, pt = t / c;
    float acc = 0.f;
    if (pt < parts)
        for (int o = pt; o < n; o += parts) acc += gs[o] * (is_row ? b[(long)o * ldb + k] : a[(long)o * lda + k]);
    part[t] = acc;
    __syncthreads();
    if (t < c) {
        float s = 0.f;
        for (int q = 0; q < parts; ++q) s += part[q * c + t];
        (is_row ? da : db)[(long)me * c + t] = s;
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

// validate_gradient (ref:lib/utils.py:100-111) over the flat gradient bucket in ONE pass: flag[0] = 1 when any value is NaN or
// +-Inf (every offending thread stores the same 1.0f: no atomics needed), else what the caller's memset left (0)
__global__ void __launch_bounds__(256) k_nonfinite_flag(const float* __restrict__ x, long n, float* __restrict__ flag) {
    bool bad = false;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const uint4 v = reinterpret_cast<const uint4*>(x)[i];
        bad |= ((v.x & 0x7f800000u) == 0x7f800000u) | ((v.y & 0x7f800000u) == 0x7f800000u) | ((v.z & 0x7f800000u) == 0x7f800000u) |
               ((v.w & 0x7f800000u) == 0x7f800000u);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad |= (__float_as_uint(x[4 * n4 + threadIdx.x]) & 0x7f800000u) == 0x7f800000u;
    if (bad) flag[0] = 1.0f;
}

// ---- SGD with momentum over flat buffers (torch.optim.SGD, dampening 0, no Nesterov; ref:main.py:59-66) ---------------------
//   d = g + wd * p;  m = mu * m + d;  p = p - lr * m;  optionally g = 0 (the next step's accumulation starts from zero)
__global__ void __launch_bounds__(256) k_sgd_step(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, long n,
                                                   float lr, float mu, float wd, int zero_grad) {
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            const float4 pv = *reinterpret_cast<const float4*>(p + i), gv = *reinterpret_cast<const float4*>(g + i);
            float4 mv = *reinterpret_cast<const float4*>(m + i);
            float4 out;
            mv.x = mu * mv.x + (gv.x + wd * pv.x); out.x = pv.x - lr * mv.x;
            mv.y = mu * mv.y + (gv.y + wd * pv.y); out.y = pv.y - lr * mv.y;
            mv.z = mu * mv.z + (gv.z + wd * pv.z); out.z = pv.z - lr * mv.z;
            mv.w = mu * mv.w + (gv.w + wd * pv.w); out.w = pv.w - lr * mv.w;
            *reinterpret_cast<float4*>(m + i) = mv;
            *reinterpret_cast<float4*>(p + i) = out;
            if (zero_grad) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (long j = i; j < n; ++j) {
                const float mj = mu * m[j] + (g[j] + wd * p[j]);
                m[j] = mj;
                p[j] -= lr * mj;
                if (zero_grad) g[j] = 0.f;
            }
        }
    }
}

}  // namespace
}  // namespace pcrcg

using namespace pcrcg;

extern "C" {

size_t pcrcg_circle_loss_ws_bytes(int n) { return sizeof(float) * kLineRec * 2 * (size_t)(n > 0 ? n : 0); }

int pcrcg_circle_loss(const float* a, int lda, const float* b, int ldb, const float* coords_dist, int ldc, int n, int c,
                      float pos_radius, float safe_radius, float pos_optimal, float neg_optimal, float pos_margin,
                      float neg_margin, float log_scale, float* out2, float* da, float* db, void* ws, size_t ws_bytes,
                      void* stream) {
    PCRCG_CHECK_ARG(n >= 1 && n <= 512 && c >= 1 && c <= 64 && lda >= c && ldb >= c && ldc >= n);
    PCRCG_CHECK_ARG(a && b && coords_dist && out2 && (!da == !db) && ws && ws_bytes >= pcrcg_circle_loss_ws_bytes(n));
    PCRCG_CHECK_ARG(((uintptr_t)a | (uintptr_t)b) % 16 == 0 && (c % 4 != 0 || (lda % 4 == 0 && ldb % 4 == 0)));
    CircleCfg cfg = {pos_radius, safe_radius, pos_optimal, neg_optimal, pos_margin, neg_margin, log_scale};
    hipStream_t st = as_stream(stream);
    float* lines = static_cast<float*>(ws);
    hipLaunchKernelGGL(k_circle_lines, dim3(2 * n), dim3(64), 0, st, a, lda, b, ldb, coords_dist, ldc, n, c, cfg, lines);
    hipLaunchKernelGGL(k_circle_grads, dim3(da ? 2 * n : 1), dim3(256), 0, st, a, lda, b, ldb, coords_dist, ldc, n, c, cfg, lines,
                       out2, da, db);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

int pcrcg_sgd_step(float* params, float* grads, float* momentum_buf, long n, float lr, float momentum, float weight_decay,
                   int zero_grads, void* stream) {
    PCRCG_CHECK_ARG(n >= 0 && (n == 0 || (params && grads && momentum_buf)));
    PCRCG_CHECK_ARG(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)momentum_buf) % 16 == 0);
    if (n > 0) {
        long blocks = (n / 4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_sgd_step, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), params, grads, momentum_buf, n, lr,
                           momentum, weight_decay, zero_grads);
    }
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

int pcrcg_nonfinite_flag(const float* x, long n, float* flag, void* stream) {
    PCRCG_CHECK_ARG(n >= 0 && flag && (n == 0 || x) && (uintptr_t)x % 16 == 0);
    hipStream_t st = as_stream(stream);
    PCRCG_CHECK_HIP(hipMemsetAsync(flag, 0, sizeof(float), st));
    if (n > 0) {
        long blocks = (n / 4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_nonfinite_flag, dim3((unsigned)blocks), dim3(256), 0, st, x, n, flag);
    }
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
    hipLaunchKernelGGL(k_bce_sums, dim3(blocks), dim3(256), 0, st, prediction, gt, n, sums);      // (counts: exact in any order)
    // the loss sum meets in one fp64 atomic per workgroup: deterministic=1 runs ONE workgroup (grid-stride loop, n is a cloud)
    hipLaunchKernelGGL(k_bce_loss, dim3(debug_opts().deterministic ? 1 : blocks), dim3(256), 0, st, prediction, gt, n, sums, grad);
    hipLaunchKernelGGL(k_bce_final, dim3(1), dim3(1), 0, st, sums, n, out3);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
}
