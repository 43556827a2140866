// trainops.hip -- kernels of the training-side rows: loss labels and backward passes.
// ABI: include/pcrcg_train.h.
#include "common.h"
#include "pcrcg_train.h"

namespace pcrcg {
namespace {

// ---- feature arg-max (ref:lib/loss.py:209-213) ---------------------------------------------------
// One thread owns a row of A (C floats in registers); rows of B stream through LDS in tiles of TB rows
// and are read back as wave-uniform (broadcast) float4s, so the inner loop is pure FMA.
template <int C>
__global__ void __launch_bounds__(256) k_feature_argmax(const float* __restrict__ a, int lda, int n,
                                                         const float* __restrict__ b, int ldb, int m,
                                                         long long* __restrict__ arg, float* __restrict__ best) {
    constexpr int TB = 128;
    __shared__ __attribute__((aligned(16))) float bs[TB * C];
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int r = row < n ? row : n - 1;
    float av[C];
#pragma unroll
    for (int k = 0; k < C; ++k) av[k] = a[(long)r * lda + k];
    float bv = -INFINITY;
    int bj = 0;
    for (int j0 = 0; j0 < m; j0 += TB) {
        const int tj = min(TB, m - j0);
        __syncthreads();
        for (int e = threadIdx.x; e < tj * C; e += 256) bs[e] = b[(long)(j0 + e / C) * ldb + e % C];
        __syncthreads();
        for (int j = 0; j < tj; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < C; k += 4) {
                const float4 v = *reinterpret_cast<const float4*>(&bs[j * C + k]);
                s = fmaf(av[k], v.x, s);
                s = fmaf(av[k + 1], v.y, s);
                s = fmaf(av[k + 2], v.z, s);
                s = fmaf(av[k + 3], v.w, s);
            }
            if (s > bv) { bv = s; bj = j0 + j; }
        }
    }
    if (row < n) {
        arg[row] = bj;
        if (best) best[row] = bv;
    }
}

// any width: A rows are re-read from memory (L1/L2 resident), one thread per row
__global__ void __launch_bounds__(256) k_feature_argmax_any(const float* __restrict__ a, int lda, int n,
                                                             const float* __restrict__ b, int ldb, int m, int c,
                                                             long long* __restrict__ arg, float* __restrict__ best) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    float bv = -INFINITY;
    int bj = 0;
    for (int j = 0; j < m; ++j) {
        float s = 0.f;
        for (int k = 0; k < c; ++k) s = fmaf(a[(long)row * lda + k], b[(long)j * ldb + k], s);
        if (s > bv) { bv = s; bj = j; }
    }
    arg[row] = bj;
    if (best) best[row] = bv;
}

}  // namespace
}  // namespace pcrcg

using namespace pcrcg;

extern "C" int pcrcg_feature_argmax(const float* a, int lda, int n, const float* b, int ldb, int m, int c,
                                    int64_t* arg, float* best, void* stream) {
    PCRCG_CHECK_ARG(n >= 0 && m >= 1 && c >= 1 && lda >= c && ldb >= c);
    if (n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(a && b && arg);
    hipStream_t st = as_stream(stream);
    long long* out = reinterpret_cast<long long*>(arg);
    const dim3 grid((n + 255) / 256);
    if (c == 32) hipLaunchKernelGGL(k_feature_argmax<32>, grid, dim3(256), 0, st, a, lda, n, b, ldb, m, out, best);
    else if (c == 64) hipLaunchKernelGGL(k_feature_argmax<64>, grid, dim3(256), 0, st, a, lda, n, b, ldb, m, out, best);
    else hipLaunchKernelGGL(k_feature_argmax_any, grid, dim3(256), 0, st, a, lda, n, b, ldb, m, c, out, best);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
