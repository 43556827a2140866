// gnn.hip -- helpers of the GNN / cross-attention overlap head on gfx950 (ref:models/gcn.py).
//
//   pcrcg_knn             get_graph_feature's neighbour selection (:15-34, :48-51) without the N x N
//                         distance matrix: one wavefront per point re-evaluates the distances in each
//                         of the k+1 selection rounds (N is a few hundred at the coarsest level).
//   pcrcg_edgeconv_reduce the DGCNN edge convolution (:37-64, :123-129) without materialising the
//                         [1, C, N, N] tensor of :55 nor the [1, 2C, N, k] edge tensor: the 1x1 conv
//                         over cat(f_i, f_j - f_i) is linear, so the host splits it into a centre term
//                         and a neighbour term (two N x C GEMMs) and this kernel forms
//                         e[i,j,c] = ctr[i,c] + nbr[idx[i,j],c] on the fly, producing max_j e and the
//                         InstanceNorm2d statistics over all (i,j).
//   pcrcg_softmax_rows    attention / saliency softmax (:151-155; ref:models/architectures.py:562-563).
#include "common.h"

namespace pcrcg {

size_t colstats_ws_bytes(int c);
int colstats_finalize(const double* partial, int nchunks, int c, double count, float eps, float* stats,
                      hipStream_t st);
int colstats_chunks();

namespace {

typedef unsigned long long u64;

__global__ void __launch_bounds__(256) k_knn(const float* __restrict__ coords, int n, int k, int* __restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float ax = coords[3 * (long)i], ay = coords[3 * (long)i + 1], az = coords[3 * (long)i + 2];
    const float sa = ax * ax + ay * ay + az * az;
    u64 last = 0;
    bool have_last = false;
    for (int round = 0; round <= k; ++round) {
        u64 best = ~0ull;
        for (int j = lane; j < n; j += 64) {
            const float bx = coords[3 * (long)j], by = coords[3 * (long)j + 1], bz = coords[3 * (long)j + 2];
            const float dot = ax * bx + ay * by + az * bz;
            const float sb = bx * bx + by * by + bz * bz;
            float d = (-2.0f * dot + sa) + sb;            // square_distance :26-31
            d = fmaxf(d, 1e-12f);                         // clamp :33
            const u64 key = ((u64)__float_as_uint(d) << 32) | (unsigned)j;
            if ((!have_last || key > last) && key < best) best = key;
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const u64 o = __shfl_xor(best, s, 64);
            best = o < best ? o : best;
        }
        last = best;
        have_last = true;
        // topk(k+1) sorted ascending, first dropped (:48-49)
        if (round >= 1 && lane == 0) idx[(long)i * k + (round - 1)] = best == ~0ull ? i : (int)(best & 0xFFFFFFFFull);
    }
}

__global__ void __launch_bounds__(256) k_edgeconv_reduce(const float* __restrict__ ctr, int ld_ctr,
                                                          const float* __restrict__ nbr, int ld_nbr,
                                                          const int* __restrict__ idx, int n, int k, int c,
                                                          float* __restrict__ emax, int ld_emax,
                                                          double* __restrict__ partial) {
    __shared__ double s_sum[4][64], s_sq[4][64];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int ch = blockIdx.y * 64 + lane;
    const int chunk = blockIdx.x, nchunks = gridDim.x;
    const int rows_per = (n + nchunks - 1) / nchunks;
    const int r0 = chunk * rows_per, r1 = min(n, r0 + rows_per);
    double s = 0.0, sq = 0.0;
    if (ch < c)
        for (int r = r0 + rl; r < r1; r += 4) {
            const float q = ctr[(long)r * ld_ctr + ch];
            float m = 0.f;
            for (int j = 0; j < k; ++j) {
                const float v = q + nbr[(long)idx[(long)r * k + j] * ld_nbr + ch];
                m = j == 0 ? v : fmaxf(m, v);
                s += (double)v;
                sq += (double)v * (double)v;
            }
            emax[(long)r * ld_emax + ch] = m;
        }
    s_sum[rl][lane] = s;
    s_sq[rl][lane] = sq;
    __syncthreads();
    if (rl == 0 && ch < c) {
        s = (s_sum[0][lane] + s_sum[1][lane]) + (s_sum[2][lane] + s_sum[3][lane]);
        sq = (s_sq[0][lane] + s_sq[1][lane]) + (s_sq[2][lane] + s_sq[3][lane]);
        partial[(long)ch * nchunks + chunk] = s;                       // layout [2][c][nchunks]
        partial[((long)c + ch) * nchunks + chunk] = sq;
    }
}

// one wavefront per row
__global__ void __launch_bounds__(256) k_softmax_rows(float* __restrict__ x, int rows, int cols, int ld, float scale) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float* row = x + (long)r * ld;
    float m = -INFINITY;
    for (int j = lane; j < cols; j += 64) m = fmaxf(m, row[j] * scale);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    float sum = 0.f;
    for (int j = lane; j < cols; j += 64) {
        const float e = expf(row[j] * scale - m);
        row[j] = e;
        sum += e;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) sum += __shfl_xor(sum, s, 64);
    const float inv = 1.0f / sum;
    for (int j = lane; j < cols; j += 64) row[j] *= inv;
}

}  // namespace
}  // namespace pcrcg

using namespace pcrcg;

extern "C" {

int pcrcg_knn(const float* coords, int n, int k, int* idx, void* stream) {
    PCRCG_CHECK_ARG(n >= 0 && k >= 1);
    if (n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(coords && idx);
    hipLaunchKernelGGL(k_knn, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), coords, n, k, idx);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

size_t pcrcg_edgeconv_ws_bytes(int c) { return colstats_ws_bytes(c); }

int pcrcg_edgeconv_reduce(const float* ctr, int ld_ctr, const float* nbr, int ld_nbr, const int* idx,
                          int n, int k, int c, float eps, float* emax, int ld_emax, float* stats,
                          void* ws, size_t ws_bytes, void* stream) {
    PCRCG_CHECK_ARG(n >= 1 && k >= 1 && c >= 1 && ld_ctr >= c && ld_nbr >= c && ld_emax >= c);
    PCRCG_CHECK_ARG(ctr && nbr && idx && emax && stats && ws);
    const int chunks = colstats_chunks();
    Carver cv(ws, ws_bytes);
    double* partial = cv.take<double>((size_t)chunks * 2 * c);
    PCRCG_CHECK_WS(cv);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(k_edgeconv_reduce, dim3(chunks, (c + 63) / 64), dim3(256), 0, st, ctr, ld_ctr, nbr, ld_nbr,
                       idx, n, k, c, emax, ld_emax, partial);
    return colstats_finalize(partial, chunks, c, (double)n * (double)k, eps, stats, st);
}

int pcrcg_softmax_rows(float* x, int rows, int cols, int ld, float scale, void* stream) {
    PCRCG_CHECK_ARG(rows >= 0 && cols >= 1 && ld >= cols);
    if (rows == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(x != nullptr);
    hipLaunchKernelGGL(k_softmax_rows, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, rows, cols, ld, scale);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
}
