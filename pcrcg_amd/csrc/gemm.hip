// gemm.hip -- fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate,
// bit-for-bit an fmaf chain -- MI355X guide section 3), with the per-row scale / bias epilogues the
// KPConv contraction and the 1x1 convolutions need:
//
//   C[m,n] = (sum_k A[m,k] * B[k,n]) * row_scale[m] + bias[n]
//
// Replaces torch.matmul(weighted_features, self.weights).sum(0) / neighbor_num
// (ref:models/blocks.py:361-372), nn.Linear (ref:models/blocks.py:487) and the 1x1 nn.Conv1d layers
// (ref:models/architectures.py:528,538-539; ref:models/gcn.py).
//
// Tiling: 256 threads = 4 wavefronts; block tile BM x BN x 16; A and B tiles are staged through
// registers into k-major LDS images so that the 32 lanes of an MFMA operand row read consecutive
// words (conflict-free ds_read_b32); the next tile's global loads are issued before the current
// tile's MFMAs.  Split-K (grid.z) with fp32 atomics fills the chip when M*N is small and K is large
// (coarse KPConv levels: M = 763, K = 7680).
#include "common.h"

namespace pcrcg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 16;

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(256) k_gemm_f32(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                   int ldb, float* __restrict__ C, int ldc, int M, int N, int Kdim,
                                                   const float* __restrict__ row_scale, const float* __restrict__ bias,
                                                   int k_per_split, int vec_a, int vec_b, int atomic_out) {
    static_assert(WAVES_M * WAVES_N == 4, "4 wavefronts per block");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;  // per-wave tile
    constexpr int TM = WM / 32, TN = WN / 32;            // 32x32 MFMA tiles per wave
    constexpr int LDA_S = BM + 4, LDB_S = BN + 4;
    constexpr int A_ITERS = BM * BK / 4 / 256;           // float4 loads per thread
    constexpr int B_ITERS = BN * BK / 4 / 256;
    static_assert(A_ITERS >= 1 && B_ITERS >= 1, "tile too small");
    __shared__ __attribute__((aligned(16))) float As[BK][LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs[BK][LDB_S];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_begin = blockIdx.z * k_per_split;
    const int k_end = min(Kdim, k_begin + k_per_split);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ra[A_ITERS], rb[B_ITERS];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
            const int e = tid + it * 256;            // float4 index in the BM x (BK/4) tile
            const int row = e / (BK / 4), k4 = (e % (BK / 4)) * 4;
            const int gm = m0 + row, gk = k0 + k4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gm < M) {
                const float* p = A + (long)gm * lda + gk;
                if (vec_a && gk + 3 < k_end) v = *reinterpret_cast<const float4*>(p);
                else {
                    if (gk < k_end) v.x = p[0];
                    if (gk + 1 < k_end) v.y = p[1];
                    if (gk + 2 < k_end) v.z = p[2];
                    if (gk + 3 < k_end) v.w = p[3];
                }
            }
            ra[it] = v;
        }
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int e = tid + it * 256;            // float4 index in the BK x (BN/4) tile
            const int kr = e / (BN / 4), n4 = (e % (BN / 4)) * 4;
            const int gk = k0 + kr, gn = n0 + n4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gk < k_end) {
                const float* p = B + (long)gk * ldb + gn;
                if (vec_b && gn + 3 < N) v = *reinterpret_cast<const float4*>(p);
                else {
                    if (gn < N) v.x = p[0];
                    if (gn + 1 < N) v.y = p[1];
                    if (gn + 2 < N) v.z = p[2];
                    if (gn + 3 < N) v.w = p[3];
                }
            }
            rb[it] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
            const int e = tid + it * 256;
            const int row = e / (BK / 4), k4 = (e % (BK / 4)) * 4;
            As[k4 + 0][row] = ra[it].x;
            As[k4 + 1][row] = ra[it].y;
            As[k4 + 2][row] = ra[it].z;
            As[k4 + 3][row] = ra[it].w;
        }
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int e = tid + it * 256;
            const int kr = e / (BN / 4), n4 = (e % (BN / 4)) * 4;
            *reinterpret_cast<float4*>(&Bs[kr][n4]) = rb[it];
        }
    };

    if (k_begin < k_end) load_tiles(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        __syncthreads();   // previous tile fully consumed
        store_tiles();
        __syncthreads();
        if (k0 + BK < k_end) load_tiles(k0 + BK);   // overlaps with the MFMAs below
        const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[kk + half][wm * WM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[kk + half][wn * WN + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int l31 = lane & 31, half = lane >> 5;
    const bool first_split = blockIdx.z == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gn = n0 + wn * WN + j * 32 + l31;
            if (gn >= N) continue;
            const float bv = (bias && first_split) ? bias[gn] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (gm >= M) continue;
                float v = acc[i][j][r];
                if (row_scale) v *= row_scale[gm];
                v += bv;
                float* dst = C + (long)gm * ldc + gn;
                if (atomic_out) atomicAdd(dst, v);
                else *dst = v;
            }
        }
}

}  // namespace
}  // namespace pcrcg

using namespace pcrcg;

extern "C" int pcrcg_gemm_f32(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n,
                              int k, const float* row_scale, const float* bias, void* stream) {
    PCRCG_CHECK_ARG(m >= 0 && n >= 0 && k >= 0);
    if (m == 0 || n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(a && b && c);
    PCRCG_CHECK_ARG(lda >= k && ldb >= n && ldc >= n);
    hipStream_t st = as_stream(stream);
    const int vec_a = (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0);
    const int vec_b = (ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(b) & 15) == 0);
    const bool narrow = n <= 64;
    const int BM = 128, BN = narrow ? 64 : 128;
    const int gx = (n + BN - 1) / BN, gy = (m + BM - 1) / BM;
    // split K until the grid covers the 256 CUs about twice
    int splits = 1;
    const int ktiles = (k + BK - 1) / BK;
    while (gx * gy * splits < 384 && splits * 2 <= ktiles / 8 && splits < 32) splits *= 2;
    int k_per_split = ((ktiles + splits - 1) / splits) * BK;
    if (k_per_split < BK) k_per_split = BK;
    splits = k > 0 ? (k + k_per_split - 1) / k_per_split : 1;
    if (splits < 1) splits = 1;
    const int atomic_out = splits > 1;
    if (atomic_out) {
        if (ldc == n) PCRCG_CHECK_HIP(hipMemsetAsync(c, 0, (size_t)m * n * sizeof(float), st));
        else PCRCG_CHECK_HIP(hipMemset2DAsync(c, (size_t)ldc * sizeof(float), 0, (size_t)n * sizeof(float), m, st));
    }
    dim3 grid(gx, gy, splits);
    if (narrow)
        hipLaunchKernelGGL((k_gemm_f32<128, 64, 4, 1>), grid, dim3(256), 0, st, a, lda, b, ldb, c, ldc, m, n, k,
                           row_scale, bias, k_per_split, vec_a, vec_b, atomic_out);
    else
        hipLaunchKernelGGL((k_gemm_f32<128, 128, 2, 2>), grid, dim3(256), 0, st, a, lda, b, ldb, c, ldc, m, n, k,
                           row_scale, bias, k_per_split, vec_a, vec_b, atomic_out);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
