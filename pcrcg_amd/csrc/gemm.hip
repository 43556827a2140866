// gemm.hip -- fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate,
// bit-for-bit an fmaf chain -- MI355X guide section 3), with the per-row scale / bias epilogues the
// KPConv contraction and the 1x1 convolutions need:
//
//   C[m,n] = (sum_k A[m,k] * Bop[k,n]) * row_scale[m] + bias[n]
//   Bop = B        (B  is [K,N] row-major: KPConv weights [15*Cin, Cout], P @ V)        trans_b = 0
//   Bop = B^T      (B  is [N,K] row-major: nn.Linear / 1x1 conv weights, Q @ K^T)       trans_b = 1
//
// Replaces torch.matmul(weighted_features, self.weights).sum(0) / neighbor_num
// (ref:models/blocks.py:361-372), nn.Linear (ref:models/blocks.py:487) and the 1x1 nn.Conv1d layers
// (ref:models/architectures.py:528,538-539; ref:models/gcn.py:123-132,165-173).
//
// Structure: 256 threads = 4 wavefronts; block tile BM x BN x 32, two LDS stages (dynamic LDS).
//   * A (and B when trans_b) is copied row-major into LDS as float4s with rows padded to 36 floats: a
//     wavefront's ds_read_b128 of 4 consecutive k for 32 consecutive rows is bank-conflict free
//     (36*r mod 64 enumerates all 16 four-bank slots).  The k index is permuted inside each group of
//     8 (MFMA t of a group sums k = 8g+t and 8g+4+t) identically for A and B, so one b128 read per
//     operand tile feeds four MFMAs.
//   * per k-step: tile s is consumed from LDS[s&1] while tile s+1 (already in registers) is written
//     to LDS[(s+1)&1] and tile s+2 is requested from HBM; ONE barrier per step.
//   * interior blocks (full tile, 16-byte aligned operands) use unguarded float4 loads; edge blocks
//     use branch-free clamped loads (a per-element "load or zero" branch makes hipcc wait vmcnt(0)
//     inside every branch).
//   * split-K (grid.z) with fp32 atomics only when even the smallest tile leaves the chip idle.
#include <cstdlib>

#include <atomic>

#include "common.h"
#include "pcrcg_train.h"

namespace pcrcg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// k-step: 32, or 64 for long-K GEMMs (256 contiguous bytes per operand row and step)
template <int BM, int BN, bool TRANS_A, bool TRANS_B, int BK>
constexpr int stage_floats() {
    return (TRANS_A ? BK * (BM + 4) : BM * (BK + 4)) + (TRANS_B ? BN * (BK + 4) : BK * (BN + 4));
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool TRANS_A, bool TRANS_B, int BK>
__global__ void __launch_bounds__(256) k_gemm_f32(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                   int ldb, float* __restrict__ C, int ldc, int M, int N, int Kdim,
                                                   const float* __restrict__ row_scale, const float* __restrict__ bias,
                                                   int k_per_split, int vec_a, int vec_b, int atomic_out,
                                                   double* __restrict__ colp, int colp_chunks) {
    static_assert(WAVES_M * WAVES_N == 4, "4 wavefronts per block");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;  // per-wave tile
    constexpr int TM = WM / 32, TN = WN / 32;            // 32x32 MFMA tiles per wave
    constexpr int KPAD = BK + 4;   // row stride of k-contiguous LDS images (floats); = 4 mod 64
    constexpr int NPAD = BN + 4;
    constexpr int MPAD = BM + 4;   // row stride of the k-major A image (TRANS_A: A is stored [K, M])
    constexpr int A_FLOATS = TRANS_A ? BK * MPAD : BM * KPAD;
    constexpr int A_ITERS = BM * BK / 4 / 256;           // float4 per thread and tile
    constexpr int B_ITERS = BN * BK / 4 / 256;
    constexpr int STAGE = stage_floats<BM, BN, TRANS_A, TRANS_B, BK>();
    static_assert(A_ITERS >= 1 && B_ITERS >= 1, "tile too small");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // XCD-aware tile order: workgroups are dispatched round-robin over the 8 XCDs (private L2s), so the
    // linear id is remapped such that each XCD walks a CONTIGUOUS range of tiles (n fastest): the n-tiles
    // that share a row panel of A then hit the same L2.  Placement only affects speed.
    const int gx = gridDim.x, ntile = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gx * blockIdx.y;
    const int xq = ntile >> 3, xr = ntile & 7, xcd = lin & 7, slot = lin >> 3;
    const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + slot;   // bijective
    const int tile_x = tile % gx, tile_y = tile / gx;
    const int m0 = tile_y * BM, n0 = tile_x * BN;
    const int k_begin = blockIdx.z * k_per_split;
    const int k_end = min(Kdim, k_begin + k_per_split);
    const bool full_a = vec_a && (m0 + BM <= M);   // block-uniform: operand tile fully inside, 16-B aligned
    // (TRANS_A: rows of the stored matrix are k, the tile spans columns m0..m0+BM -- same condition)
    const bool full_b = vec_b && (n0 + BN <= N);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ra[A_ITERS], rb[B_ITERS];

    // rows row0.. of a [rows, K] row-major matrix, columns k0..k0+31 (k-contiguous operand)
    auto kmajor_fast = [&](const float* __restrict__ P, int ld, int row0, int k0, int it) {
        const int e = tid + it * 256;
        const int r = e / (BK / 4), k4 = (e % (BK / 4)) * 4;
        return *reinterpret_cast<const float4*>(P + (long)(row0 + r) * ld + k0 + k4);
    };
    auto kmajor_edge = [&](const float* __restrict__ P, int ld, int row0, int rows, int k0, int it) {
        const int e = tid + it * 256;
        const int r = e / (BK / 4), k4 = (e % (BK / 4)) * 4;
        const int gr = row0 + r, gk = k0 + k4, ke = k_end - 1;
        const float* p = P + (long)min(gr, rows - 1) * ld;     // always a valid address
        const bool rok = gr < rows;
        float4 v;
        v.x = p[min(gk, ke)];
        v.y = p[min(gk + 1, ke)];
        v.z = p[min(gk + 2, ke)];
        v.w = p[min(gk + 3, ke)];
        v.x = (rok && gk < k_end) ? v.x : 0.f;
        v.y = (rok && gk + 1 < k_end) ? v.y : 0.f;
        v.z = (rok && gk + 2 < k_end) ? v.z : 0.f;
        v.w = (rok && gk + 3 < k_end) ? v.w : 0.f;
        return v;
    };
    auto load_tiles = [&](int k0) {
        const bool kfull = k0 + BK <= k_end;
        if (TRANS_A) {           // stored [K, M]: tile rows are k, float4s run along m
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it) {
                const int e = tid + it * 256;
                const int kr = e / (BM / 4), m4 = (e % (BM / 4)) * 4;
                if (full_a && kfull) {
                    ra[it] = *reinterpret_cast<const float4*>(A + (long)(k0 + kr) * lda + m0 + m4);
                } else {
                    const int gk = k0 + kr, gm = m0 + m4, me = M - 1;
                    const float* p = A + (long)min(gk, k_end - 1) * lda;
                    const bool kok = gk < k_end;
                    float4 v;
                    v.x = p[min(gm, me)];
                    v.y = p[min(gm + 1, me)];
                    v.z = p[min(gm + 2, me)];
                    v.w = p[min(gm + 3, me)];
                    v.x = (kok && gm < M) ? v.x : 0.f;
                    v.y = (kok && gm + 1 < M) ? v.y : 0.f;
                    v.z = (kok && gm + 2 < M) ? v.z : 0.f;
                    v.w = (kok && gm + 3 < M) ? v.w : 0.f;
                    ra[it] = v;
                }
            }
        } else if (full_a && kfull) {
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it) ra[it] = kmajor_fast(A, lda, m0, k0, it);
        } else {
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it) ra[it] = kmajor_edge(A, lda, m0, M, k0, it);
        }
        if (full_b && kfull) {
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it) {
                if (TRANS_B) {
                    rb[it] = kmajor_fast(B, ldb, n0, k0, it);
                } else {
                    const int e = tid + it * 256;            // float4 index in the BK x (BN/4) tile
                    const int kr = e / (BN / 4), n4 = (e % (BN / 4)) * 4;
                    rb[it] = *reinterpret_cast<const float4*>(B + (long)(k0 + kr) * ldb + n0 + n4);
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it) {
                if (TRANS_B) {
                    rb[it] = kmajor_edge(B, ldb, n0, N, k0, it);
                } else {
                    const int e = tid + it * 256;
                    const int kr = e / (BN / 4), n4 = (e % (BN / 4)) * 4;
                    const int gk = k0 + kr, gn = n0 + n4, ne = N - 1;
                    const float* p = B + (long)min(gk, k_end - 1) * ldb;
                    const bool kok = gk < k_end;
                    float4 v;
                    v.x = p[min(gn, ne)];
                    v.y = p[min(gn + 1, ne)];
                    v.z = p[min(gn + 2, ne)];
                    v.w = p[min(gn + 3, ne)];
                    v.x = (kok && gn < N) ? v.x : 0.f;
                    v.y = (kok && gn + 1 < N) ? v.y : 0.f;
                    v.z = (kok && gn + 2 < N) ? v.z : 0.f;
                    v.w = (kok && gn + 3 < N) ? v.w : 0.f;
                    rb[it] = v;
                }
            }
        }
    };
    auto store_tiles = [&](float* As, float* Bs) {
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
            const int e = tid + it * 256;
            if (TRANS_A) {
                const int kr = e / (BM / 4), m4 = (e % (BM / 4)) * 4;
                *reinterpret_cast<float4*>(&As[kr * MPAD + m4]) = ra[it];
            } else {
                const int r = e / (BK / 4), k4 = (e % (BK / 4)) * 4;
                *reinterpret_cast<float4*>(&As[r * KPAD + k4]) = ra[it];
            }
        }
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int e = tid + it * 256;
            if (TRANS_B) {
                const int r = e / (BK / 4), k4 = (e % (BK / 4)) * 4;
                *reinterpret_cast<float4*>(&Bs[r * KPAD + k4]) = rb[it];
            } else {
                const int kr = e / (BN / 4), n4 = (e % (BN / 4)) * 4;
                *reinterpret_cast<float4*>(&Bs[kr * NPAD + n4]) = rb[it];
            }
        }
    };

    const int half = lane >> 5, l31 = lane & 31;
    const int nsteps = k_end > k_begin ? (k_end - k_begin + BK - 1) / BK : 0;
    if (nsteps > 0) {
        load_tiles(k_begin);
        store_tiles(smem, smem + A_FLOATS);
        if (nsteps > 1) load_tiles(k_begin + BK);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const float* As = smem + (s & 1) * STAGE;
        const float* Bs = As + A_FLOATS;
        float* An = smem + ((s + 1) & 1) * STAGE;
        if (s + 1 < nsteps) store_tiles(An, An + A_FLOATS);          // tile s+1: registers -> other stage
        if (s + 2 < nsteps) load_tiles(k_begin + (s + 2) * BK);       // tile s+2: HBM -> registers
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (TRANS_A) {
                    const float* ap = &As[(g * 8 + 4 * half) * MPAD + wm * WM + i * 32 + l31];
                    a[i] = make_float4(ap[0], ap[MPAD], ap[2 * MPAD], ap[3 * MPAD]);
                } else {
                    a[i] = *reinterpret_cast<const float4*>(&As[(wm * WM + i * 32 + l31) * KPAD + g * 8 + 4 * half]);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (TRANS_B) {
                    b[j] = *reinterpret_cast<const float4*>(&Bs[(wn * WN + j * 32 + l31) * KPAD + g * 8 + 4 * half]);
                } else {
                    const float* bp = &Bs[(g * 8 + 4 * half) * NPAD + wn * WN + j * 32 + l31];
                    b[j] = make_float4(bp[0], bp[NPAD], bp[2 * NPAD], bp[3 * NPAD]);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            t == 0 ? a[i].x : (t == 1 ? a[i].y : (t == 2 ? a[i].z : a[i].w)),
                            t == 0 ? b[j].x : (t == 1 ? b[j].y : (t == 2 ? b[j].z : b[j].w)), acc[i][j], 0, 0, 0);
        }
        __syncthreads();   // stage s fully read, stage s+1 fully written
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // (all row scales / biases are fetched up front with clamped indices: a load inside the per-element
    // branches would be waited for element by element)
    const bool first_split = blockIdx.z == 0;
    float rs[TM][16], bv[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            rs[i][r] = row_scale ? row_scale[min(gm, M - 1)] : 1.0f;
        }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gn = n0 + wn * WN + j * 32 + l31;
        bv[j] = (bias && first_split) ? bias[min(gn, N - 1)] : 0.0f;
    }
    float cs[TN], cq[TN];   // per-column sum / sum of squares over this wavefront's rows (InstanceNorm partials)
#pragma unroll
    for (int j = 0; j < TN; ++j) { cs[j] = 0.f; cq[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gn = n0 + wn * WN + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float v = acc[i][j][r] * rs[i][r] + bv[j];
                if (gm < M && gn < N) {
                    float* dst = C + (long)gm * ldc + gn;
                    if (atomic_out) atomicAdd(dst, v);
                    else *dst = v;
                    cs[j] += v;
                    cq[j] += v * v;
                }
            }
        }
    // Column statistics of the stored tile, one chunk per (block row, wavefront row): layout
    // [2][N][chunks] fp64, finished by pcrcg_instnorm_stats_from_partials (deterministic, no atomics).
    if (colp) {
        const int chunk = tile_y * WAVES_M + wm;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float s = cs[j] + __shfl_xor(cs[j], 32, 64), q2 = cq[j] + __shfl_xor(cq[j], 32, 64);
            const int gn = n0 + wn * WN + j * 32 + l31;
            if (half == 0 && gn < N) {
                colp[(long)gn * colp_chunks + chunk] = (double)s;
                colp[((long)N + gn) * colp_chunks + chunk] = (double)q2;
            }
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool TRANS_A, bool TRANS_B, int BK>
int launch_one(dim3 grid, hipStream_t st, const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m,
               int n, int k, const float* row_scale, const float* bias, int k_per_split, int vec_a, int vec_b,
               int atomic_out, double* colp, int colp_chunks) {
    constexpr size_t lds = 2 * sizeof(float) * stage_floats<BM, BN, TRANS_A, TRANS_B, BK>();
    auto kern = k_gemm_f32<BM, BN, WAVES_M, WAVES_N, TRANS_A, TRANS_B, BK>;
    PCRCG_GRANT_LDS(kern);   // > 64 KiB of dynamic LDS must be requested once per kernel
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias, k_per_split,
                       vec_a, vec_b, atomic_out, colp, colp_chunks);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int BK>
int launch(bool trans_b, dim3 grid, hipStream_t st, const float* a, int lda, const float* b, int ldb, float* c,
           int ldc, int m, int n, int k, const float* row_scale, const float* bias, int k_per_split, int vec_a,
           int vec_b, int atomic_out, double* colp, int colp_chunks) {
    if (trans_b)
        return launch_one<BM, BN, WAVES_M, WAVES_N, false, true, BK>(grid, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias,
                                                          k_per_split, vec_a, vec_b, atomic_out, colp, colp_chunks);
    return launch_one<BM, BN, WAVES_M, WAVES_N, false, false, BK>(grid, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias,
                                                       k_per_split, vec_a, vec_b, atomic_out, colp, colp_chunks);
}

}  // namespace
}  // namespace pcrcg

namespace pcrcg {
int gemm_x6_dispatch(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k,
                     const float* row_scale, const float* bias, void* colstats, size_t colstats_bytes, int* h_chunks,
                     hipStream_t st, bool a_bf16, bool c_zeroed, int a_kmajor = 0, int b_kmajor = 0,
                     bool colstats_sums = false, const GemmExtra* ex = nullptr, const GemmGroup* grp = nullptr);   // gemm_x6.hip
int gemm_x6_splits(int m, int n, int k, long m_total = 0);
}

using namespace pcrcg;

// Arithmetic of the C = A * B^T products: 0 = v_mfma_f32_32x32x2_f32 (fp32 operands), 1 = six
// v_mfma_f32_32x32x16_bf16 on the exact three-term bf16 split of the fp32 operands (gemm_x6.hip; fp32-class
// accuracy at 2.7x the matrix rate).  Default 1; PCRCG_GEMM_MODE / pcrcg_gemm_set_mode override.
// (One process-wide atomic word: read by every host thread that enqueues products, written by pcrcg_gemm_set_mode.)
static std::atomic<int> g_gemm_mode{-1};
static int gemm_mode() {
    int mode = g_gemm_mode.load(std::memory_order_relaxed);
    if (mode < 0) {
        const char* e = getenv("PCRCG_GEMM_MODE");
        int want = e ? (atoi(e) != 0) : 1, expect = -1;
        g_gemm_mode.compare_exchange_strong(expect, want, std::memory_order_relaxed);   // an explicit set_mode wins the race
        mode = g_gemm_mode.load(std::memory_order_relaxed);
    }
    return mode;
}
extern "C" void pcrcg_gemm_set_mode(int mode) { g_gemm_mode.store(mode != 0, std::memory_order_relaxed); }
extern "C" void pcrcg_thread_shares_gpu(int on) { gemm_x6_set_shared(on); }
namespace pcrcg { int gemm_x6_redo_counts(unsigned long long* out, int reset); }
extern "C" int pcrcg_gemm_redo_counts(unsigned long long* out2, int reset) { return gemm_x6_redo_counts(out2, reset); }
extern "C" int pcrcg_gemm_get_mode(void) { return gemm_mode(); }

static int gemm_dispatch(const float* a, int lda, int trans_a, const float* b, int ldb, int trans_b, float* c, int ldc,
                         int m, int n, int k, const float* row_scale, const float* bias, void* colstats,
                         size_t colstats_bytes, int* h_chunks, void* stream, bool c_zeroed = false,
                         bool colstats_sums = false);

namespace pcrcg {
// For the network runner (runner.hip): does a C = A * B^T product of this shape accumulate split-K partial sums into C
// (so that a C taken from the runner's pre-zeroed arena saves the product's own memset)?  Only the default arithmetic.
// (m: rows of the largest product of a grouped launch, m_total: of all of them -- the plan looks at both)
bool gemm_bt_accumulates(int m, int n, int k, long m_total) { return gemm_mode() == 1 && gemm_x6_splits(m, n, k, m_total) > 1; }
// pcrcg_gemm_f32_colstats (trans_b = 1) / pcrcg_gemm_bf16a_f32_colstats with the promise that C is all zeros.
// colstats_sums: `colstats` is a ZEROED [2][n] fp64 accumulator; when the product writes every element once, its epilogue
// adds the column sums / sums of squares there with atomics and reports *h_chunks = -1 (else 0: nothing was added).
bool gemm_colstats_sums_ok() { return gemm_mode() == 1; }
// C (+)= A[gathered rows] * B^T for k-contiguous fp32 operands (GemmExtra, common.h); split-bf16 arithmetic only
bool gemm_extra_ok() { return gemm_mode() == 1; }
int gemm_bt_extra(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k, hipStream_t st,
                  bool c_zeroed, const GemmExtra& ex, const float* bias, void* colstats, size_t colstats_bytes, int* h_chunks,
                  bool colstats_sums) {
    if (h_chunks) *h_chunks = 0;
    PCRCG_CHECK_ARG(m >= 0 && n >= 0 && k >= 1 && lda >= k && ldb >= k && ldc >= n);
    if (m == 0 || n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(a && b && c && gemm_mode() == 1);
    return gemm_x6_dispatch(a, lda, b, ldb, c, ldc, m, n, k, nullptr, bias, colstats, colstats_bytes, h_chunks, st, false,
                            c_zeroed, 0, 0, colstats_sums, &ex);
}
// The two forms above with an optional SECOND product in the same launch (GemmPair, common.h); pair == NULL: as above.
bool gemm_pair_ok() { return gemm_mode() == 1; }
int gemm_bt_colstats_pair(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k,
                          const float* row_scale, const float* bias, void* colstats, size_t colstats_bytes, int* h_chunks,
                          hipStream_t st, bool c_zeroed, bool colstats_sums, const GemmGroup* pair) {
    if (!pair || pair->n == 0)
        return gemm_dispatch(a, lda, 0, b, ldb, 1, c, ldc, m, n, k, row_scale, bias, colstats, colstats_bytes, h_chunks, st,
                             c_zeroed, colstats_sums);
    if (h_chunks) *h_chunks = 0;
    PCRCG_CHECK_ARG(m >= 1 && n >= 1 && k >= 1 && lda >= k && ldb >= k && ldc >= n);
    PCRCG_CHECK_ARG(a && b && c && gemm_mode() == 1);
    return gemm_x6_dispatch(a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias, colstats, colstats_bytes, h_chunks, st, false,
                            c_zeroed, 0, 0, colstats_sums, nullptr, pair);
}
int gemm_bt_extra_pair(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k, hipStream_t st,
                       bool c_zeroed, const GemmExtra& ex, const float* bias, void* colstats, size_t colstats_bytes,
                       int* h_chunks, bool colstats_sums, const GemmGroup* pair) {
    if (h_chunks) *h_chunks = 0;
    PCRCG_CHECK_ARG(m >= 0 && n >= 0 && k >= 1 && lda >= k && ldb >= k && ldc >= n);
    if (m == 0 || n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(a && b && c && gemm_mode() == 1);
    return gemm_x6_dispatch(a, lda, b, ldb, c, ldc, m, n, k, nullptr, bias, colstats, colstats_bytes, h_chunks, st, false,
                            c_zeroed, 0, 0, colstats_sums, &ex, pair);
}
// C (+)= (op(A) * op(B)) * row_scale[m] + bias[n] for the training runner (train_runner.hip): op = identity or transpose
// as in pcrcg_gemm_f32_ex; accumulate adds the product onto C with fp32 atomics (gradients of tensors with several
// consumers, parameter gradients of shared weights).  Split-bf16 arithmetic only.
int gemm_general(const float* a, int lda, int trans_a, const float* b, int ldb, int trans_b, float* c, int ldc, int m, int n,
                 int k, const float* row_scale, const float* bias, bool accumulate, hipStream_t st, int grad_operand) {
    PCRCG_CHECK_ARG(m >= 0 && n >= 0 && k >= 0);
    if (m == 0 || n == 0 || k == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(a && b && c && !(trans_a && trans_b));
    PCRCG_CHECK_ARG((trans_a ? lda >= m : lda >= k) && ldc >= n && (trans_b ? ldb >= k : ldb >= n));
    if (gemm_mode() != 1) {
        set_error("gemm_general: the training runner needs the split-bf16 arithmetic (pcrcg_gemm_set_mode(1))");
        return PCRCG_EBADARG;
    }
    GemmExtra ex;
    ex.accumulate = accumulate;
    ex.grad_operand = grad_operand;
    return gemm_x6_dispatch(a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias, nullptr, 0, nullptr, st, false, false,
                            trans_a ? 1 : 0, trans_b ? 0 : 1, false, (accumulate || grad_operand) ? &ex : nullptr);
}
int gemm_bt_colstats(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k,
                     const float* row_scale, const float* bias, void* colstats, size_t colstats_bytes, int* h_chunks,
                     hipStream_t st, bool c_zeroed, bool colstats_sums) {
    return gemm_dispatch(a, lda, 0, b, ldb, 1, c, ldc, m, n, k, row_scale, bias, colstats, colstats_bytes, h_chunks, st,
                         c_zeroed, colstats_sums);
}
int gemm_bf16a_bt_colstats(const void* a_bf16, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k,
                           const float* row_scale, const float* bias, void* colstats, size_t colstats_bytes,
                           int* h_chunks, hipStream_t st, bool c_zeroed, bool colstats_sums) {
    if (h_chunks) *h_chunks = 0;
    PCRCG_CHECK_ARG(m >= 0 && n >= 0 && k >= 32 && k % 32 == 0 && lda >= k && lda % 8 == 0 && ldb >= k && ldc >= n);
    if (m == 0 || n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(a_bf16 && b && c && (reinterpret_cast<uintptr_t>(a_bf16) & 15) == 0);
    // the bf16-A kernel has no guarded A loads: every k-step must take the vector path, which also needs B aligned
    PCRCG_CHECK_ARG(ldb % 4 == 0 && (reinterpret_cast<uintptr_t>(b) & 15) == 0);
    return gemm_x6_dispatch(static_cast<const float*>(a_bf16), lda, b, ldb, c, ldc, m, n, k, row_scale, bias, colstats,
                            colstats_bytes, h_chunks, st, true, c_zeroed, 0, 0, colstats_sums);
}
}  // namespace pcrcg

extern "C" size_t pcrcg_gemm_colstats_bytes(int m, int n) {
    const size_t chunks = (size_t)((m > 0 ? m : 1) + 31) / 32 + 4;
    return carve_bytes(2 * (size_t)(n > 0 ? n : 1) * chunks, sizeof(double));
}

extern "C" int pcrcg_gemm_f32(const float* a, int lda, const float* b, int ldb, int trans_b, float* c, int ldc,
                              int m, int n, int k, const float* row_scale, const float* bias, void* stream) {
    return pcrcg_gemm_f32_colstats(a, lda, b, ldb, trans_b, c, ldc, m, n, k, row_scale, bias, nullptr, 0, nullptr,
                                   stream);
}

extern "C" int pcrcg_gemm_f32_colstats(const float* a, int lda, const float* b, int ldb, int trans_b, float* c,
                                       int ldc, int m, int n, int k, const float* row_scale, const float* bias,
                                       void* colstats, size_t colstats_bytes, int* h_chunks, void* stream) {
    return gemm_dispatch(a, lda, 0, b, ldb, trans_b, c, ldc, m, n, k, row_scale, bias, colstats, colstats_bytes, h_chunks,
                         stream);
}

// C = (A @ B^T) * row_scale + bias with A stored as bf16 ([m, k], lda in bf16 elements; 16-byte aligned rows, k % 32 == 0)
// and B fp32 [n, k]: the bf16 feature-storage variant's contraction (gemm_x6.hip, ATERMS = 1).
extern "C" int pcrcg_gemm_bf16a_f32_colstats(const void* a_bf16, int lda, const float* b, int ldb, float* c, int ldc, int m,
                                             int n, int k, const float* row_scale, const float* bias, void* colstats,
                                             size_t colstats_bytes, int* h_chunks, void* stream) {
    return gemm_bf16a_bt_colstats(a_bf16, lda, b, ldb, c, ldc, m, n, k, row_scale, bias, colstats, colstats_bytes, h_chunks,
                                  as_stream(stream), false, false);
}

// C (+)= f(A)[rows] * B^T + bias: the products of the runner that fold a neighbouring operator into their A loads.
//  * idx != NULL: output row r uses row idx[r * ld_idx] of A ([ns, k], lda) -- the zero row `zero_row` (>= k floats of 0)
//    when that index is outside [0, ns), the shadow neighbour (nearest_upsample, ref:models/blocks.py:77-87);
//  * a_sums != NULL: A is the raw output of a product whose InstanceNorm + LeakyReLU is applied on load,
//    f(a) = lrelu((a - mean_k) * rstd_k, a_slope), from the fp64 column sums a_sums [2][k] over a_count rows
//    (ref:models/blocks.py:456-470); a gathered shadow row stays zero;
//  * accumulate != 0 adds the product to C instead of storing it (cat(skip) as a second product).
extern "C" int pcrcg_gemm_f32_fused(const float* a, int lda, const int64_t* idx, int ld_idx, int ns, const float* zero_row,
                                    const void* a_sums, double a_count, float a_eps, float a_slope, const float* b, int ldb,
                                    const float* bias, float* c, int ldc, int m, int n, int k, int accumulate, void* stream) {
    PCRCG_CHECK_ARG(!idx || (ld_idx >= 1 && ns >= 0 && zero_row));
    PCRCG_CHECK_ARG(!a_sums || a_count >= 1.0);
    if (gemm_mode() != 1) {
        set_error("pcrcg_gemm_f32_fused: only the split-bf16 arithmetic (pcrcg_gemm_set_mode(1)) implements it");
        return PCRCG_EBADARG;
    }
    GemmExtra ex;
    ex.a_idx = reinterpret_cast<const long long*>(idx);
    ex.a_idx_ld = ld_idx;
    ex.a_ns = ns;
    ex.a_zero = zero_row;
    ex.accumulate = accumulate != 0;
    ex.a_sums = static_cast<const double*>(a_sums);
    ex.a_count = a_count;
    ex.a_eps = a_eps;
    ex.a_slope = a_slope;
    return gemm_bt_extra(a, lda, b, ldb, c, ldc, m, n, k, as_stream(stream), false, ex, bias, nullptr, 0, nullptr, false);
}

// Aop = A^T when trans_a (A stored [K, M] row-major): the weight-gradient products dW = X^T * dY of the
// training rows (include/pcrcg_train.h), whose reduction dimension is the number of points.
extern "C" int pcrcg_gemm_f32_ex(const float* a, int lda, int trans_a, const float* b, int ldb, int trans_b, float* c,
                                 int ldc, int m, int n, int k, const float* row_scale, const float* bias, void* stream) {
    return gemm_dispatch(a, lda, trans_a, b, ldb, trans_b, c, ldc, m, n, k, row_scale, bias, nullptr, 0, nullptr, stream);
}

extern "C" int pcrcg_gemm_f32_grad(const float* a, int lda, int trans_a, const float* b, int ldb, int trans_b, float* c,
                                   int ldc, int m, int n, int k, const float* row_scale, const float* bias, int grad_operand,
                                   void* stream) {
    PCRCG_CHECK_ARG(grad_operand >= 0 && grad_operand <= 2);
    if (grad_operand == 0 || pcrcg_gemm_get_mode() != 1 || (trans_a && trans_b))
        return gemm_dispatch(a, lda, trans_a, b, ldb, trans_b, c, ldc, m, n, k, row_scale, bias, nullptr, 0, nullptr, stream);
    return pcrcg::gemm_general(a, lda, trans_a, b, ldb, trans_b, c, ldc, m, n, k, row_scale, bias, false, as_stream(stream),
                               grad_operand);
}

static int gemm_dispatch(const float* a, int lda, int trans_a, const float* b, int ldb, int trans_b, float* c, int ldc,
                         int m, int n, int k, const float* row_scale, const float* bias, void* colstats,
                         size_t colstats_bytes, int* h_chunks, void* stream, bool c_zeroed, bool colstats_sums) {
    if (h_chunks) *h_chunks = 0;
    PCRCG_CHECK_ARG(m >= 0 && n >= 0 && k >= 0);
    if (m == 0 || n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(a && b && c);
    PCRCG_CHECK_ARG((trans_a ? lda >= m : lda >= k) && ldc >= n);
    PCRCG_CHECK_ARG(trans_b ? ldb >= k : ldb >= n);
    hipStream_t st = as_stream(stream);
    // split-bf16 arithmetic for A * B^T (the forward), A * B (dX = dY * W) and A^T * B (dW = X^T * dY); A^T * B^T stays fp32
    if (gemm_mode() == 1 && !(trans_a && trans_b))
        return gemm_x6_dispatch(a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias, colstats, colstats_bytes, h_chunks, st, false,
                                c_zeroed, trans_a ? 1 : 0, trans_b ? 0 : 1, colstats_sums);
    if (colstats_sums) { colstats = nullptr; colstats_bytes = 0; }     // (the fp32 kernels only know the partials layout)
    const int vec_a = (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0);
    const int vec_b = (ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(b) & 15) == 0);
    // Tile / split selection (sweep in scripts/gemm_tune.py on the path's shapes): these GEMMs are skinny
    // (N = 64..2048, K up to 7680) and each block streams its own slice of A from HBM, so many small
    // blocks beat few large ones: 128x128 only when that still yields >= 1024 blocks, otherwise 64x64
    // (128x64 for N <= 64 with a long M), with K split until ~1024 blocks are in flight.
    struct Tile { int bm, bn; };
    static const Tile tiles[4] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
    auto ntiles = [&](int t) {
        return (long)((m + tiles[t].bm - 1) / tiles[t].bm) * ((n + tiles[t].bn - 1) / tiles[t].bn);
    };
    int pick = -1;
    pick = debug_opts().gemm_tile;                                           // tuning aid (-1: automatic)
    if (pick < 0 || pick > 3) pick = (n > 64 && ntiles(0) >= 1024) ? 0 : 3;
    if (trans_a) pick = 3;                                                   // only the 64x64 tile is built for A^T
    const int BM = tiles[pick].bm, BN = tiles[pick].bn;
    const int gx = (n + BN - 1) / BN, gy = (m + BM - 1) / BM;
    constexpr int BK = 32;   // a 64-deep k-step (2 blocks/CU) measured 12 % slower on the path's shapes
    int splits = 1;
    const int ktiles = (k + BK - 1) / BK;
    const int split_target = debug_opts().gemm_split_target;
    const int max_splits = trans_a ? 256 : 32;   // A^T products reduce over the points: few tiles, very long K
    while ((long)gx * gy * splits < split_target && k / (2 * splits) >= 192 && splits < max_splits) splits *= 2;
    if (debug_opts().gemm_splitk > 0) splits = debug_opts().gemm_splitk;     // tuning aid
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
    const bool log_shapes = debug_opts().gemm_log != 0;   // tuning aid
    if (log_shapes)
        fprintf(stderr, "pcrcg_gemm m=%d n=%d k=%d lda=%d ldb=%d ldc=%d tb=%d grid=%dx%dx%d rs=%d bias=%d stats=%d\n", m, n,
                k, lda, ldb, ldc, trans_b, gx, gy, splits, row_scale != nullptr, bias != nullptr, colstats != nullptr);
    // column statistics ride along only when every output element is written exactly once
    double* colp = nullptr;
    int colp_chunks = 0;
    if (colstats && h_chunks && !atomic_out) {
        colp_chunks = gy * (pick == 1 ? 4 : 2);   // WAVES_M of the chosen tile
        if (carve_bytes(2 * (size_t)n * colp_chunks, sizeof(double)) <= colstats_bytes) {
            colp = static_cast<double*>(colstats);
            *h_chunks = colp_chunks;
        } else {
            colp_chunks = 0;
        }
    }
#define GO(BMV, BNV, WMV, WNV)                                                                                   \
    return launch<BMV, BNV, WMV, WNV, BK>(trans_b != 0, grid, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias, \
                                      k_per_split, vec_a, vec_b, atomic_out, colp, colp_chunks)
    if (trans_a) {
        if (trans_b)
            return launch_one<64, 64, 2, 2, true, true, BK>(grid, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias,
                                                            k_per_split, vec_a, vec_b, atomic_out, colp, colp_chunks);
        return launch_one<64, 64, 2, 2, true, false, BK>(grid, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias,
                                                         k_per_split, vec_a, vec_b, atomic_out, colp, colp_chunks);
    }
    if (pick == 0) { GO(128, 128, 2, 2); }
    if (pick == 1) { GO(128, 64, 4, 1); }
    if (pick == 2) { GO(64, 128, 2, 2); }
    GO(64, 64, 2, 2);
#undef GO
}
