// gemm_x6.hip -- fp32-accurate GEMM on the gfx950 bf16 matrix cores:  C = (A * B^T) * row_scale[m] + bias[n]
// with A [M,K] and B [N,K] both row-major fp32 (k-contiguous: nn.Linear / 1x1 conv weights, K-contiguous
// copies of the KPConv weights).  Same contract and epilogue as k_gemm_f32 (gemm.hip); replaces the same
// reference lines (ref:models/blocks.py:361-372, :487; ref:models/architectures.py:528,538-539;
// ref:models/gcn.py:123-132,165-173).
//
// Arithmetic.  v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (157 TF), 1/16 of the bf16 matrix rate.
// Every fp32 value is EXACTLY the sum of three bf16 values (24 significand bits = 3 x 8):
//     x = x1 + x2 + x3,   x1 = trunc_bf16(x), x2 = trunc_bf16(x - x1), x3 = x - x1 - x2   (all exact)
// so  a*b = a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1) + [a2b3 + a3b2 + a3b3],  and the bracket is below
// 2^-23 |ab| -- the size of ONE fp32 rounding.  bf16 x bf16 products are exact in the matrix core and are
// accumulated in fp32, so six v_mfma_f32_32x32x16_bf16 per 16-deep k-chunk give an fp32-class result
// (measured on the CPU restatement, scripts/exp_splitbf16.py: the full-width model's outputs are as close to
// a float64 run as the plain fp32 model's are, 9e-7 vs 1.3e-6) at 16/6 = 2.7x the fp32 matrix rate.
// The three-term split is the point: the two-term variant (3 MFMAs) leaves 2^-16 per product and moved
// intermediate activations by 1.7e-4 in the same experiment -- outside this path's 1e-4 bar.
//
// Since round 4 the forward products (k-contiguous fp32 operands) run the fp16 TWO-term form instead -- three matrix
// instructions per chunk, two LDS planes, the same loop structure; see split2h and the H2 kernels below -- and fall back to
// the three-term form above, inside the kernel, for any tile whose operands leave fp16's normal range at EITHER end
// (round 5: rows whose values are all below 2^-14 as well as values beyond 65504).  The training rows' k-major products
// without a named gradient operand and bf16-stored operands use the three-term form directly.
//
// Structure: 256 threads = 4 wavefronts (2 x 2), block tile BM x BN x 32.  Operand tiles are loaded as
// float4 pairs (8 consecutive k of one row per thread and pass), split on the VALU and written to LDS as
// three bf16 planes with 80-byte rows (64 B of data + 16 B pad: the ds_read_b128 of 8 consecutive k for the
// 32 rows of an MFMA operand is bank-conflict free, 20*r mod 64 enumerates the sixteen 4-bank slots).  One
// LDS stage; the next tile's global loads are in flight while the current one is multiplied (register
// prefetch), and two or more co-resident blocks per CU overlap one block's split/write phase with the
// others' MFMA phase.
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>

#include <hip/hip_ext.h>

#include "common.h"
#include "pcrcg_train.h"

namespace pcrcg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32;

// Diagnostics of the fp16 form's two range checks: tiles that took the bf16 redo because of a value beyond fp16's range [0]
// or because of a row below it [1] (one atomic per such tile; read and reset through pcrcg_gemm_redo_counts)
__device__ unsigned long long g_x6_redo[2];

// kernel-side form of GemmGroup (common.h): the operands of up to 3 further products; extra = 0: a single product.
// Product e's row tiles start at row off[e] of the launch's tile space.
constexpr int kGroupExtra = 3;
struct GemmPairArgs {
    int extra = 0;
    int off[kGroupExtra] = {0, 0, 0}, m[kGroupExtra] = {0, 0, 0};
    const float* a[kGroupExtra] = {nullptr, nullptr, nullptr};
    float* c[kGroupExtra] = {nullptr, nullptr, nullptr};
    const float* rs[kGroupExtra] = {nullptr, nullptr, nullptr};
    double* colp[kGroupExtra] = {nullptr, nullptr, nullptr};
    int colp_chunks[kGroupExtra] = {0, 0, 0};
    const long long* a_idx[kGroupExtra] = {nullptr, nullptr, nullptr};
    int a_ns[kGroupExtra] = {0, 0, 0};
    const double* a_sums[kGroupExtra] = {nullptr, nullptr, nullptr};
    double a_count[kGroupExtra] = {0.0, 0.0, 0.0};
};

// Register loads hidden from hipcc's s_waitcnt bookkeeping (guide 5.7, form ii).  hipcc merges the vmcnt state of
// the two register sets of the k-loop conservatively and waits for BOTH at the top of every step (ISA checked),
// i.e. it turns prefetch distance 2 into 1; with the loads in asm the waits below are counted by hand instead:
// vm_wait<N>() leaves the N newest loads in flight, pin() makes every later use of a destination register
// depend on a statement that follows the wait.
__device__ __forceinline__ void gload16(f32x4& dst, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pin(f32x4& r) { asm volatile("" : "+v"(r)); }
__device__ __forceinline__ void gload4(float& dst, const float* p) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void pin(float& r) { asm volatile("" : "+v"(r)); }

// Operand layouts.  LAY = 0: element (row, k) at P[row * ld + k] (k-contiguous: activations, nn.Linear weights, the
// K-contiguous KPConv weights).  LAY = 1: element (row, k) at P[k * ld + row] (k-major: the operands of the training
// rows' products dX = dY * W and dW = X^T * dY, include/pcrcg_train.h).  An Item is 8 consecutive k of one tile row as
// the loads deliver it: two 16-byte loads for LAY 0, eight 4-byte loads (consecutive lanes = consecutive rows, 256
// contiguous bytes per instruction) for LAY 1.  Item e of a tile with ROWS rows: LAY 0 -> row e / 4, k-group e % 4;
// LAY 1 -> row e % ROWS, k-group e / ROWS; both land at LDS row * ROWB + k-group * 16 (conflict-free either way).
template <int LAY> struct Item;
template <> struct Item<0> {
    f32x4 v[2];
    static constexpr int LOADS = 2;
    __device__ __forceinline__ float get(int i) const { return v[i >> 2][i & 3]; }
    __device__ __forceinline__ void set(int i, float x) { v[i >> 2][i & 3] = x; }
    __device__ __forceinline__ void hold() { pin(v[0]); pin(v[1]); }
};
template <> struct Item<1> {
    float s[8];
    static constexpr int LOADS = 8;
    __device__ __forceinline__ float get(int i) const { return s[i]; }
    __device__ __forceinline__ void set(int i, float x) { s[i] = x; }
    __device__ __forceinline__ void hold() {
#pragma unroll
        for (int i = 0; i < 8; ++i) pin(s[i]);
    }
};
template <int LAY, int ROWS>
__device__ __forceinline__ void item_pos(int e, int& row, int& kg) {
    if (LAY == 0) { row = e >> 2; kg = e & 3; }
    else { row = e % ROWS; kg = e / ROWS; }
}
// unguarded loads of a full 32-deep k-slab; rows past the matrix are CLAMPED to its last row: they only feed output
// rows / columns the epilogue masks, so no zero fill is needed along M or N (only along K, in the guarded form)
template <int LAY, int ROWS>
__device__ __forceinline__ void item_load_fast(Item<LAY>& d, const float* __restrict__ P, int ld, int row0, int rows, int k0,
                                               int e) {
    int row, kg;
    item_pos<LAY, ROWS>(e, row, kg);
    if constexpr (LAY == 0) {
        const float* p = P + (long)min(row0 + row, rows - 1) * ld + k0 + kg * 8;
        gload16(d.v[0], p);
        gload16(d.v[1], p + 4);
    } else {
        const float* p = P + (long)(k0 + kg * 8) * ld + min(row0 + row, rows - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) gload4(d.s[j], p + (long)j * ld);
    }
}
template <int LAY, int ROWS>
__device__ __forceinline__ void item_load_edge(Item<LAY>& d, const float* __restrict__ P, int ld, int row0, int rows, int k0,
                                               int k_end, int e) {
    int row, kg;
    item_pos<LAY, ROWS>(e, row, kg);
    const int gr = row0 + row, gk = k0 + kg * 8, ke = k_end - 1;
    const bool rok = gr < rows;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)                                  // always a valid address
        v[j] = LAY == 0 ? P[(long)min(gr, rows - 1) * ld + min(gk + j, ke)] : P[(long)min(gk + j, ke) * ld + min(gr, rows - 1)];
#pragma unroll
    for (int j = 0; j < 8; ++j) d.set(j, (rok && gk + j < k_end) ? v[j] : 0.f);
}

constexpr int ROWB = 80;   // bytes per LDS row of one plane: 32 bf16 + 16 B pad

// two fp32 -> their three bf16 terms, packed pairwise (low half = first element)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    const float h0 = __uint_as_float(u0 & 0xffff0000u), h1 = __uint_as_float(u1 & 0xffff0000u);
    const float r0 = x0 - h0, r1 = x1 - h1;                                   // exact
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    const float m0 = __uint_as_float(v0 & 0xffff0000u), m1 = __uint_as_float(v1 & 0xffff0000u);
    const float l0 = r0 - m0, l1 = r1 - m1;                                   // exact, <= 8 significant bits
    p1 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);                          // {hi16(x1), hi16(x0)}
    p2 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    p3 = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
}

// The fp16 TWO-term form (H2 kernels): x = h + 2^-11 l with h = fp16_rn(x), l = fp16_rn((x - h) * 2^11) -- x - h is exact
// and at most half an ulp of h, so the scaled remainder is no larger than x and l loses nothing to the fp16 subnormal
// range that h did not.  |x - h - 2^-11 l| <= 2^-22 |x| (2^-36 absolute below 2^-14), and
//     a b = ha hb + 2^-11 (ha lb + la hb) + [2^-22 la lb]
// with the bracket and the representation error both at 2^-22 |ab|: three v_mfma_f32_32x32x16_f16 per 16-deep k-chunk
// (fp16 x fp16 products are exact in the matrix core's fp32 accumulation, fp16 subnormals are kept: scripts/micro/
// mfma_f16_denormal.hip) instead of six bf16 ones, two LDS planes instead of three, 4 instead of 11 VALU operations per
// operand pair.  Measured on the path's shapes: 4.5-5.8e-7 of a float64 product, a plain fp32 GEMM's error (the
// three-term bf16 form: 2.4e-7).  What the form cannot do is hold values outside fp16's NORMAL range, at either end:
// |x| >= 65520 becomes +-inf and leaves a non-finite partial sum behind; below 2^-14 h is a subnormal (below 2^-25: zero)
// and the split's error is an absolute 2^-36 instead of a relative 2^-22 -- harmless beside larger values of the same row,
// fatal for a row whose values are ALL that small.  Both are caught per tile after the loop (see the checks there), and a
// workgroup that finds either throws its sums away and runs its tile again with the three-term bf16 loop, which has fp32's
// range -- no flag for the host, no different result contract.
constexpr float kH2Scale = 2048.0f;          // 2^11
// Four VALU instructions per operand pair (round 5; the compiler's own code for the same arithmetic takes six): h by
// v_cvt_pk_f16_f32, y = 2^11 x by one packed multiply, and l = fp16_rn(y - 2^11 h) by the mixed-precision FMAs, which read
// h's halves as fp16 operands and round their (exact) fp32 result straight into the two halves of l.
// scripts/micro/split_mix.hip checks the pair bit for bit against the plain C++ form over random, subnormal and special values.
__device__ __forceinline__ void split2h(float x0, float x1, unsigned& p1, unsigned& p2) {
    const f32x2 x = {x0, x1};
    const f16x2 h = __builtin_convertvector(x, f16x2);                        // v_cvt_pk_f16_f32, round to nearest even
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    const f32x2 y = x * kH2Scale;                                             // exact
    const float m = -kH2Scale;
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hb), "s"(m), "v"(y[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hb), "s"(m), "v"(y[1]));
    p1 = hb;
    p2 = l;
}

template <int BM, int BN>
constexpr int lds_bytes() { return 3 * (BM + BN) * ROWB; }
template <int BM, int BN>
constexpr int x6_threads() { return (BM == 128 && BN == 128) ? 512 : 256; }

// How a launch's tiles are laid over the eight XCDs (see the kernel).  Each XCD takes T / 8 consecutive tiles of the
// order (split, row tile, column tile) [0] or (split, column tile, row tile) [1]; the estimate below counts what the
// eight private L2s fetch between them under either -- an A block (row tile x split) and a B block (column tile x split)
// once per XCD that touches it, again per pass when the blocks revisited between passes exceed what an L2 keeps -- and
// picks the cheaper.  Tall products (many row tiles per XCD) keep order 0: A streams once, B's few blocks stay cached.
// Short, wide ones (the coarse levels: 381 .. 763 rows against 512 .. 2048 output columns) take order 1: every XCD then
// owns a few column tiles of the weights instead of reading all of them (measured before: fetch = A + 8 B).
struct TileMap { int gx, gy, gs, order; long split_stride; };   // split_stride != 0: split s writes its partial tile to C + s * split_stride (no atomics)
static int x6_tile_order(int gx, int gy, int gs, int bm, int bn, int k_per_split) {
    const int forced = debug_opts().x6_order;
    if (forced >= 0 && forced <= 2) return forced;
    const double a_blk = 4.0 * bm * k_per_split, b_blk = 4.0 * bn * k_per_split, keep = 2.0e6;   // bytes; L2 = 4 MB per XCD
    const double total = (double)gx * gy * gs, q = total / 8.0;                                     // tiles per XCD
    const double per_split = (double)gx * gy;
    const double splits_touched = q >= per_split ? q / per_split : 1.0;
    const double in_split = q >= per_split ? per_split : q;                                         // tiles per XCD inside one split
    // order 0: in_split tiles = rows of gx column tiles
    double rows0 = in_split / gx; if (rows0 < 1.0) rows0 = 1.0;
    const double cols0 = in_split < gx ? in_split : gx;
    const double b0 = cols0 * b_blk, a0 = rows0 * a_blk;
    const double cost0 = splits_touched * (a0 + (b0 <= keep ? b0 : b0 * rows0));
    // order 1: in_split tiles = columns of gy row tiles
    double cols1 = in_split / gy; if (cols1 < 1.0) cols1 = 1.0;
    const double rows1 = in_split < gy ? in_split : gy;
    const double a1 = rows1 * a_blk, b1 = cols1 * b_blk;
    const double cost1 = splits_touched * (b1 + (a1 <= keep ? a1 : a1 * cols1));
    return cost1 < 0.9 * cost0 ? 1 : 0;
}

// ATERMS = 3: A is fp32 and is split like B.  ATERMS = 1: A already IS bf16 in memory (the bf16 feature-storage
// variant's wf, lda in bf16 elements): its tile is copied straight into plane 0 and only the three products a1*b3,
// a1*b2, a1*b1 run -- exact in B, bf16-rounded in A by the storage format, fp32 accumulate.  K % 32 == 0 required.
// KNOCK (always 0 in the library; scripts/micro/x6_knock.hip instantiates other values to time the kernel with one of
// its parts removed -- results wrong, timing meaningful): 1 no global loads inside the k-loop, 2 no split arithmetic,
// 4 no LDS stores, 8 no MFMA, 16 no LDS operand reads, 32 no barriers.
template <int BM, int BN, int MINB, int ATERMS, int ALAY, int BLAY, int ANORM = 0, int KNOCK = 0, int H2 = 0>
__global__ void __launch_bounds__((x6_threads<BM, BN>()), MINB) k_gemm_x6(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                        int ldb, float* __restrict__ C, int ldc, int M, int N, int Kdim,
                                                        const float* __restrict__ row_scale,
                                                        const float* __restrict__ bias, int k_per_split, int vec_a,
                                                        int vec_b, int atomic_out, double* __restrict__ colp,
                                                        int colp_chunks, const long long* __restrict__ a_idx,
                                                        int a_idx_ld, int a_ns, const float* __restrict__ a_zero,
                                                        const double* __restrict__ a_sums, double a_count, float a_eps,
                                                        float a_slope, GemmPairArgs pr, TileMap tm, float h2_sa, float h2_sb) {
    // 256 threads = 2 x 2 wavefronts; the 128 x 128 tile runs 512 threads = 2 x 4 wavefronts (64 x 32 each): the same
    // registers per thread and wavefronts per CU as the 64 x 64 tile at half its L1 fills per flop
    constexpr int NT = x6_threads<BM, BN>();
    constexpr int WAVES_M = 2, WAVES_N = NT / 128;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_ITERS = BM * 4 / NT;    // (row, 8-k group) items per thread
    constexpr int B_ITERS = BN * 4 / NT;
    constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
    static_assert(A_ITERS >= 1 && B_ITERS >= 1, "tile too small");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const As = smem;
    unsigned char* const Bs = smem + 3 * A_PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // XCD-aware tile order over the WHOLE launch (1-D grid of gx * gy * gs workgroups; the dispatcher hands workgroup L
    // to XCD L % 8): each XCD walks a contiguous range of the (split, row tile, column tile) space, in the order
    // x6_tile_order() picked -- so that the operand blocks an XCD touches are few and their re-use happens inside its own
    // L2 (the eight L2s are private: what two XCDs both read is fetched twice)
    const int gx = tm.gx;
    int ntile = tm.gx * tm.gy * tm.gs, lin = blockIdx.x, base = 0;
    if (tm.order == 2) {            // measurement aid (x6_order=2): the round-2 map, every split's tiles spread over all XCDs
        ntile = tm.gx * tm.gy;
        base = lin / ntile * ntile;
        lin -= base;
    }
    const int xq = ntile >> 3, xr = ntile & 7, xcd = lin & 7, slot = lin >> 3;
    const int tile = base + (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + slot;
    int tile_x, tile_y;
    const int split = tile / (tm.gx * tm.gy), in_split = tile % (tm.gx * tm.gy);
    if (tm.order != 1) {            // split, row tile, column tile (fastest)
        tile_x = in_split % gx;
        tile_y = in_split / gx;
    } else {                        // split, column tile, row tile (fastest)
        tile_y = in_split % tm.gy;
        tile_x = in_split / tm.gy;
    }
    int m0 = tile_y * BM;
    const int n0 = tile_x * BN;
    // Several products that share B in ONE launch (GemmGroup, common.h: the same layer of several fragment pairs): row
    // tiles from off[e] on belong to product e + 1 -- its own A, C, row scale, statistics and gather table; from here on
    // the workgroup works on that product's matrices as if it had been launched alone.
    if (pr.extra > 0 && m0 >= pr.off[0]) {
        int e = 0;
        if (pr.extra > 1 && m0 >= pr.off[1]) e = 1;
        if (pr.extra > 2 && m0 >= pr.off[2]) e = 2;
        m0 -= pr.off[e];
        tile_y = m0 / BM;
        M = pr.m[e];
        A = pr.a[e];
        C = pr.c[e];
        row_scale = pr.rs[e];
        colp = pr.colp[e];
        colp_chunks = pr.colp_chunks[e];
        a_idx = pr.a_idx[e];
        a_ns = pr.a_ns[e];
        a_sums = pr.a_sums[e];
        a_count = pr.a_count[e];
    }
    const int k_begin = split * k_per_split;
    const int k_end = min(Kdim, k_begin + k_per_split);
    C += (long)split * tm.split_stride;          // the deterministic mode's two-pass reduction (gemm_x6_dispatch)

    f32x16 acc[TM][TN];
    f32x16 acc_lo[H2 ? TM : 1][H2 ? TN : 1];       // H2: the cross terms ha lb + la hb (worth 2^-11 of acc's units)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.0f;
                if constexpr (H2) acc_lo[i][j][r] = 0.0f;
            }

    // k-contiguous fp32 A: the row of every A item of this thread is the same in all k-steps, so its address is formed
    // once -- and may come through a gather: a_idx != NULL reads row a_idx[r * a_idx_ld] of A for output row r, a zero
    // row (a_zero, >= K floats) when that index is not in [0, a_ns) (nearest-neighbour upsampling with its shadow index
    // folded into the product: ref:models/blocks.py:77-87 closest_pool followed by the decoder's unary block)
    const float* arow[A_ITERS];
    bool arok[A_ITERS], azero[A_ITERS];
    if constexpr (ALAY == 0 && ATERMS == 3) {
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
            const int gr = m0 + ((tid + it * NT) >> 2);
            const int r = min(gr, M - 1);
            arok[it] = gr < M;
            azero[it] = false;
            if (a_idx) {
                const long long g = a_idx[(long)r * a_idx_ld];
                azero[it] = !(g >= 0 && g < a_ns);
                arow[it] = azero[it] ? a_zero : A + g * lda;
            } else {
                arow[it] = A + (long)r * lda;
            }
        }
    }
    // ANORM: A is the RAW output of a product whose InstanceNorm + LeakyReLU has not been applied: the statistics come
    // as fp64 column sums (a_sums [2][K] over a_count rows) and every A element is normalised on its way into the split,
    // a' = lrelu((a - mean_k) * rstd_k, a_slope) -- the consumer does the producer's normalisation pass (ref:models/
    // blocks.py:456-470 followed by the next block's nn.Linear).  (mean, rstd) of this block's k range live in LDS
    // behind the operand planes.  A gathered shadow row stays zero (closest_pool pads AFTER the normalisation).
    float* const s_mean = reinterpret_cast<float*>(smem + 3 * (A_PLANE + B_PLANE));
    float* const s_rstd = s_mean + (ANORM ? ((k_per_split + BK - 1) / BK) * BK : 0);
    int* const s_ovf = reinterpret_cast<int*>(s_rstd + (ANORM ? ((k_per_split + BK - 1) / BK) * BK : 0));   // H2: see below
    // H2: one word per tile row of A and of B collecting the row's largest |x| (the underflow check after the fp16 loop); it
    // lives in A's third plane, which the fp16 loop never touches
    unsigned* const s_seen = reinterpret_cast<unsigned*>(As + 2 * A_PLANE);
    static_assert(!H2 || (BM + BN) * 4 <= A_PLANE, "row words must fit the unused plane");
    if constexpr (H2) {
        if (tid == 0) *s_ovf = 0;
        for (int i = tid; i < BM + BN; i += NT) s_seen[i] = 0u;
        if constexpr (!ANORM) __syncthreads();
    }
    if constexpr (ANORM) {
        for (int kk = tid; kk < k_end - k_begin; kk += NT) {
            const double mu = a_sums[k_begin + kk] / a_count;
            double var = a_sums[(long)Kdim + k_begin + kk] / a_count - mu * mu;
            if (var < 0.0) var = 0.0;
            s_mean[kk] = (float)mu;
            s_rstd[kk] = (float)(1.0 / sqrt(var + (double)a_eps));
        }
        __syncthreads();
    }
    auto normalise = [&](Item<ALAY>* qa, int k0, bool guard) {      // in registers, before the split
        if constexpr (ANORM && ALAY == 0 && ATERMS == 3) {
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it) {
                const int kb = k0 - k_begin + ((tid + it * NT) & 3) * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kk = guard ? min(kb + j, k_end - k_begin - 1) : kb + j;
                    float v = (qa[it].get(j) - s_mean[kk]) * s_rstd[kk];
                    v = v >= 0.f ? v : v * a_slope;
                    const bool keep = !azero[it] && (!guard || kb + j < k_end - k_begin);
                    qa[it].set(j, keep ? v : 0.f);
                }
            }
        }
    };

    // two register sets: tile s is consumed from set s&1 while tiles s+1 (other set) and s+2 (this set, re-issued
    // right after its split) are in flight -- these GEMMs stream A from HBM, so bytes in flight are the currency
    Item<ALAY> ra[2][A_ITERS];
    Item<BLAY> rb[2][B_ITERS];
    // global loads per thread and tile (the bf16-A form issues one 16-byte load per item)
    constexpr int TILE_LOADS = (ATERMS == 1 ? 1 : Item<ALAY>::LOADS) * A_ITERS + Item<BLAY>::LOADS * B_ITERS;
    const unsigned short* const Ah = reinterpret_cast<const unsigned short*>(A);

    // FAST: both operands aligned for their loads and the slab k0..k0+31 inside [k_begin, k_end).  The choice is made
    // OUTSIDE the k-loop: a branch between load flavours inside it makes hipcc merge their vmcnt bookkeeping and
    // wait for (nearly) everything in flight at the top of every step, which turns prefetch distance 2 into 1.
    auto load_tiles = [&](auto fast, int k0, Item<ALAY>* qa, Item<BLAY>* qb) {
        if constexpr (decltype(fast)::value) {
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it) {
                if constexpr (ATERMS == 1) {
                    const int e = tid + it * NT;
                    gload16(qa[it].v[0], reinterpret_cast<const float*>(Ah + (long)min(m0 + (e >> 2), M - 1) * lda + k0 + (e & 3) * 8));
                } else if constexpr (ALAY == 0) {
                    const float* p = arow[it] + k0 + ((tid + it * NT) & 3) * 8;
                    gload16(qa[it].v[0], p);
                    gload16(qa[it].v[1], p + 4);
                } else {
                    item_load_fast<ALAY, BM>(qa[it], A, lda, m0, M, k0, tid + it * NT);
                }
            }
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it) item_load_fast<BLAY, BN>(qb[it], B, ldb, n0, N, k0, tid + it * NT);
        } else {
            if constexpr (ATERMS != 1) {      // (the bf16-A form is dispatched only for aligned operands and K % 32 == 0)
#pragma unroll
                for (int it = 0; it < A_ITERS; ++it) {
                    if constexpr (ALAY == 0) {
                        const int gk = k0 + ((tid + it * NT) & 3) * 8, ke = k_end - 1;
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = arow[it][min(gk + j, ke)];          // always a valid address
#pragma unroll
                        for (int j = 0; j < 8; ++j) qa[it].set(j, (arok[it] && gk + j < k_end) ? v[j] : 0.f);
                    } else {
                        item_load_edge<ALAY, BM>(qa[it], A, lda, m0, M, k0, k_end, tid + it * NT);
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it) item_load_edge<BLAY, BN>(qb[it], B, ldb, n0, N, k0, k_end, tid + it * NT);
        }
    };
    auto store_one = [&](unsigned char* base, int plane_bytes, int row, int kg, auto& src) {
        unsigned q1[4], q2[4], q3[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (KNOCK & 2) q1[j] = q2[j] = q3[j] = __float_as_uint(src.get(2 * j)) ^ __float_as_uint(src.get(2 * j + 1));
            else split2(src.get(2 * j), src.get(2 * j + 1), q1[j], q2[j], q3[j]);
        }
        const u32x4 p1 = {q1[0], q1[1], q1[2], q1[3]}, p2 = {q2[0], q2[1], q2[2], q2[3]},
                    p3 = {q3[0], q3[1], q3[2], q3[3]};
        unsigned char* d = base + row * ROWB + kg * 16;
        *reinterpret_cast<u32x4*>(d) = p1;
        *reinterpret_cast<u32x4*>(d + plane_bytes) = p2;
        *reinterpret_cast<u32x4*>(d + 2 * plane_bytes) = p3;
    };
    auto store_tiles = [&](Item<ALAY>* qa, Item<BLAY>* qb) {
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
            int row, kg;
            item_pos<ALAY, BM>(tid + it * NT, row, kg);
            if constexpr (ATERMS == 1) *reinterpret_cast<f32x4*>(As + row * ROWB + kg * 16) = qa[it].v[0];
            else store_one(As, A_PLANE, row, kg, qa[it]);
        }
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            int row, kg;
            item_pos<BLAY, BN>(tid + it * NT, row, kg);
            store_one(Bs, B_PLANE, row, kg, qb[it]);
        }
    };

    const int half = lane >> 5, l31 = lane & 31;
    const int nsteps = k_end > k_begin ? (k_end - k_begin + BK - 1) / BK : 0;
    auto multiply = [&]() {
#pragma unroll
        for (int c = 0; c < BK / 16; ++c) {
            bf16x8 a[TM][ATERMS], b[TN][3];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < ATERMS; ++p) {
                    if constexpr (KNOCK & 16) { u32x4 z = {(unsigned)lane, 1u, 2u, 3u}; asm volatile("" : "+v"(z)); a[i][p] = __builtin_bit_cast(bf16x8, z); }
                    else a[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                        As + p * A_PLANE + (wm * WM + i * 32 + l31) * ROWB + (c * 2 + half) * 16));
                }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    if constexpr (KNOCK & 16) { u32x4 z = {(unsigned)lane, 5u, 6u, 7u}; asm volatile("" : "+v"(z)); b[j][p] = __builtin_bit_cast(bf16x8, z); }
                    else b[j][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                        Bs + p * B_PLANE + (wn * WN + j * 32 + l31) * ROWB + (c * 2 + half) * 16));
                }
            if constexpr (KNOCK & 8) {              // operands stay live, no matrix instruction
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int p = 0; p < ATERMS; ++p) asm volatile("" ::"v"(a[i][p]));
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) asm volatile("" ::"v"(b[j][p]));
                continue;
            }
            // smallest terms first: a3b1 a2b2 a1b3 | a2b1 a1b2 | a1b1
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int pa = (t == 0 ? 2 : t == 1 ? 1 : t == 2 ? 0 : t == 3 ? 1 : 0);
                const int pb = (t == 0 ? 0 : t == 1 ? 1 : t == 2 ? 2 : t == 3 ? 0 : t == 4 ? 1 : 0);
                if (pa >= ATERMS) continue;      // bf16 A: only a1*b3, a1*b2, a1*b1
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
            }
        }
    };
    // steps [0, nfast) use the fast loads; a k tail (K not a multiple of 32) or unaligned operands use guarded ones
    const int nfast = (vec_a && vec_b) ? (k_end - k_begin) / BK : 0;
    // Wait until only the other set's loads are in flight, then split this set into LDS.  `live` = 0 zeroes the tile
    // (the phantom second half of an odd tile count).
    auto consume = [&](auto last, Item<ALAY>* qa, Item<BLAY>* qb, bool live, int k0) {
        if constexpr (!(KNOCK & 1) && !decltype(last)::value) vm_wait<TILE_LOADS>();
        else vm_wait<0>();                                                // the last tile: no other set in flight behind it
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
            if constexpr (ATERMS == 1) pin(qa[it].v[0]);
            else qa[it].hold();
        }
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) qb[it].hold();
        if (!live) {
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it)
#pragma unroll
                for (int j = 0; j < (ATERMS == 1 ? 4 : 8); ++j) qa[it].set(j, 0.f);
        } else {
            normalise(qa, k0, false);
        }
        if constexpr (!(KNOCK & 32)) __syncthreads();                     // previous tile fully read
        if constexpr (!(KNOCK & 4)) store_tiles(qa, qb);                  // registers -> LDS (split)
        else if constexpr (KNOCK & 2) {                                   // neither split nor stores: the loaded registers stay live
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(qa[it].get(j)));
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(qb[it].get(j)));
        } else {                                                          // (keep the split alive without the stores)
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it) {
                unsigned q1, q2, q3;
                if constexpr (ATERMS == 3) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        split2(qa[it].get(2 * j), qa[it].get(2 * j + 1), q1, q2, q3);
                        asm volatile("" ::"v"(q1), "v"(q2), "v"(q3));
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it) {
                unsigned q1, q2, q3;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    split2(qb[it].get(2 * j), qb[it].get(2 * j + 1), q1, q2, q3);
                    asm volatile("" ::"v"(q1), "v"(q2), "v"(q3));
                }
            }
        }
    };
    bool run_x6 = true;
    if constexpr (H2) {
        // ---- the fp16 two-term loop (see split2h): the structure of the loop below with two planes, three products and the
        // range check; a workgroup that meets a value fp16 cannot hold leaves it and starts over with the bf16 loop
        static_assert(ATERMS == 3, "the fp16 form is built for fp32 operands");
        // H2 == 2 (the train step's products): h2_sa / h2_sb are exact power-of-two factors applied to A / B before the split
        // (the GRADIENT operand, whose values live far below fp16's normal range, is lifted by 2^16) and taken out of the
        // sums again right after the loop.  H2 == 1 (every forward product): no lift, no code for it.
        float seen_a[A_ITERS], seen_b[B_ITERS];
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) seen_a[it] = 0.f;
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) seen_b[it] = 0.f;
        auto store_one_h = [&](unsigned char* base, int plane_bytes, int row, int kg, auto& src) {
            unsigned q1[4], q2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split2h(src.get(2 * j), src.get(2 * j + 1), q1[j], q2[j]);
            const u32x4 p1 = {q1[0], q1[1], q1[2], q1[3]}, p2 = {q2[0], q2[1], q2[2], q2[3]};
            unsigned char* d = base + row * ROWB + kg * 16;
            *reinterpret_cast<u32x4*>(d) = p1;
            *reinterpret_cast<u32x4*>(d + plane_bytes) = p2;
        };
        // the row's largest |x| so far: four v_max3_f32 per item.  Taken BEFORE the split, while the loaded registers are
        // still what the split reads: a later use of them lengthens their live range, the allocator then moves the
        // asm-loaded values to other registers, and it places those copies before the s_waitcnt that completes the loads
        // (seen in the ISA of the first form of this check; tests/test_isa_hazards.py scans every build for it).
        auto track = [&](auto& src, float& seen) {
#pragma unroll
            for (int j = 0; j < 4; ++j) seen = fmaxf(fmaxf(fabsf(src.get(2 * j)), fabsf(src.get(2 * j + 1))), seen);
        };
        auto multiply_h = [&]() {
#pragma unroll
            for (int c = 0; c < BK / 16; ++c) {
                f16x8 a[TM][2], b[TN][2];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        a[i][p] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(
                            As + p * A_PLANE + (wm * WM + i * 32 + l31) * ROWB + (c * 2 + half) * 16));
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        b[j][p] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(
                            Bs + p * B_PLANE + (wn * WN + j * 32 + l31) * ROWB + (c * 2 + half) * 16));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc_lo[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc_lo[i][j], 0, 0, 0);
                    }
            }
        };
        auto consume_h = [&](auto last, Item<ALAY>* qa, Item<BLAY>* qb, bool live, int k0) {
            if constexpr (decltype(last)::value) vm_wait<0>();            // the last tile: no other set in flight behind it
            else vm_wait<TILE_LOADS>();
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it) qa[it].hold();
            if constexpr (!ANORM) {
#pragma unroll
                for (int it = 0; it < B_ITERS; ++it) qb[it].hold();
            }
            if (!live) {
#pragma unroll
                for (int it = 0; it < A_ITERS; ++it)
#pragma unroll
                    for (int j = 0; j < 8; ++j) qa[it].set(j, 0.f);
            } else {
                normalise(qa, k0, false);
            }
            if constexpr (ANORM) {        // B is taken up only now: its registers stay where the loads put them while A is normalised
#pragma unroll
                for (int it = 0; it < B_ITERS; ++it) qb[it].hold();
            }
            if constexpr (H2 == 2) {      // the lifted form only (as a run-time test hipcc turns this into a multiply AND a
                                          // select per element -- 12 of the 36 VALU instructions per item, in every product)
#pragma unroll
                for (int it = 0; it < A_ITERS; ++it)
#pragma unroll
                    for (int j = 0; j < 8; ++j) qa[it].set(j, qa[it].get(j) * h2_sa);
#pragma unroll
                for (int it = 0; it < B_ITERS; ++it)
#pragma unroll
                    for (int j = 0; j < 8; ++j) qb[it].set(j, qb[it].get(j) * h2_sb);
            }
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it) track(qa[it], seen_a[it]);
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it) track(qb[it], seen_b[it]);
            __builtin_amdgcn_sched_barrier(0);                            // (keeps the maxima HERE: see track)
            __syncthreads();                                              // previous tile fully read
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it) {
                int row, kg;
                item_pos<ALAY, BM>(tid + it * NT, row, kg);
                store_one_h(As, A_PLANE, row, kg, qa[it]);
            }
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it) {
                int row, kg;
                item_pos<BLAY, BN>(tid + it * NT, row, kg);
                store_one_h(Bs, B_PLANE, row, kg, qb[it]);
            }
        };
        if (nfast > 0) {
            std::true_type fast;
            const int k_last = k_begin + (nfast - 1) * BK;
            std::false_type more;
            std::true_type last;
            load_tiles(fast, k_begin, ra[0], rb[0]);
            load_tiles(fast, min(k_begin + BK, k_last), ra[1], rb[1]);
            // Tiles s, s + 1 are consumed while s + 2, s + 3 are fetched -- as long as there ARE tiles to fetch: the last two
            // tiles run outside the loop with nothing behind them (round 6; before, every product re-read its last tile
            // twice as "prefetch": with K = 64 that was half of all tile loads of a kernel bound by its L1 fills).
            int s = 0;
            for (; s + 2 < nfast; s += 2) {
                consume_h(more, ra[0], rb[0], true, k_begin + s * BK);
                load_tiles(fast, k_begin + (s + 2) * BK, ra[0], rb[0]);
                __syncthreads();
                multiply_h();
                consume_h(more, ra[1], rb[1], true, k_begin + (s + 1) * BK);
                load_tiles(fast, min(k_begin + (s + 3) * BK, k_last), ra[1], rb[1]);   // (an odd count: the phantom's load)
                __syncthreads();
                multiply_h();
            }
            consume_h(more, ra[0], rb[0], true, k_begin + s * BK);
            __syncthreads();
            multiply_h();
            consume_h(last, ra[1], rb[1], s + 1 < nfast, k_begin + (s + 1) * BK);
            __syncthreads();
            multiply_h();
        }
        // The range check costs the loop nothing: an operand at or beyond fp16's range became +-inf in BOTH of its terms
        // (h = inf, l = (x - inf) * 2^11 = -inf), and inf times anything -- zero included -- leaves inf or NaN in every
        // sum of its row (A) or column (B) of the tile, which nothing can cancel.  So: any non-finite partial sum anywhere
        // in the workgroup -> the whole tile again with the loop below (which also gives inf / NaN INPUTS the result it
        // always gave them).
        bool bad = false;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) bad = bad || !(fabsf(acc[i][j][r]) <= 3.0e38f) || !(fabsf(acc_lo[i][j][r]) <= 3.0e38f);
        if (bad) *s_ovf = 1;
        __syncthreads();
        const bool over = *s_ovf != 0;
        // ... and the other end of fp16's range.  Below 2^-14 h is an fp16 SUBNORMAL (below 2^-25 it is zero): the split then
        // has an absolute floor of 2^-36 per value instead of 2^-22 relative.  That is harmless for small values BESIDE larger
        // ones of the same row (the floor stays below 2^-22 of the row's largest value as long as that one is a normal fp16),
        // and it is a loss of fp32's contract for a row -- of A or of B, i.e. an output row or column -- whose values are ALL
        // that small (measured 1e-6 relative at |x| ~ 1e-5, 1e-3 at 1e-8, everything at 1e-11).  So every thread keeps the
        // largest |x| of what it splits (four v_max3_f32 per item, the only cost inside the loop; after the lift, if any),
        // the row's maxima meet in LDS here (as integers: non-negative floats order like their bit patterns), and a row that
        // is not all zeros but has no value of at least 2^-14 sends the tile to the three-term bf16 loop, whose terms are
        // exact for every finite fp32 value down to 2^-110.
        if (nfast > 0) {
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it) {
                int row, kg;
                item_pos<ALAY, BM>(tid + it * NT, row, kg);
                if (seen_a[it] > 0.f) atomicMax(&s_seen[row], __float_as_uint(seen_a[it]));
            }
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it) {
                int row, kg;
                item_pos<BLAY, BN>(tid + it * NT, row, kg);
                if (seen_b[it] > 0.f) atomicMax(&s_seen[BM + row], __float_as_uint(seen_b[it]));
            }
        }
        __syncthreads();
        for (int i = tid; i < BM + BN; i += NT) {
            const unsigned v = s_seen[i];
            if (v != 0u && v < 0x38800000u) *s_ovf = 1;          // 0x38800000 = 2^-14, the smallest normal fp16
        }
        __syncthreads();
        run_x6 = *s_ovf != 0;
        if (run_x6 && tid == 0) atomicAdd(&g_x6_redo[over ? 0 : 1], 1ull);
        if (run_x6) {                   // out of fp16's range: everything again, with the loop that has fp32's
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.0f; acc_lo[i][j][r] = 0.0f; }
            __syncthreads();            // nobody still reads the fp16 planes
        } else {                        // the fp16 sums stand: cross terms in, operand scales out (the k tail below adds plain sums)
            const float unscale = H2 == 2 ? 1.0f / (h2_sa * h2_sb) : 1.0f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] + acc_lo[i][j][r] * (1.0f / kH2Scale)) * unscale;
        }
    }
    if (run_x6 && nfast > 0) {
        // ONE region without control-flow joins between a load and its wait: every join makes hipcc copy the
        // (still in flight) destination registers of the asm loads, which is exactly the garbage the guide warns of
        // (seen in the ISA of an earlier version with peeled tail steps: v_mov of a set before its s_waitcnt).
        // So all loads are unconditional -- past the last tile they re-read it (an L1/L2 hit) and are never used --,
        // an odd tile count is rounded up with a zeroed phantom tile, and nothing is peeled.
        std::true_type fast;
        const int k_last = k_begin + (nfast - 1) * BK;
        std::false_type more;
        std::true_type last;
        load_tiles(fast, k_begin, ra[0], rb[0]);
        load_tiles(fast, min(k_begin + BK, k_last), ra[1], rb[1]);
        int s = 0;
        for (; s + 2 < nfast; s += 2) {              // (as in the fp16 loop: the last two tiles have nothing to prefetch behind them)
            consume(more, ra[0], rb[0], true, k_begin + s * BK);
            if constexpr (!(KNOCK & 1)) load_tiles(fast, k_begin + (s + 2) * BK, ra[0], rb[0]);   // tile s+2 into the freed set
            if constexpr (!(KNOCK & 32)) __syncthreads();
            multiply();
            consume(more, ra[1], rb[1], true, k_begin + (s + 1) * BK);
            if constexpr (!(KNOCK & 1)) load_tiles(fast, min(k_begin + (s + 3) * BK, k_last), ra[1], rb[1]);
            if constexpr (!(KNOCK & 32)) __syncthreads();
            multiply();
        }
        consume(more, ra[0], rb[0], true, k_begin + s * BK);
        if constexpr (!(KNOCK & 32)) __syncthreads();
        multiply();
        consume(last, ra[1], rb[1], s + 1 < nfast, k_begin + (s + 1) * BK);
        if constexpr (!(KNOCK & 32)) __syncthreads();
        multiply();
    }
    for (int s = nfast; s < nsteps; ++s) {       // k tail / unaligned operands: guarded loads hipcc counts itself
        std::false_type slow;
        load_tiles(slow, k_begin + s * BK, ra[0], rb[0]);
        normalise(ra[0], k_begin + s * BK, true);
        __syncthreads();
        store_tiles(ra[0], rb[0]);
        __syncthreads();
        multiply();
    }

    // epilogue (as k_gemm_f32): C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool first_split = split == 0;
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
    float cs[TN], cq[TN];
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
    if (colp && colp_chunks < 0) {
        // sums mode: [2][N] accumulators zeroed by the caller.  The workgroup's two row halves meet in LDS first (the operand
        // planes are free now), so a row tile costs one pair of atomics per column, not two
        float* const red = reinterpret_cast<float*>(smem);
        __syncthreads();                                   // every wavefront is past its last operand read
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float s = cs[j] + __shfl_xor(cs[j], 32, 64), q2 = cq[j] + __shfl_xor(cq[j], 32, 64);
            if (wm == 1 && half == 0) {
                red[2 * (wn * WN + j * 32 + l31)] = s;
                red[2 * (wn * WN + j * 32 + l31) + 1] = q2;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float s = cs[j] + __shfl_xor(cs[j], 32, 64), q2 = cq[j] + __shfl_xor(cq[j], 32, 64);
            const int gn = n0 + wn * WN + j * 32 + l31;
            if (wm == 0 && half == 0 && gn < N) {
                unsafeAtomicAdd(&colp[gn], (double)s + (double)red[2 * (wn * WN + j * 32 + l31)]);
                unsafeAtomicAdd(&colp[(long)N + gn], (double)q2 + (double)red[2 * (wn * WN + j * 32 + l31) + 1]);
            }
        }
    } else if (colp) {
        const int chunk = tile_y * WAVES_M + wm;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float s = cs[j] + __shfl_xor(cs[j], 32, 64), q2 = cq[j] + __shfl_xor(cq[j], 32, 64);
            const int gn = n0 + wn * WN + j * 32 + l31;
            if (half == 0 && gn < N) {
                {
                    colp[(long)gn * colp_chunks + chunk] = (double)s;
                    colp[((long)N + gn) * colp_chunks + chunk] = (double)q2;
                }
            }
        }
    }
}

// set by gemm_x6_dispatch around a launch whose splits store partial tiles (deterministic mode), 0 / false otherwise
thread_local long g_split_stride = 0;
thread_local bool g_det_pass = false;
// the deterministic mode's partial-tile buffer: one per stream, grown on demand (a debugging mode: synchronous allocation)
struct DetBuf { float* p = nullptr; size_t n = 0; };
std::mutex g_det_mu;
std::map<hipStream_t, DetBuf> g_det_bufs;
float* det_partials(hipStream_t st, size_t floats) {
    std::lock_guard<std::mutex> g(g_det_mu);
    DetBuf& b = g_det_bufs[st];
    if (b.n < floats) {
        if (hipStreamSynchronize(st) != hipSuccess) return nullptr;
        if (b.p) (void)hipFree(b.p);
        b.p = nullptr;
        b.n = 0;
        const size_t want = floats + floats / 4;
        if (hipMalloc(&b.p, want * sizeof(float)) != hipSuccess) { b.p = nullptr; return nullptr; }
        b.n = want;
    }
    return b.p;
}

// C (+)= sum over the splits' partial tiles, in split order: the second pass of the deterministic mode's split-K
__global__ void __launch_bounds__(256) k_split_reduce(const float* __restrict__ part, long stride, int splits, float* __restrict__ c,
                                                      int ldc, int m, int n, int accumulate) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)m * n) return;
    const long r = t / n;
    const int col = (int)(t - r * n);
    float sum = part[t];
    for (int s = 1; s < splits; ++s) sum += part[(long)s * stride + t];
    float* dst = c + r * ldc + col;
    *dst = accumulate ? *dst + sum : sum;
}

template <int BM, int BN, int MINB, int ATERMS, int ALAY, int BLAY, int ANORM = 0, int KNOCK = 0, int H2 = 0>
int launch_x6(dim3 grid, hipStream_t st, const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n,
              int k, const float* row_scale, const float* bias, int k_per_split, int vec_a, int vec_b, int atomic_out,
              double* colp, int colp_chunks, const long long* a_idx = nullptr, int a_idx_ld = 0, int a_ns = 0,
              const float* a_zero = nullptr, const double* a_sums = nullptr, double a_count = 0.0, float a_eps = 0.f,
              float a_slope = 1.f, GemmPairArgs pr = GemmPairArgs(), float h2_sa = 1.f, float h2_sb = 1.f) {
    // grid = (column tiles, row tiles, splits) as the caller counts them; launched 1-D (see the kernel's tile order)
    TileMap tm;
    tm.split_stride = g_split_stride;
    tm.gx = (int)grid.x;
    tm.gy = (int)grid.y;
    tm.gs = (int)grid.z;
    tm.order = x6_tile_order(tm.gx, tm.gy, tm.gs, BM, BN, k_per_split);
    const size_t lds = lds_bytes<BM, BN>() + (ANORM ? 2 * sizeof(float) * (size_t)(((k_per_split + BK - 1) / BK) * BK) : 0) + (H2 ? 16 : 0);
    auto kern = k_gemm_x6<BM, BN, MINB, ATERMS, ALAY, BLAY, ANORM, KNOCK, H2>;
    PCRCG_GRANT_LDS(kern);
    int m_all = m;                                             // a grouped launch's products all count
    for (int e = 0; e < pr.extra; ++e) m_all += pr.m[e];
    KpProfScope prof(st, m_all, n, k, (ATERMS == 1 || H2) ? 3 : 6, 3);  // bench.py's GEMM roofline: the kernel's own start / stop events
    hipExtLaunchKernelGGL(kern, dim3(tm.gx * tm.gy * tm.gs), dim3(x6_threads<BM, BN>()), lds, st, prof.a, prof.b, 0, a, lda, b, ldb, c, ldc, m, n,
                          k, row_scale, bias, k_per_split, vec_a, vec_b, atomic_out, colp, colp_chunks, a_idx, a_idx_ld, a_ns,
                          a_zero, a_sums, a_count, a_eps, a_slope, pr, tm, h2_sa, h2_sb);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}


}  // namespace

// Tile / split choice from the sweep of scripts/gemm_x6_bench.py over the path's shapes: the 64x64 tile wins or
// ties everywhere (the kernel is bound by per-block latency, so many small blocks beat few large ones) except
// for very tall N <= 64 products, where 128x64 halves the re-reads of B.  K is split (fp32 atomics into a
// zeroed C) only when the tiles alone leave most CUs idle, or when K is so long that one block's k-loop
// dominates; each split costs atomic traffic (and a memset unless the caller hands over a zeroed C), so never
// below 256 k per split.
// set (per host thread) by a caller that enqueues its forwards beside other streams' -- the pair engine's model threads
// (pcrcg_thread_shares_gpu, include/pcrcg.h); read by the plan below
static thread_local int g_x6_shared = 0;
bool gemm_x6_shared() { return g_x6_shared > 0; }
void gemm_x6_set_shared(int on) { g_x6_shared = on ? 1 : 0; }

struct X6Plan { int pick, bm, bn, gx, gy, splits, k_per_split; };
static X6Plan x6_plan_for(int pick, int m, int n, int k, bool reduce_rows) {
    static const int tiles[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
    X6Plan p;
    p.pick = pick;
    p.bm = tiles[p.pick][0];
    p.bn = tiles[p.pick][1];
    p.gx = (n + p.bn - 1) / p.bn;
    p.gy = (m + p.bm - 1) / p.bm;
    int splits = 1;
    const int ktiles = (k + BK - 1) / BK;
    const int max_splits = reduce_rows ? 128 : 32;            // dW = X^T dY reduces over the points: few tiles, very long K
    // Split targets.  Alone on the GPU a product wants ~200 workgroups before it stops splitting (and 1024 while K is very
    // long).  Beside the pair engine's other streams the idle CUs of a small product are filled by their kernels anyway,
    // while every split costs fp32 atomics and a zeroed output: a host thread that has declared the GPU shared
    // (pcrcg_thread_shares_gpu) does not split at all (targets 1 / 1).  Round 4: 32 / 128, +4.5 % in the engine (483 -> 505
    // pairs/s; no splitting at all: 506), where the same targets cost a forward running alone 1.9 % (3.24 -> 3.30 ms) and the
    // train step 4 % (15.4 -> 16.1 ms).  Round 6, after the k-loop stopped loading behind its last tile: 577.0 with 32 / 128,
    // 581 with 16 / 64, 582.7-583.2 with no splits (574 with 64 / 256; profiles/r06_ab_splitk_targets.txt) -- and an unsplit
    // product leaves its column statistics in its epilogue and needs neither atomics nor a zeroed output.
    // dW = X^T dY (reduce_rows: a handful of tiles, K = all points) always keeps the lone-stream targets.
    int t1 = 200, t2 = 1024;
    if (gemm_x6_shared() && !reduce_rows) { t1 = debug_opts().x6_t1; t2 = debug_opts().x6_t2; }   // tuning aids (defaults 1 / 1: no splits)
    while ((long)p.gx * p.gy * splits < t1 && k / (2 * splits) >= 256 && splits < max_splits) splits *= 2;
    while ((long)p.gx * p.gy * splits < t2 && k / splits > 1024 && splits < max_splits) splits *= 2;
    if (debug_opts().x6_splitk > 0) splits = debug_opts().x6_splitk;        // tuning aid
    p.k_per_split = ((ktiles + splits - 1) / splits) * BK;
    if (p.k_per_split < BK) p.k_per_split = BK;
    p.splits = k > 0 ? (k + p.k_per_split - 1) / p.k_per_split : 1;
    if (p.splits < 1) p.splits = 1;
    return p;
}
// m: rows of the (largest) product; m_total: rows of the whole launch (the products of a group share the plan)
static X6Plan x6_plan(int m, int n, int k, bool kmajor = false, bool reduce_rows = false, long m_total = 0) {
    int pick = debug_opts().x6_tile;                                         // tuning aid (-1: automatic)
    if (kmajor) return x6_plan_for(3, m, n, k, reduce_rows);  // the k-major operand forms are built for 64 x 64 only
    if (pick >= 0 && pick <= 3) return x6_plan_for(pick, m, n, k, reduce_rows);
    // x6_big=1 (off by default): 128 x 128 on eight wavefronts halves the L1 fills per flop (profiles/
    // r03_gemm_load_skeleton.txt) and wins 5-10 % in a back-to-back loop where its tiles still fill the chip (15456x128x1920
    // 58.5 vs 63.0 us, 3934x256x3840 57.1 vs 63.6, 60000x256x128 39.7 vs 41.8; 15456x128x512 with 121 tiles 28.0 vs 22.4)
    // -- but inside a forward it gains nothing (GEMM time per forward 1.86 vs 1.82 ms) and inside the four-stream engine it
    // loses (442 vs 454 pairs/s, same box): two 512-thread workgroups hold a CU that four small ones share more gracefully
    if (n >= 128 && debug_opts().x6_big) {
        const X6Plan big = x6_plan_for(0, m, n, k, reduce_rows);
        const long rows = m_total > m ? m_total : m;
        if ((long)big.gx * ((rows + 127) / 128) * big.splits >= 240) return big;
    }
    return x6_plan_for((n <= 64 && m >= 32768) ? 1 : 3, m, n, k, reduce_rows);
}

// tiles the fp16 form handed to the bf16 redo since the last reset: out[0] beyond fp16's range, out[1] rows below it
int gemm_x6_redo_counts(unsigned long long* out, int reset) {
    if (out) PCRCG_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x6_redo), sizeof(unsigned long long) * 2));
    if (reset) {
        const unsigned long long z[2] = {0ull, 0ull};
        PCRCG_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_x6_redo), z, sizeof(z)));
    }
    return PCRCG_OK;
}

// the split-K factor gemm_x6_dispatch uses for an [m, n, k] product (> 1: it accumulates into a zeroed C)
int gemm_x6_splits(int m, int n, int k, long m_total) { return (m > 0 && n > 0) ? x6_plan(m, n, k, false, false, m_total).splits : 1; }

// pcrcg_debug_release(): the deterministic mode's partial-tile buffers of every stream (the caller has drained them)
void gemm_x6_release_det() {
    std::lock_guard<std::mutex> g(g_det_mu);
    for (auto& kv : g_det_bufs)
        if (kv.second.p) (void)hipFree(kv.second.p);
    g_det_bufs.clear();
}

// Called by gemm_dispatch (gemm.hip) for C = A * B^T products when the split-bf16 mode is on.  c_zeroed: C is
// already all zeros (the runner's zero arena), so a split-K product needs no memset of its own.
int gemm_x6_dispatch(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k,
                     const float* row_scale, const float* bias, void* colstats, size_t colstats_bytes, int* h_chunks,
                     hipStream_t st, bool a_bf16, bool c_zeroed, int a_kmajor, int b_kmajor, bool colstats_sums,
                     const GemmExtra* ex, const GemmGroup* grp) {
    // k-major operands (a_kmajor: A stored [K, M]; b_kmajor: B stored [K, N]) are read with 4-byte loads: no alignment rule
    const int vec_a = (a_bf16 || a_kmajor) ? 1 : (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0);
    const int vec_b = b_kmajor ? 1 : (ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(b) & 15) == 0);
    if (a_kmajor && !b_kmajor) { set_error("gemm_x6: A^T * B^T is not built"); return PCRCG_EBADARG; }
    // (further products in the same launch: the plan of the largest one serves all)
    int m_plan = m;
    const int n_extra = grp ? grp->n : 0;
    if (n_extra < 0 || n_extra > kGroupExtra) { set_error("gemm_x6: at most %d further products per launch", kGroupExtra); return PCRCG_EBADARG; }
    for (int e = 0; e < n_extra; ++e) {
        if (grp->p[e].m <= 0 || !grp->p[e].a || !grp->p[e].c) { set_error("gemm_x6: empty product in a group"); return PCRCG_EBADARG; }
        m_plan = grp->p[e].m > m_plan ? grp->p[e].m : m_plan;
    }
    long m_total = m;
    for (int e = 0; e < n_extra; ++e) m_total += grp->p[e].m;
    // (the lifted fp16 form of the train step's products is built for the 64 x 64 tile only, like the k-major forms)
    const bool lifted = ex && ex->grad_operand && debug_opts().x6_h2 != 0 && !a_bf16;
    const X6Plan plan = x6_plan(m_plan, n, k, a_kmajor || b_kmajor || (ex && ex->a_sums) || lifted, a_kmajor != 0, m_total);
    const int pick = plan.pick, BM = plan.bm, BN = plan.bn, gx = plan.gx, splits = plan.splits;
    const int gy0 = (m + BM - 1) / BM;
    int gy = gy0, gye[kGroupExtra] = {0, 0, 0};
    for (int e = 0; e < n_extra; ++e) { gye[e] = (grp->p[e].m + BM - 1) / BM; gy += gye[e]; }
    if (n_extra && (a_kmajor || b_kmajor)) { set_error("gemm_x6: grouped launches are built for the A * B^T form"); return PCRCG_EBADARG; }
    const int k_per_split = plan.k_per_split;
    const bool accumulate = ex && ex->accumulate;
    const bool gather = ex && ex->a_idx;
    if ((gather && (a_bf16 || a_kmajor || b_kmajor)) || (accumulate && a_bf16)) {
        set_error("gemm_x6: gather is built for k-contiguous fp32 operands, accumulate for fp32 operands");
        return PCRCG_EBADARG;
    }
    if (gather && !ex->a_zero) { set_error("gemm_x6: gather needs a zero row"); return PCRCG_EBADARG; }
    const bool anorm = ex && ex->a_sums;
    if (anorm && (a_bf16 || a_kmajor || b_kmajor || pick != 3)) {
        set_error("gemm_x6: normalise-on-load is built for the 64 x 64 tile of k-contiguous fp32 operands");
        return PCRCG_EBADARG;
    }
    // ---- deterministic mode (PCRCG_DEBUG=deterministic=1): split-K WITHOUT atomics.  The splits store their partial tiles
    // to a scratch buffer (this very function once more, with C redirected and g_det_pass set: plain stores, split s at
    // s * split_stride), and k_split_reduce adds them up in split order -- onto C when the product accumulates.  Costs one
    // extra pass over splits x M x N floats; the scratch is the library's own (per stream, grown on demand).
    if (splits > 1 && debug_opts().deterministic && !g_det_pass) {
        const size_t rows_all = (size_t)m_total;
        float* part = det_partials(st, (size_t)splits * rows_all * (size_t)n);
        if (!part) { set_error("gemm_x6: no memory for the deterministic mode's partial tiles"); return PCRCG_ELAUNCH; }
        GemmExtra ex2;
        if (ex) ex2 = *ex;
        ex2.accumulate = 0;
        GemmGroup grp2;
        size_t row = (size_t)m;
        if (grp) {
            grp2 = *grp;
            for (int e = 0; e < n_extra; ++e) {
                grp2.p[e].c = part + row * (size_t)n;
                grp2.p[e].c_zeroed = true;
                grp2.p[e].colstats = nullptr;
                if (grp->p[e].h_chunks) *grp->p[e].h_chunks = 0;
                grp2.p[e].h_chunks = nullptr;
                row += (size_t)grp->p[e].m;
            }
        }
        g_det_pass = true;
        g_split_stride = (long)(rows_all * (size_t)n);
        const int rc = gemm_x6_dispatch(a, lda, b, ldb, part, n, m, n, k, row_scale, bias, nullptr, 0, nullptr, st, a_bf16, true,
                                        a_kmajor, b_kmajor, false, ex ? &ex2 : nullptr, grp ? &grp2 : nullptr);
        g_det_pass = false;
        g_split_stride = 0;
        if (rc != PCRCG_OK) return rc;
        row = 0;
        for (int e = -1; e < n_extra; ++e) {
            const int me = e < 0 ? m : grp->p[e].m;
            float* ce = e < 0 ? c : grp->p[e].c;
            const long total = (long)me * n;
            hipLaunchKernelGGL(k_split_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part + row * (size_t)n,
                               (long)(rows_all * (size_t)n), splits, ce, ldc, me, n, accumulate ? 1 : 0);
            row += (size_t)me;
        }
        PCRCG_CHECK_LAUNCH();
        return PCRCG_OK;
    }
    const int atomic_out = g_det_pass ? 0 : (splits > 1 || accumulate);
    if (splits > 1 && !c_zeroed && !accumulate) {
        if (ldc == n) PCRCG_CHECK_HIP(hipMemsetAsync(c, 0, (size_t)m * n * sizeof(float), st));
        else PCRCG_CHECK_HIP(hipMemset2DAsync(c, (size_t)ldc * sizeof(float), 0, (size_t)n * sizeof(float), m, st));
    }
    for (int e = 0; e < n_extra; ++e)
        if (splits > 1 && !grp->p[e].c_zeroed && !accumulate) {
            if (ldc == n) PCRCG_CHECK_HIP(hipMemsetAsync(grp->p[e].c, 0, (size_t)grp->p[e].m * n * sizeof(float), st));
            else PCRCG_CHECK_HIP(hipMemset2DAsync(grp->p[e].c, (size_t)ldc * sizeof(float), 0, (size_t)n * sizeof(float), grp->p[e].m, st));
        }
    dim3 grid(gx, gy, splits);
    const bool log_shapes = debug_opts().gemm_log != 0;   // tuning aid
    if (log_shapes)
        fprintf(stderr, "pcrcg_gemm_x6 m=%d n=%d k=%d lda=%d ldb=%d ldc=%d tile=%dx%d grid=%dx%dx%d rs=%d bias=%d stats=%d\n", m,
                n, k, lda, ldb, ldc, BM, BN, gx, gy, splits, row_scale != nullptr, bias != nullptr, colstats != nullptr);
    double* colp = nullptr;
    int colp_chunks = 0;
    if (colstats && h_chunks && !atomic_out && colstats_sums) {
        // column sums by fp64 atomics into [2][n] accumulators the caller has zeroed (few row tiles: no contention to
        // speak of); *h_chunks = -1 tells the caller that the buffer holds sums, not partials
        if (2 * (size_t)n * sizeof(double) <= colstats_bytes) {
            colp = static_cast<double*>(colstats);
            colp_chunks = -1;
            *h_chunks = -1;
        }
    } else if (colstats && h_chunks && !atomic_out) {
        colp_chunks = gy0 * 2;   // WAVES_M
        if (carve_bytes(2 * (size_t)n * colp_chunks, sizeof(double)) <= colstats_bytes) {
            colp = static_cast<double*>(colstats);
            *h_chunks = colp_chunks;
        } else {
            colp_chunks = 0;
        }
    }
    GemmPairArgs pa;
    pa.extra = n_extra;
    int row_off = gy0 * BM;
    for (int e = 0; e < n_extra; ++e) {
        const GemmPair& q = grp->p[e];
        pa.off[e] = row_off;
        row_off += gye[e] * BM;
        pa.m[e] = q.m;
        pa.a[e] = q.a;
        pa.c[e] = q.c;
        pa.rs[e] = q.row_scale;
        pa.a_idx[e] = q.a_idx;
        pa.a_ns[e] = q.a_ns;
        pa.a_sums[e] = q.a_sums;
        pa.a_count[e] = q.a_count;
        if (q.h_chunks) *q.h_chunks = 0;
        // the product's statistics: the same form as the first one's, into its own buffer (same size rule)
        if (colp && q.colstats && q.h_chunks) {
            pa.colp[e] = static_cast<double*>(q.colstats);
            pa.colp_chunks[e] = colp_chunks < 0 ? -1 : gye[e] * 2;
            *q.h_chunks = pa.colp_chunks[e];
            if (colp_chunks > 0 && carve_bytes(2 * (size_t)n * pa.colp_chunks[e], sizeof(double)) > colstats_bytes) {
                pa.colp[e] = nullptr;
                pa.colp_chunks[e] = 0;
                *q.h_chunks = 0;
            }
        }
    }
    // The train step tells which operand holds GRADIENTS (GemmExtra::grad_operand): the fp16 form lifts it by 2^16 (exactly):
    // rows whose gradients reach 2^-30 = 9.3e-10 then split as normal fp16 values; rows entirely below that, and values
    // beyond 1, send their tile to the bf16 redo (the kernel's two range checks) -- fp32-class at every scale, the lift only
    // decides how often the cheap loop suffices.  A k-major product whose caller says nothing keeps the bf16 form.
    const int grad_op = ex ? ex->grad_operand : 0;
    const float h2_sa = grad_op == 1 ? 65536.0f : 1.0f, h2_sb = grad_op == 2 ? 65536.0f : 1.0f;
    const bool h2_on = debug_opts().x6_h2 != 0 && !a_bf16;
    if (a_kmajor) {  // dW = X^T * dY
        if (h2_on && grad_op)
            return launch_x6<64, 64, 3, 3, 1, 1, 0, 0, 2>(grid, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias, k_per_split, vec_a,
                                                          vec_b, atomic_out, colp, colp_chunks, nullptr, 0, 0, nullptr, nullptr, 0.0, 0.f,
                                                          1.f, GemmPairArgs(), h2_sa, h2_sb);
        return launch_x6<64, 64, 3, 3, 1, 1>(grid, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias, k_per_split, vec_a,
                                             vec_b, atomic_out, colp, colp_chunks);
    }
    if (b_kmajor) {  // dX = dY * W
        if (h2_on && grad_op)
            return launch_x6<64, 64, 4, 3, 0, 1, 0, 0, 2>(grid, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias, k_per_split, vec_a,
                                                          vec_b, atomic_out, colp, colp_chunks, nullptr, 0, 0, nullptr, nullptr, 0.0, 0.f,
                                                          1.f, GemmPairArgs(), h2_sa, h2_sb);
        return launch_x6<64, 64, 4, 3, 0, 1>(grid, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias, k_per_split, vec_a,
                                             vec_b, atomic_out, colp, colp_chunks);
    }
    const bool h2 = h2_on;
    if (h2 && grad_op && !anorm)      // the k-contiguous dX = dY W^T of the autograd mirror (pcrcg_gemm_f32_grad): the lifted 64 x 64 form
        return launch_x6<64, 64, 4, 3, 0, 0, 0, 0, 2>(grid, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias, k_per_split, vec_a,
                                                      vec_b, atomic_out, colp, colp_chunks, gather ? ex->a_idx : nullptr,
                                                      gather ? ex->a_idx_ld : 0, gather ? ex->a_ns : 0,
                                                      gather ? ex->a_zero : nullptr, nullptr, 0.0, 0.f, 1.f, pa, h2_sa, h2_sb);
#define GO(BMV, BNV, MINB)                                                                                              \
    do {                                                                                                                \
        if (h2)                                                                                                         \
            return launch_x6<BMV, BNV, MINB, 3, 0, 0, 0, 0, 1>(grid, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias,  \
                                                      k_per_split, vec_a, vec_b, atomic_out, colp, colp_chunks,             \
                                                      gather ? ex->a_idx : nullptr, gather ? ex->a_idx_ld : 0,              \
                                                      gather ? ex->a_ns : 0, gather ? ex->a_zero : nullptr, nullptr, 0.0,   \
                                                      0.f, 1.f, pa, h2_sa, h2_sb);                                          \
        if (a_bf16)                                                                                                     \
            return launch_x6<BMV, BNV, MINB, 1, 0, 0>(grid, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias,           \
                                                      k_per_split, vec_a, vec_b, atomic_out, colp, colp_chunks, nullptr, 0, \
                                                      0, nullptr, nullptr, 0.0, 0.f, 1.f, pa);                                \
        return launch_x6<BMV, BNV, MINB, 3, 0, 0>(grid, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias, k_per_split,  \
                                                  vec_a, vec_b, atomic_out, colp, colp_chunks, gather ? ex->a_idx : nullptr,\
                                                  gather ? ex->a_idx_ld : 0, gather ? ex->a_ns : 0,                         \
                                                  gather ? ex->a_zero : nullptr, nullptr, 0.0, 0.f, 1.f, pa);               \
    } while (0)
    if (anorm && h2)
        return launch_x6<64, 64, 4, 3, 0, 0, 1, 0, 1>(grid, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias, k_per_split, vec_a,
                                                      vec_b, atomic_out, colp, colp_chunks, gather ? ex->a_idx : nullptr,
                                                      gather ? ex->a_idx_ld : 0, gather ? ex->a_ns : 0,
                                                      gather ? ex->a_zero : nullptr, ex->a_sums, ex->a_count, ex->a_eps, ex->a_slope,
                                                      pa);
    if (anorm)
        return launch_x6<64, 64, 4, 3, 0, 0, 1>(grid, st, a, lda, b, ldb, c, ldc, m, n, k, row_scale, bias, k_per_split, vec_a,
                                                vec_b, atomic_out, colp, colp_chunks, gather ? ex->a_idx : nullptr,
                                                gather ? ex->a_idx_ld : 0, gather ? ex->a_ns : 0,
                                                gather ? ex->a_zero : nullptr, ex->a_sums, ex->a_count, ex->a_eps, ex->a_slope,
                                                pa);
    if (pick == 0) { GO(128, 128, 2); }
    if (pick == 1) { GO(128, 64, 2); }
    if (pick == 2) { GO(64, 128, 2); }
    GO(64, 64, 4);
#undef GO
}

}  // namespace pcrcg
