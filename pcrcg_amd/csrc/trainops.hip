// trainops.hip -- kernels of the training-side rows: loss labels and backward passes.
// ABI: include/pcrcg_train.h.
#include <map>
#include <mutex>

#include "common.h"
#include "pcrcg_train.h"

namespace pcrcg {
namespace {

// ---- feature arg-max (ref:lib/loss.py:209-213) ---------------------------------------------------
// One thread owns a row of A (C floats in registers); rows of B stream through LDS in tiles of TB rows
// and are read back as wave-uniform (broadcast) float4s, so the inner loop is pure FMA.  The overlap region
// has only a few thousand rows, i.e. a few dozen 256-row blocks, so the columns are split over grid.y and
// the partial winners are merged with a 64-bit atomicMax on (orderable score bits, ~column): the larger
// score wins, equal scores keep the smaller column.
__device__ inline unsigned long long pack_best(float v, int j) {
    const unsigned int b = __float_as_uint(v);
    const unsigned int key = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)j);
}

template <int C>
__global__ void __launch_bounds__(256) k_feature_argmax(const float* __restrict__ a, int lda, int n,
                                                         const float* __restrict__ b, int ldb, int m, int cols_per,
                                                         unsigned long long* __restrict__ packed) {
    constexpr int TB = 128;
    __shared__ __attribute__((aligned(16))) float bs[TB * C];
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int r = row < n ? row : n - 1;
    float av[C];
#pragma unroll
    for (int k = 0; k < C; ++k) av[k] = a[(long)r * lda + k];
    float bv = -INFINITY;
    int bj = 0;
    const int jbeg = blockIdx.y * cols_per, jend = min(m, jbeg + cols_per);
    for (int j0 = jbeg; j0 < jend; j0 += TB) {
        const int tj = min(TB, jend - j0);
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
    if (row < n && jbeg < jend) atomicMax(&packed[row], pack_best(bv, bj));
}

// 32-wide descriptors on the fp32 matrix cores (round 5): a wavefront keeps 32 rows of A as its MFMA operand (lane (row,
// half) holds A[row][16 half + s]) and walks the columns of its range 32 at a time -- lane (column, half) loads the same 16
// entries of its B row as four float4 --, 16 v_mfma_f32_32x32x2_f32 per 32 x 32 block of scores; every lane keeps the best
// score and column of its 16 rows over the columns it sees (= those congruent to its lane index), strictly-greater, so the
// smaller column survives a tie; the 32 lanes of a row meet by shuffles at the end, and the column ranges (grid.y) through
// the same 64-bit atomicMax as the VALU kernel.  fp32 operands, fp32 accumulation: the reference's torch.matmul arithmetic
// up to summation order.
typedef float fa_f16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) k_feature_argmax_mfma32(const float* __restrict__ a, int lda, int n,
                                                                const float* __restrict__ b, int ldb, int m, int cols_per,
                                                                unsigned long long* __restrict__ packed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int row0 = (blockIdx.x * 4 + wave) * 32;
    if (row0 >= n) return;
    float av[16];
    {
        const float* ap = a + (long)min(row0 + l31, n - 1) * lda + 16 * half;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(ap + 4 * q);
            av[4 * q] = t.x; av[4 * q + 1] = t.y; av[4 * q + 2] = t.z; av[4 * q + 3] = t.w;
        }
    }
    float best[16];
    int bj[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { best[r] = -INFINITY; bj[r] = 0; }
    const int jbeg = blockIdx.y * cols_per, jend = min(m, jbeg + cols_per);
    auto load_b = [&](int j0, float4 (&t)[4]) {
        const float* bp = b + (long)min(j0 + l31, m - 1) * ldb + 16 * half;
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = *reinterpret_cast<const float4*>(bp + 4 * q);
    };
    float4 nxt[4];
    if (jbeg < jend) load_b(jbeg, nxt);
    for (int j0 = jbeg; j0 < jend; j0 += 32) {
        const int col = j0 + l31;
        float bv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) { bv[4 * q] = nxt[q].x; bv[4 * q + 1] = nxt[q].y; bv[4 * q + 2] = nxt[q].z; bv[4 * q + 3] = nxt[q].w; }
        if (j0 + 32 < jend) load_b(j0 + 32, nxt);           // the next block's rows are in flight behind this block's MFMAs
        fa_f16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bv[s2], acc, 0, 0, 0);
        if (col < jend) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (acc[r] > best[r]) { best[r] = acc[r]; bj[r] = col; }
        }
    }
    if (jbeg >= jend) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned long long p = pack_best(best[r], bj[r]);         // (-inf, 0) from a lane that saw no column loses to any score
#pragma unroll
        for (int sh = 16; sh >= 1; sh >>= 1) {
            const unsigned long long o = __shfl_xor(p, sh, 64);
            p = o > p ? o : p;
        }
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (l31 == 0 && row < n) atomicMax(&packed[row], p);
    }
}

// any width: A rows are re-read from memory (L1/L2 resident), one thread per row
__global__ void __launch_bounds__(256) k_feature_argmax_any(const float* __restrict__ a, int lda, int n,
                                                             const float* __restrict__ b, int ldb, int m, int c,
                                                             int cols_per, unsigned long long* __restrict__ packed) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    float bv = -INFINITY;
    int bj = 0;
    const int jbeg = blockIdx.y * cols_per, jend = min(m, jbeg + cols_per);
    for (int j = jbeg; j < jend; ++j) {
        float s = 0.f;
        for (int k = 0; k < c; ++k) s = fmaf(a[(long)row * lda + k], b[(long)j * ldb + k], s);
        if (s > bv) { bv = s; bj = j; }
    }
    if (jbeg < jend) atomicMax(&packed[row], pack_best(bv, bj));
}

__global__ void __launch_bounds__(256) k_feature_argmax_unpack(const unsigned long long* __restrict__ packed, int n,
                                                                long long* __restrict__ arg, float* __restrict__ best) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    const unsigned long long p = packed[row];
    arg[row] = (long long)(0xFFFFFFFFu - (unsigned int)(p & 0xFFFFFFFFull));
    if (best) {
        const unsigned int key = (unsigned int)(p >> 32);
        best[row] = __uint_as_float((key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key);
    }
}

// ---- deterministic accumulation (PCRCG_DEBUG=deterministic=1, include/pcrcg.h) ----------------------------------------
// The scatter kernels below add into a support row from many queries at once; with fp32 atomics the order of those adds --
// and so the last bits of the result -- changes from run to run.  Integer addition is associative: under the switch every
// contribution is rounded ONCE to 64-bit fixed point (scale 2^S, S chosen on the device from the largest magnitude of the
// source gradient so that 2^22 contributions cannot overflow: resolution 2^-39 of that magnitude, finer than fp32) and added
// with a 64-bit integer atomic into a zeroed scratch buffer of the destination's shape; k_fix_flush then adds the exact
// integer sums to the destination (one fp32 rounding per element) and leaves the scratch zeroed for the next use.  The
// scratch is the library's own (one buffer per stream, grown on demand): a debugging mode pays for its memory itself.
struct FixAcc {
    long long* acc;            // NULL: plain fp32 atomics (the default)
    const unsigned* maxbits;   // bit pattern of the largest |source gradient| (k_absmax_bits)
    int fan_log2;              // log2 of (contributions per element x the largest factor a contribution carries)
};
// An all-zero source gradient (maxbits == 0: e.g. an output nobody differentiated) has nothing to add: scale 0, every
// contribution rounds to the integer 0 and the flush skips its (still zero) sums.  Tiny gradients would ask for a scale
// beyond fp32's range: the exponent is clamped (the resolution is then coarser than 2^-39 of the magnitude but still far
// below fp32's).  (Round 5 returned +inf for both: 0 * inf = NaN into __float2ll_rn.)
__device__ __forceinline__ float fix_scale(const FixAcc& f) {
    const unsigned bits = *f.maxbits;
    if (bits == 0u) return 0.0f;
    const int e = (int)((bits >> 23) & 0xff) - 127;                // largest magnitude < 2^(e + 1)
    const int s = 62 - f.fan_log2 - (e + 1);
    return ldexpf(1.0f, s > 126 ? 126 : s);
}
template <bool DET>
__device__ __forceinline__ void scatter_add(float* dst, long off, float v, const FixAcc& f, float scale) {
    if constexpr (DET) atomicAdd(reinterpret_cast<unsigned long long*>(f.acc) + off, (unsigned long long)__float2ll_rn(v * scale));
    else atomicAdd(dst + off, v);
}
__global__ void __launch_bounds__(256) k_absmax_bits(const float* __restrict__ x, long rows, int cols, long ld, unsigned* __restrict__ out) {
    unsigned m = 0u;
    const long total = rows * cols;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long r = t / cols;
        const unsigned b = __float_as_uint(x[r * ld + (t - r * cols)]) & 0x7fffffffu;
        m = b > m && b < 0x7f800000u ? b : m;                      // (non-finite gradients poison the step anyway)
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const unsigned o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
__global__ void __launch_bounds__(256) k_fix_flush(long long* __restrict__ acc, float* __restrict__ dst, long n, FixAcc f) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const long long v = acc[t];
    if (v != 0) {                                                  // (a zero scale leaves only zero sums: never here)
        dst[t] += (float)((double)v / (double)fix_scale(f));
        acc[t] = 0;
    }
}

// ---- KPConv backward w.r.t. the input features ---------------------------------------------------
//   dx[idx[q,h], c] += sum_k w[q,h,k] * d_wf[q,k,c],   w as in kpconv.hip (rigid kernel, linear influence)
// One wavefront per (query, 64-channel chunk): lanes = channels hold the 15 rows d_wf[q,k,c0+lane] in
// registers; neighbours are taken four at a time -- lane (hsub, j) evaluates the influence of neighbour
// h0+hsub on kernel point j, exactly the forward's assignment -- and the 15 weights of each neighbour are
// broadcast through SGPRs (v_readlane).  The scatter uses hardware fp32 atomics: a support point is a
// neighbour of ~H queries, so its row receives ~H contributions in arbitrary order.
constexpr int K = PCRCG_KPOINTS;

template <bool DET>
__global__ void __launch_bounds__(256) k_kpconv_bwd_dx(const float* __restrict__ q_pts, int nq,
                                                        const float* __restrict__ s_pts, int ns,
                                                        const long long* __restrict__ idx, int H, int ld_idx,
                                                        const float* __restrict__ d_wf, int cin,
                                                        const float* __restrict__ kp, float extent,
                                                        float* __restrict__ dx, int nchunk, FixAcc fx) {
    const int lane = threadIdx.x & 63;
    const int hsub = lane >> 4, j = lane & 15;
    const float fscale = DET ? fix_scale(fx) : 1.0f;
    const long item = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (item >= (long)nq * nchunk) return;
    const int q = (int)(item / nchunk), chunk = (int)(item - (long)q * nchunk);
    const int cc = chunk * 64 + lane;
    const bool cok = cc < cin;
    const bool jvalid = j < K;
    const float kpx = jvalid ? kp[3 * j] : 0.f, kpy = jvalid ? kp[3 * j + 1] : 0.f, kpz = jvalid ? kp[3 * j + 2] : 0.f;
    const float inv_extent = 1.0f / extent;
    const float qx = q_pts[3 * (long)q], qy = q_pts[3 * (long)q + 1], qz = q_pts[3 * (long)q + 2];
    float g[K];
#pragma unroll
    for (int k = 0; k < K; ++k) g[k] = d_wf[((long)q * K + k) * cin + (cok ? cc : cin - 1)];
    for (int hc = 0; hc < H; hc += 64) {
        const int h = hc + lane;
        const long long iv = idx[(long)q * ld_idx + (h < H ? h : H - 1)];
        const int i = (h < H && iv >= 0 && iv < ns) ? (int)iv : -1;
        const long ic = i >= 0 ? i : 0;
        const float px = s_pts[3 * ic] - qx, py = s_pts[3 * ic + 1] - qy, pz = s_pts[3 * ic + 2] - qz;
        const int hn = H - hc < 64 ? H - hc : 64;
        for (int h0 = 0; h0 < hn; h0 += 4) {
            const int src = h0 + hsub;
            const int ii = __shfl(i, src, 64);
            const float nx = __shfl(px, src, 64), ny = __shfl(py, src, 64), nz = __shfl(pz, src, 64);
            float w = 0.f;
            if (ii >= 0 && src < hn && jvalid) {
                const float ddx = nx - kpx, ddy = ny - kpy, ddz = nz - kpz;
                w = fmaxf(1.0f - __builtin_amdgcn_sqrtf(ddx * ddx + ddy * ddy + ddz * ddz) * inv_extent, 0.0f);
            }
            const int wi = __float_as_int(w);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int is = __shfl(i, h0 + s, 64);              // wave-uniform
                if (h0 + s >= hn || is < 0) continue;
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k)
                    acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(wi, s * 16 + k)), g[k], acc);
                if (cok) scatter_add<DET>(dx, (long)is * cin + cc, acc, fx, fscale);
            }
        }
    }
}

// ---- pooling backward ---------------------------------------------------------------------------
// max_pool: the gradient of y[q,c] = max_h x[idx[q,h],c] goes to the FIRST neighbour attaining the maximum
// (shadow neighbours contribute the value 0 and swallow the gradient when they win, ref:models/blocks.py:95).
// The same scatter on the matrix cores (round 5).  Per query the contribution of its H neighbours is the small product
//   G[h, c] = sum_k w[h, k] d_wf[q, k, c]        (w: the H x 15 influence weights, ref:models/blocks.py:300-340)
// and row h of G is added to dx[idx[q, h], :].  One wavefront per (query, 64-channel chunk) takes 16 neighbours at a time
// through v_mfma_f32_16x16x4_f32: A = w (lane (j, hsub): neighbour j of the tile, kernel point 4 step + hsub), B = d_wf rows
// (lane (j, hsub): channel 16 t + j, the same kernel point; loaded once per query), D register r of lane (j, hsub) = neighbour
// 4 hsub + r, channel 16 t + j -- so one atomic instruction covers four neighbours' 64-byte runs.  fp32 in, fp32 out: the
// products are the VALU kernel's up to summation order.
typedef float bw_f32x4 __attribute__((ext_vector_type(4)));
template <bool DET>
__global__ void __launch_bounds__(256) k_kpconv_bwd_dx_mfma(const float* __restrict__ q_pts, int nq,
                                                             const float* __restrict__ s_pts, int ns,
                                                             const long long* __restrict__ idx, int H, int ld_idx,
                                                             const float* __restrict__ d_wf, int cin,
                                                             const float* __restrict__ kp, float extent,
                                                             float* __restrict__ dx, int nchunk, FixAcc fx) {
    const int lane = threadIdx.x & 63;
    const int hsub = lane >> 4, j = lane & 15;
    const float fscale = DET ? fix_scale(fx) : 1.0f;
    const long item = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (item >= (long)nq * nchunk) return;
    const int q = (int)(item / nchunk), chunk = (int)(item - (long)q * nchunk);
    const int c0 = chunk * 64;
    const int nt = (cin - c0 + 15) / 16 < 4 ? (cin - c0 + 15) / 16 : 4;       // 16-channel groups of this chunk (wave-uniform)
    const float inv_extent = 1.0f / extent;
    const float qx = q_pts[3 * (long)q], qy = q_pts[3 * (long)q + 1], qz = q_pts[3 * (long)q + 2];
    float kx[4], ky[4], kz[4], d[4][4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int k = 4 * st + hsub;                    // this lane's kernel point of step st (15: the padding row)
        const int kc = k < K ? k : K - 1;
        kx[st] = kp[3 * kc]; ky[st] = kp[3 * kc + 1]; kz[st] = kp[3 * kc + 2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = c0 + 16 * t + j;
            const float v = d_wf[((long)q * K + kc) * cin + (c < cin ? c : cin - 1)];
            d[st][t] = (k < K && c < cin) ? v : 0.f;
        }
    }
    for (int hc = 0; hc < H; hc += 64) {
        // lanes = neighbours: index + centred coordinates, once per 64 neighbours
        const int h = hc + lane;
        const long long iv = idx[(long)q * ld_idx + (h < H ? h : H - 1)];
        const int i = (h < H && iv >= 0 && iv < ns) ? (int)iv : -1;
        const long ic = i >= 0 ? i : 0;
        const float px = s_pts[3 * ic] - qx, py = s_pts[3 * ic + 1] - qy, pz = s_pts[3 * ic + 2] - qz;
        const int hn = H - hc < 64 ? H - hc : 64;
        for (int h0 = 0; h0 < hn; h0 += 16) {
            const int ii = __shfl(i, h0 + j, 64);
            const float nx = __shfl(px, h0 + j, 64), ny = __shfl(py, h0 + j, 64), nz = __shfl(pz, h0 + j, 64);
            if (__ballot(ii >= 0) == 0) continue;
            float w[4];
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const float ddx = nx - kx[st], ddy = ny - ky[st], ddz = nz - kz[st];
                const float wv = fmaxf(1.0f - __builtin_amdgcn_sqrtf(ddx * ddx + ddy * ddy + ddz * ddz) * inv_extent, 0.0f);
                w[st] = (ii >= 0 && 4 * st + hsub < K) ? wv : 0.f;
            }
            bw_f32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = (bw_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t >= nt) break;
#pragma unroll
                for (int st = 0; st < 4; ++st) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[st], d[st][t], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ir = __shfl(i, h0 + 4 * hsub + r, 64);
                if (ir < 0) continue;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int c = c0 + 16 * t + j;
                    if (t < nt && c < cin) scatter_add<DET>(dx, (long)ir * cin + c, acc[t][r], fx, fscale);
                }
            }
        }
    }
}
template <bool DET>
__global__ void __launch_bounds__(256) k_gather_max_bwd(const float* __restrict__ x, int ns, int c,
                                                         const long long* __restrict__ idx, int nq, int h, int ld_idx,
                                                         const float* __restrict__ y, const float* __restrict__ dy,
                                                         float* __restrict__ dx, int nchunk, FixAcc fx) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (item >= (long)nq * nchunk) return;
    const int q = (int)(item / nchunk), chunk = (int)(item - (long)q * nchunk);
    const int cc = chunk * 64 + lane;
    if (cc >= c) return;
    const float fscale = DET ? fix_scale(fx) : 1.0f;
    const float m = y[(long)q * c + cc], gq = dy[(long)q * c + cc];
    const long long* row = idx + (long)q * ld_idx;
    for (int jn = 0; jn < h; ++jn) {
        const long long i = row[jn];
        const bool real = i >= 0 && i < ns;
        const float v = real ? x[i * c + cc] : 0.f;
        if (v == m) {
            if (real) scatter_add<DET>(dx, i * c + cc, gq, fx, fscale);
            break;
        }
    }
}

// closest_pool: dx[idx[q,0], :] += dy[q, :]
// four channels per lane (c % 4 == 0): the neighbours' rows are read as float4, as the forward reads them
template <bool DET>
__global__ void __launch_bounds__(256) k_gather_max_bwd4(const float* __restrict__ x, int ns, int c,
                                                          const long long* __restrict__ idx, int nq, int h, int ld_idx,
                                                          const float* __restrict__ y, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int nchunk, FixAcc fx) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (item >= (long)nq * nchunk) return;
    const int q = (int)(item / nchunk), chunk = (int)(item - (long)q * nchunk);
    const int cb = chunk * 64 + lane;
    if (cb >= (c >> 2)) return;
    const float fscale = DET ? fix_scale(fx) : 1.0f;
    const float4 m = reinterpret_cast<const float4*>(y + (long)q * c)[cb], gq = reinterpret_cast<const float4*>(dy + (long)q * c)[cb];
    const long long* row = idx + (long)q * ld_idx;
    // where each channel meets its maximum FIRST (that row takes the gradient); the adds are issued after the walk, whole
    // wavefront at once: an atomic inside the walk would go out once per neighbour with a handful of lanes active
    int open = 15;
    long long at[4] = {-1, -1, -1, -1};
    const float me[4] = {m.x, m.y, m.z, m.w};
    for (int j0 = 0; j0 < h && open; j0 += 4) {            // four rows in flight (clamped, branch-free loads), then in order
        long long iv[4];
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) iv[u] = row[j0 + u < h ? j0 + u : h - 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool real = iv[u] >= 0 && iv[u] < ns;
            const float4 t = reinterpret_cast<const float4*>(x + (real ? iv[u] : 0) * c)[cb];
            v[u] = real ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            if (!real) iv[u] = -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j0 + u >= h) break;
            const float ve[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((open >> k & 1) && ve[k] == me[k]) {
                    open &= ~(1 << k);
                    at[k] = iv[u];
                }
        }
    }
    const float ge[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (at[k] >= 0) scatter_add<DET>(dx, at[k] * c + 4 * cb + k, ge[k], fx, fscale);
}
template <bool DET>
__global__ void __launch_bounds__(256) k_gather_first_bwd(const float* __restrict__ dy, int ld_dy, int c,
                                                           const long long* __restrict__ idx, int nq, int ld_idx, int ns,
                                                           float* __restrict__ dx, FixAcc fx) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (q >= nq) return;
    const long long i = idx[(long)q * ld_idx];
    if (i < 0 || i >= ns) return;
    const float fscale = DET ? fix_scale(fx) : 1.0f;
    for (int cc = lane; cc < c; cc += 64) scatter_add<DET>(dx, i * c + cc, dy[(long)q * ld_dy + cc], fx, fscale);
}

// ---- InstanceNorm (+ LeakyReLU) backward ----------------------------------------------------------
//   xhat = (x - mean) * rstd,  y = lrelu(xhat),  g = dy * (xhat > 0 ? 1 : slope)
//   dx = rstd * (g - mean_n(g) - xhat * mean_n(g * xhat))
// Column sums in fp64, two deterministic stages (layout [2][c][chunks], as the forward statistics).
constexpr int kBwdChunks = 128;

// SUMS: the workgroups' partial sums meet in a ZEROED [2][c] buffer by fp64 atomics instead of being stored per chunk (few
// row chunks: the coarse levels and the GNN -- no finishing launch; not under deterministic=1)
template <bool SUMS>
__global__ void __launch_bounds__(256) k_in_bwd_partial(const float* __restrict__ x, int n, int c, int ldx,
                                                         const float* __restrict__ stats, const float* __restrict__ dy,
                                                         int ld_dy, float slope, double* __restrict__ partial) {
    __shared__ double s_a[4][64], s_b[4][64];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int ch = blockIdx.y * 64 + lane;
    const int chunk = blockIdx.x, nchunks = gridDim.x;
    const long rows_per = ((long)n + nchunks - 1) / nchunks;
    const long r0 = chunk * rows_per, r1 = min((long)n, r0 + rows_per);
    double a = 0.0, b = 0.0;
    if (ch < c) {
        const float mean = stats[2 * ch], rstd = stats[2 * ch + 1];
        long r = r0 + rl;
        for (; r + 12 < r1; r += 16) {   // four independent row pairs in flight per thread
            float xv[4], gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xv[u] = x[(r + 4 * u) * ldx + ch];
                gv[u] = dy[(r + 4 * u) * ld_dy + ch];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float xh = (xv[u] - mean) * rstd;
                const float g = gv[u] * (xh > 0.f ? 1.0f : slope);
                a += (double)g;
                b += (double)g * (double)xh;
            }
        }
        for (; r < r1; r += 4) {
            const float xh = (x[r * ldx + ch] - mean) * rstd;
            const float g = dy[r * ld_dy + ch] * (xh > 0.f ? 1.0f : slope);
            a += (double)g;
            b += (double)g * (double)xh;
        }
    }
    s_a[rl][lane] = a;
    s_b[rl][lane] = b;
    __syncthreads();
    if (rl == 0 && ch < c) {
        const double ta = (s_a[0][lane] + s_a[1][lane]) + (s_a[2][lane] + s_a[3][lane]);
        const double tb = (s_b[0][lane] + s_b[1][lane]) + (s_b[2][lane] + s_b[3][lane]);
        if (SUMS) {
            atomicAdd(&partial[ch], ta);
            atomicAdd(&partial[c + ch], tb);
        } else {
            partial[(long)ch * nchunks + chunk] = ta;
            partial[((long)c + ch) * nchunks + chunk] = tb;
        }
    }
}

__global__ void __launch_bounds__(256) k_in_bwd_final(const double* __restrict__ partial, int nchunks, int c, double count,
                                                       float* __restrict__ means /* [2c]: mean(g), mean(g*xhat) */) {
    const int lane = threadIdx.x & 63;
    const int ch = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ch >= c) return;
    double a = 0.0, b = 0.0;
    for (int k = lane; k < nchunks; k += 64) {
        a += partial[(long)ch * nchunks + k];
        b += partial[((long)c + ch) * nchunks + k];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d, 64);
        b += __shfl_xor(b, d, 64);
    }
    if (lane == 0) {
        means[2 * ch] = (float)(a / count);
        means[2 * ch + 1] = (float)(b / count);
    }
}

__global__ void __launch_bounds__(256) k_in_bwd_apply(const float* __restrict__ x, long total, int c, int ldx,
                                                       const float* __restrict__ stats, const float* __restrict__ dy,
                                                       int ld_dy, float slope, const float* __restrict__ means,
                                                       float* __restrict__ dx, int ld_dx) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long r = t / c;
    const int ch = (int)(t - r * c);
    const float rstd = stats[2 * ch + 1];
    const float xh = (x[r * ldx + ch] - stats[2 * ch]) * rstd;
    const float g = dy[r * ld_dy + ch] * (xh > 0.f ? 1.0f : slope);
    dx[r * ld_dx + ch] = rstd * (g - means[2 * ch] - xh * means[2 * ch + 1]);
}

// float4 variant: c, ldx, ld_dy, ld_dx multiples of 4 and 16-byte aligned bases
__global__ void __launch_bounds__(256) k_in_bwd_apply4(const float* __restrict__ x, long total4, int c4, int ldx,
                                                        const float* __restrict__ stats, const float* __restrict__ dy,
                                                        int ld_dy, float slope, const float* __restrict__ means,
                                                        float* __restrict__ dx, int ld_dx) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const long r = t / c4;
    const int q = (int)(t - r * c4);
    const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + 4 * q);
    const float4 gv = *reinterpret_cast<const float4*>(dy + r * ld_dy + 4 * q);
    const float4 s0 = *reinterpret_cast<const float4*>(stats + 8 * q), s1 = *reinterpret_cast<const float4*>(stats + 8 * q + 4);
    const float4 m0 = *reinterpret_cast<const float4*>(means + 8 * q), m1 = *reinterpret_cast<const float4*>(means + 8 * q + 4);
    auto one = [&](float xe, float ge, float mean, float rstd, float mg, float mgx) {
        const float xh = (xe - mean) * rstd;
        const float g = ge * (xh > 0.f ? 1.0f : slope);
        return rstd * (g - mg - xh * mgx);
    };
    const float4 o = make_float4(one(xv.x, gv.x, s0.x, s0.y, m0.x, m0.y), one(xv.y, gv.y, s0.z, s0.w, m0.z, m0.w),
                                 one(xv.z, gv.z, s1.x, s1.y, m1.x, m1.y), one(xv.w, gv.w, s1.z, s1.w, m1.z, m1.w));
    *reinterpret_cast<float4*>(dx + r * ld_dx + 4 * q) = o;
}

// k_in_bwd_apply4 with mean(g), mean(g xhat) taken from the [2][c] fp64 SUMS of k_in_bwd_partial<true>
__global__ void __launch_bounds__(256) k_in_bwd_apply4_sums(const float* __restrict__ x, long total4, int c4, int ldx,
                                                             const float* __restrict__ stats, const float* __restrict__ dy,
                                                             int ld_dy, float slope, const double* __restrict__ sums,
                                                             double inv_count, float* __restrict__ dx, int ld_dx) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const long r = t / c4;
    const int q = (int)(t - r * c4), c = 4 * c4;
    const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + 4 * q);
    const float4 gv = *reinterpret_cast<const float4*>(dy + r * ld_dy + 4 * q);
    const float4 s0 = *reinterpret_cast<const float4*>(stats + 8 * q), s1 = *reinterpret_cast<const float4*>(stats + 8 * q + 4);
    float mg[4], mgx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        mg[i] = (float)(sums[4 * q + i] * inv_count);
        mgx[i] = (float)(sums[c + 4 * q + i] * inv_count);
    }
    auto one = [&](float xe, float ge, float mean, float rstd, float a, float b) {
        const float xh = (xe - mean) * rstd;
        const float g = ge * (xh > 0.f ? 1.0f : slope);
        return rstd * (g - a - xh * b);
    };
    const float4 o = make_float4(one(xv.x, gv.x, s0.x, s0.y, mg[0], mgx[0]), one(xv.y, gv.y, s0.z, s0.w, mg[1], mgx[1]),
                                 one(xv.z, gv.z, s1.x, s1.y, mg[2], mgx[2]), one(xv.w, gv.w, s1.z, s1.w, mg[3], mgx[3]));
    *reinterpret_cast<float4*>(dx + r * ld_dx + 4 * q) = o;
}

// ---- softmax backward: ds = scale * p * (dp - sum_j p*dp), one wavefront per row -----------------
__global__ void __launch_bounds__(256) k_softmax_bwd(const float* __restrict__ p, int ld_p, const float* __restrict__ dp,
                                                      int ld_dp, int rows, int cols, float scale, float* __restrict__ ds,
                                                      int ld_ds) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float dot = 0.f;
    for (int jn = lane; jn < cols; jn += 64) dot = fmaf(p[(long)r * ld_p + jn], dp[(long)r * ld_dp + jn], dot);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) dot += __shfl_xor(dot, d, 64);
    for (int jn = lane; jn < cols; jn += 64)
        ds[(long)r * ld_ds + jn] = scale * p[(long)r * ld_p + jn] * (dp[(long)r * ld_dp + jn] - dot);
}

// ---- DGCNN edge conv backward (ref:models/gcn.py:37-64,121-129) ----------------------------------
//   e[i,j,c] = ctr[i,c] + nbr[idx[i,j],c];  n = (e - mean) * rstd over all N*k edges;  y[i,c] = lrelu(max_j n[i,j,c])
// With dn[i,c] = dy[i,c] * lrelu'(n_max[i,c]), S1[c] = sum_i dn, S2[c] = sum_i dn * n_max and E = N*k:
//   de[i,j,c] = rstd * ([j == j*] dn[i,c] - S1/E - n[i,j,c] * S2/E)
//   dctr[i,c] = sum_j de[i,j,c],   dnbr[s,c] = sum_{(i,j): idx[i,j] = s} de[i,j,c]        (fp32 atomics)
// Stage 1 reduces S1, S2 (fp64 atomics into [2][c]; a few hundred points), stage 2 applies.
__global__ void __launch_bounds__(256) k_edge_bwd_sums(const float* __restrict__ ctr, const float* __restrict__ nbr,
                                                        const int* __restrict__ idx, int n, int k, int c,
                                                        const float* __restrict__ stats, const float* __restrict__ dy,
                                                        float slope, double* __restrict__ sums) {
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int ch = blockIdx.y * 64 + lane;
    if (ch >= c) return;
    const float mean = stats[2 * ch], rstd = stats[2 * ch + 1];
    double a = 0.0, b = 0.0;
    for (int r = blockIdx.x * 4 + rl; r < n; r += gridDim.x * 4) {
        const float q = ctr[(long)r * c + ch];
        float m = 0.f;
        for (int j = 0; j < k; ++j) {
            const float v = q + nbr[(long)idx[(long)r * k + j] * c + ch];
            m = j == 0 ? v : fmaxf(m, v);
        }
        const float nm = (m - mean) * rstd;
        const float dn = dy[(long)r * c + ch] * (nm > 0.f ? 1.0f : slope);
        a += (double)dn;
        b += (double)dn * (double)nm;
    }
    atomicAdd(&sums[ch], a);
    atomicAdd(&sums[c + ch], b);
}

template <bool DET>
__global__ void __launch_bounds__(256) k_edge_bwd_apply(const float* __restrict__ ctr, const float* __restrict__ nbr,
                                                         const int* __restrict__ idx, int n, int k, int c,
                                                         const float* __restrict__ stats, const float* __restrict__ dy,
                                                         float slope, const double* __restrict__ sums,
                                                         float* __restrict__ dctr, float* __restrict__ dnbr, FixAcc fx) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ch = blockIdx.y * 64 + lane;
    if (r >= n || ch >= c) return;
    const float mean = stats[2 * ch], rstd = stats[2 * ch + 1];
    const double inv_e = 1.0 / ((double)n * (double)k);
    const float s1 = (float)(sums[ch] * inv_e), s2 = (float)(sums[c + ch] * inv_e);
    const float q = ctr[(long)r * c + ch];
    float m = 0.f;
    int jstar = 0;
    for (int j = 0; j < k; ++j) {
        const float v = q + nbr[(long)idx[(long)r * k + j] * c + ch];
        if (j == 0 || v > m) { m = v; jstar = j; }        // first maximum wins
    }
    const float nm = (m - mean) * rstd;
    const float dn = dy[(long)r * c + ch] * (nm > 0.f ? 1.0f : slope);
    const float fscale = DET ? fix_scale(fx) : 1.0f;
    float acc = 0.f;
    for (int j = 0; j < k; ++j) {
        const int s = idx[(long)r * k + j];
        const float nv = (q + nbr[(long)s * c + ch] - mean) * rstd;
        const float de = rstd * ((j == jstar ? dn : 0.f) - s1 - nv * s2);
        acc += de;
        scatter_add<DET>(dnbr, (long)s * c + ch, de, fx, fscale);
    }
    dctr[(long)r * c + ch] = acc;
}

}  // namespace
}  // namespace pcrcg

using namespace pcrcg;

// ---- the deterministic mode's scratch (see FixAcc): one zeroed 64-bit buffer + one word per stream, grown on demand ----
namespace {
struct DetScratch { long long* acc = nullptr; size_t elems = 0; unsigned* word = nullptr; };
std::mutex g_det_lock;
std::map<hipStream_t, DetScratch> g_det;
// -> a FixAcc over a zeroed buffer of at least `elems` elements whose scale follows the largest |src| value; NULL acc on failure
int det_begin(hipStream_t st, size_t elems, const float* src, long rows, int cols, long ld, int fan_log2, FixAcc* out) {
    DetScratch d;
    {
        std::lock_guard<std::mutex> g(g_det_lock);
        DetScratch& slot = g_det[st];
        if (slot.elems < elems) {
            // (a debugging mode: synchronous allocation; the old buffer is idle once the stream has drained)
            PCRCG_CHECK_HIP(hipStreamSynchronize(st));
            if (slot.acc) (void)hipFree(slot.acc);
            const size_t want = elems + elems / 4;
            PCRCG_CHECK_HIP(hipMalloc(&slot.acc, want * sizeof(long long)));
            PCRCG_CHECK_HIP(hipMemset(slot.acc, 0, want * sizeof(long long)));
            slot.elems = want;
        }
        if (!slot.word) PCRCG_CHECK_HIP(hipMalloc(&slot.word, 256));
        d = slot;
    }
    PCRCG_CHECK_HIP(hipMemsetAsync(d.word, 0, sizeof(unsigned), st));
    const long total = rows * cols;
    long blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (total > 0) hipLaunchKernelGGL(k_absmax_bits, dim3((unsigned)blocks), dim3(256), 0, st, src, rows, cols, ld, d.word);
    PCRCG_CHECK_LAUNCH();
    out->acc = d.acc;
    out->maxbits = d.word;
    out->fan_log2 = fan_log2;
    return PCRCG_OK;
}
}  // namespace
namespace pcrcg {
// pcrcg_debug_release(): the deterministic mode's fixed-point scratch of every stream (the caller has drained them)
void trainops_release_det() {
    std::lock_guard<std::mutex> g(g_det_lock);
    for (auto& kv : g_det) {
        if (kv.second.acc) (void)hipFree(kv.second.acc);
        if (kv.second.word) (void)hipFree(kv.second.word);
    }
    g_det.clear();
}
}  // namespace pcrcg
namespace {
int det_end(hipStream_t st, const FixAcc& fx, float* dst, size_t elems) {
    hipLaunchKernelGGL(k_fix_flush, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, fx.acc, dst, (long)elems, fx);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
const FixAcc kNoFix{nullptr, nullptr, 0};
}  // namespace

extern "C" size_t pcrcg_feature_argmax_ws_bytes(int n) { return carve_bytes((size_t)(n > 0 ? n : 1), 8); }

extern "C" int pcrcg_feature_argmax(const float* a, int lda, int n, const float* b, int ldb, int m, int c,
                                    int64_t* arg, float* best, void* ws, size_t ws_bytes, void* stream) {
    PCRCG_CHECK_ARG(n >= 0 && m >= 1 && c >= 1 && lda >= c && ldb >= c);
    if (n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(a && b && arg && ws);
    Carver cv(ws, ws_bytes);
    unsigned long long* packed = cv.take<unsigned long long>((size_t)n);
    PCRCG_CHECK_WS(cv);
    hipStream_t st = as_stream(stream);
    PCRCG_CHECK_HIP(hipMemsetAsync(packed, 0, (size_t)n * 8, st));
    const int gx = (n + 255) / 256;
    if (c == 32 && lda % 4 == 0 && ldb % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0 &&
        debug_opts().bwd_mfma) {
        // the matrix-core kernel: 128 rows per workgroup, column ranges so that ~1k workgroups are in flight
        const int gxm = (n + 127) / 128;
        int splits = (1024 + gxm - 1) / gxm;
        const int max_splits = (m + 255) / 256;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        const int cols_per = ((m + splits - 1) / splits + 31) / 32 * 32;
        hipLaunchKernelGGL(k_feature_argmax_mfma32, dim3(gxm, (m + cols_per - 1) / cols_per), dim3(256), 0, st, a, lda, n, b, ldb, m,
                           cols_per, packed);
        hipLaunchKernelGGL(k_feature_argmax_unpack, dim3(gx), dim3(256), 0, st, packed, n, reinterpret_cast<long long*>(arg), best);
        PCRCG_CHECK_LAUNCH();
        return PCRCG_OK;
    }
    int splits = (2048 + gx - 1) / gx;                      // ~2k blocks in flight
    const int max_splits = (m + 127) / 128;                 // at least one LDS tile of columns per block
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int cols_per = ((m + splits - 1) / splits + 127) / 128 * 128;
    const dim3 grid(gx, (m + cols_per - 1) / cols_per);
    if (c == 32) hipLaunchKernelGGL(k_feature_argmax<32>, grid, dim3(256), 0, st, a, lda, n, b, ldb, m, cols_per, packed);
    else if (c == 64) hipLaunchKernelGGL(k_feature_argmax<64>, grid, dim3(256), 0, st, a, lda, n, b, ldb, m, cols_per, packed);
    else hipLaunchKernelGGL(k_feature_argmax_any, grid, dim3(256), 0, st, a, lda, n, b, ldb, m, c, cols_per, packed);
    hipLaunchKernelGGL(k_feature_argmax_unpack, dim3(gx), dim3(256), 0, st, packed, n, reinterpret_cast<long long*>(arg),
                       best);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

extern "C" int pcrcg_kpconv_backward_dx(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx, int h,
                                        int ld_idx, const float* d_wf, int cin, const float* kp, float extent, float* dx,
                                        void* stream) {
    PCRCG_CHECK_ARG(nq >= 0 && ns >= 1 && h >= 1 && ld_idx >= h && cin >= 1 && extent > 0.0f);
    if (nq == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(q_pts && s_pts && idx && d_wf && kp && dx);
    const int nchunk = (cin + 63) / 64;
    const long items = (long)nq * nchunk;
    if (debug_opts().deterministic) {       // |contribution| <= 15 max|d_wf| (influence weights <= 1), at most nq of them per element
        FixAcc fx;
        PCRCG_PROPAGATE(det_begin(as_stream(stream), (size_t)ns * cin, d_wf, nq, PCRCG_KPOINTS * cin, (long)PCRCG_KPOINTS * cin, 22, &fx));
        if (debug_opts().bwd_mfma)
            hipLaunchKernelGGL(k_kpconv_bwd_dx_mfma<true>, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, as_stream(stream), q_pts,
                               nq, s_pts, ns, reinterpret_cast<const long long*>(idx), h, ld_idx, d_wf, cin, kp, extent, dx, nchunk, fx);
        else
            hipLaunchKernelGGL(k_kpconv_bwd_dx<true>, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, as_stream(stream), q_pts, nq,
                               s_pts, ns, reinterpret_cast<const long long*>(idx), h, ld_idx, d_wf, cin, kp, extent, dx, nchunk, fx);
        PCRCG_CHECK_LAUNCH();
        return det_end(as_stream(stream), fx, dx, (size_t)ns * cin);
    }
    if (debug_opts().bwd_mfma)
        hipLaunchKernelGGL(k_kpconv_bwd_dx_mfma<false>, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, as_stream(stream), q_pts, nq,
                           s_pts, ns, reinterpret_cast<const long long*>(idx), h, ld_idx, d_wf, cin, kp, extent, dx, nchunk, kNoFix);
    else
        hipLaunchKernelGGL(k_kpconv_bwd_dx<false>, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, as_stream(stream), q_pts, nq,
                           s_pts, ns, reinterpret_cast<const long long*>(idx), h, ld_idx, d_wf, cin, kp, extent, dx, nchunk, kNoFix);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

extern "C" int pcrcg_gather_max_backward(const float* x, int ns, int c, const int64_t* idx, int nq, int h, int ld_idx,
                                         const float* y, const float* dy, float* dx, void* stream) {
    PCRCG_CHECK_ARG(ns >= 0 && c >= 1 && nq >= 0 && h >= 1 && ld_idx >= h);
    if (nq == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(x && idx && y && dy && dx);
    const bool vec = c % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
    const int nchunk = vec ? (c / 4 + 63) / 64 : (c + 63) / 64;
    const long items = (long)nq * nchunk;
    const dim3 grid((unsigned)((items + 3) / 4));
    const long long* ix = reinterpret_cast<const long long*>(idx);
    if (debug_opts().deterministic) {
        FixAcc fx;
        PCRCG_PROPAGATE(det_begin(as_stream(stream), (size_t)ns * c, dy, nq, c, c, 20, &fx));
        if (vec) hipLaunchKernelGGL(k_gather_max_bwd4<true>, grid, dim3(256), 0, as_stream(stream), x, ns, c, ix, nq, h, ld_idx, y, dy, dx, nchunk, fx);
        else hipLaunchKernelGGL(k_gather_max_bwd<true>, grid, dim3(256), 0, as_stream(stream), x, ns, c, ix, nq, h, ld_idx, y, dy, dx, nchunk, fx);
        PCRCG_CHECK_LAUNCH();
        return det_end(as_stream(stream), fx, dx, (size_t)ns * c);
    }
    if (vec) hipLaunchKernelGGL(k_gather_max_bwd4<false>, grid, dim3(256), 0, as_stream(stream), x, ns, c, ix, nq, h, ld_idx, y, dy, dx, nchunk, kNoFix);
    else hipLaunchKernelGGL(k_gather_max_bwd<false>, grid, dim3(256), 0, as_stream(stream), x, ns, c, ix, nq, h, ld_idx, y, dy, dx, nchunk, kNoFix);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

extern "C" int pcrcg_gather_first_backward(const float* dy, int ld_dy, int c, const int64_t* idx, int nq, int ld_idx,
                                           int ns, float* dx, void* stream) {
    PCRCG_CHECK_ARG(c >= 1 && nq >= 0 && ld_idx >= 1 && ld_dy >= c && ns >= 0);
    if (nq == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(dy && idx && dx);
    if (debug_opts().deterministic) {
        FixAcc fx;
        PCRCG_PROPAGATE(det_begin(as_stream(stream), (size_t)ns * c, dy, nq, c, ld_dy, 20, &fx));
        hipLaunchKernelGGL(k_gather_first_bwd<true>, dim3((nq + 3) / 4), dim3(256), 0, as_stream(stream), dy, ld_dy, c,
                           reinterpret_cast<const long long*>(idx), nq, ld_idx, ns, dx, fx);
        PCRCG_CHECK_LAUNCH();
        return det_end(as_stream(stream), fx, dx, (size_t)ns * c);
    }
    hipLaunchKernelGGL(k_gather_first_bwd<false>, dim3((nq + 3) / 4), dim3(256), 0, as_stream(stream), dy, ld_dy, c,
                       reinterpret_cast<const long long*>(idx), nq, ld_idx, ns, dx, kNoFix);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

extern "C" size_t pcrcg_instnorm_backward_ws_bytes(int c) {
    return carve_bytes(2 * (size_t)(c > 0 ? c : 1) * kBwdChunks, sizeof(double)) + carve_bytes(2 * (size_t)(c > 0 ? c : 1), 4);
}

extern "C" int pcrcg_instnorm_backward(const float* x, int n, int c, int ldx, const float* stats, const float* dy,
                                       int ld_dy, float slope, float* dx, int ld_dx, void* ws, size_t ws_bytes,
                                       void* stream) {
    PCRCG_CHECK_ARG(n >= 0 && c >= 1 && ldx >= c && ld_dy >= c && ld_dx >= c);
    if (n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(x && stats && dy && dx && ws);
    Carver cv(ws, ws_bytes);
    double* partial = cv.take<double>(2 * (size_t)c * kBwdChunks);
    float* means = cv.take<float>(2 * (size_t)c);
    PCRCG_CHECK_WS(cv);
    hipStream_t st = as_stream(stream);
    // row chunks of >= 32 rows (8 per thread): the few-hundred-row tensors of the coarse levels are latency chains otherwise
    int chunks = (n + 31) / 32;
    if (chunks > kBwdChunks) chunks = kBwdChunks;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(k_in_bwd_partial<false>, dim3(chunks, (c + 63) / 64), dim3(256), 0, st, x, n, c, ldx, stats, dy, ld_dy,
                       slope, partial);
    hipLaunchKernelGGL(k_in_bwd_final, dim3((c + 3) / 4), dim3(256), 0, st, partial, chunks, c, (double)n, means);
    const long total = (long)n * c;
    const bool vec = (c % 4 == 0) && (ldx % 4 == 0) && (ld_dy % 4 == 0) && (ld_dx % 4 == 0) &&
                     (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) |
                        reinterpret_cast<uintptr_t>(dx)) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(k_in_bwd_apply4, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, x, total / 4,
                           c / 4, ldx, stats, dy, ld_dy, slope, means, dx, ld_dx);
    else
        hipLaunchKernelGGL(k_in_bwd_apply, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, total, c, ldx,
                           stats, dy, ld_dy, slope, means, dx, ld_dx);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

namespace pcrcg {
// InstanceNorm + LeakyReLU backward in TWO launches for tensors of few row chunks: `sums` is a ZEROED [2][c] fp64 buffer
// (the train tape keeps it in its gradient region, which one memset clears) that the statistics kernel's workgroups add to
bool instnorm_backward_sums_ok(const float* x, int n, int c, int ldx, const float* dy, int ld_dy, const float* dx, int ld_dx) {
    return !debug_opts().deterministic && n >= 1 && n <= 32 * 32 && c % 4 == 0 && ldx % 4 == 0 && ld_dy % 4 == 0 && ld_dx % 4 == 0 &&
           ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
}
int instnorm_backward_sums(const float* x, int n, int c, int ldx, const float* stats, const float* dy, int ld_dy, float slope,
                           float* dx, int ld_dx, double* sums, hipStream_t st) {
    PCRCG_CHECK_ARG(instnorm_backward_sums_ok(x, n, c, ldx, dy, ld_dy, dx, ld_dx) && ldx >= c && ld_dy >= c && ld_dx >= c);
    PCRCG_CHECK_ARG(x && stats && dy && dx && sums);
    const int chunks = (n + 31) / 32;
    hipLaunchKernelGGL(k_in_bwd_partial<true>, dim3(chunks, (c + 63) / 64), dim3(256), 0, st, x, n, c, ldx, stats, dy, ld_dy, slope,
                       sums);
    const long total4 = (long)n * (c / 4);
    hipLaunchKernelGGL(k_in_bwd_apply4_sums, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, x, total4, c / 4, ldx, stats,
                       dy, ld_dy, slope, sums, 1.0 / (double)n, dx, ld_dx);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
}  // namespace pcrcg

extern "C" int pcrcg_softmax_rows_backward(const float* p, int ld_p, const float* dp, int ld_dp, int rows, int cols,
                                           float scale, float* ds, int ld_ds, void* stream) {
    PCRCG_CHECK_ARG(rows >= 0 && cols >= 1 && ld_p >= cols && ld_dp >= cols && ld_ds >= cols);
    if (rows == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(p && dp && ds);
    hipLaunchKernelGGL(k_softmax_bwd, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), p, ld_p, dp, ld_dp, rows, cols,
                       scale, ds, ld_ds);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

extern "C" size_t pcrcg_edgeconv_backward_ws_bytes(int c) { return carve_bytes(2 * (size_t)(c > 0 ? c : 1), sizeof(double)); }

extern "C" int pcrcg_edgeconv_backward(const float* ctr, const float* nbr, const int* idx, int n, int k, int c,
                                       const float* stats, const float* dy, float slope, float* dctr, float* dnbr,
                                       void* ws, size_t ws_bytes, void* stream) {
    PCRCG_CHECK_ARG(n >= 0 && k >= 1 && c >= 1);
    if (n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(ctr && nbr && idx && stats && dy && dctr && dnbr && ws);
    Carver cv(ws, ws_bytes);
    double* sums = cv.take<double>(2 * (size_t)c);
    PCRCG_CHECK_WS(cv);
    hipStream_t st = as_stream(stream);
    PCRCG_CHECK_HIP(hipMemsetAsync(sums, 0, 2 * (size_t)c * sizeof(double), st));
    const bool det = debug_opts().deterministic != 0;
    int gx = (n + 3) / 4;
    if (gx > 64) gx = 64;
    if (det) gx = 1;      // one workgroup per channel block: every sum receives exactly one add
    hipLaunchKernelGGL(k_edge_bwd_sums, dim3(gx, (c + 63) / 64), dim3(256), 0, st, ctr, nbr, idx, n, k, c, stats, dy, slope,
                       sums);
    if (det) {
        // |de| <= rstd (|dn| + |S1/E| + |n| |S2/E|) <= 2^9 max|dy| (2 + 2^11) (rstd <= eps^-1/2, |n| <= sqrt(E), mean|n| <= 1),
        // at most n k <= 2^15 edges per neighbour row
        FixAcc fx;
        PCRCG_PROPAGATE(det_begin(st, (size_t)n * c, dy, n, c, c, 36, &fx));
        hipLaunchKernelGGL(k_edge_bwd_apply<true>, dim3((n + 3) / 4, (c + 63) / 64), dim3(256), 0, st, ctr, nbr, idx, n, k, c, stats,
                           dy, slope, sums, dctr, dnbr, fx);
        PCRCG_CHECK_LAUNCH();
        return det_end(st, fx, dnbr, (size_t)n * c);
    }
    hipLaunchKernelGGL(k_edge_bwd_apply<false>, dim3((n + 3) / 4, (c + 63) / 64), dim3(256), 0, st, ctr, nbr, idx, n, k, c, stats,
                       dy, slope, sums, dctr, dnbr, kNoFix);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

// ---- whole-op entry points (the exports SURVEY.md 8b recommends for a C caller) ---------------------
extern "C" size_t pcrcg_kpconv_forward_ws_bytes(int nq, int ns, int cin) {
    return pcrcg_kpconv_ws_bytes(ns) + carve_bytes((size_t)(nq > 0 ? nq : 1) * PCRCG_KPOINTS * (size_t)cin, 4) +
           carve_bytes((size_t)(nq > 0 ? nq : 1), 4);
}

// Layout of `ws` (kept by the caller until the backward call): [ aggregate scratch | wf [nq,15*cin] | inv_n [nq] ]
static void kpconv_ws_layout(void* ws, size_t ws_bytes, int nq, int ns, int cin, Carver* cv, void** agg, size_t* agg_bytes,
                             float** wf, float** inv_n) {
    *agg_bytes = pcrcg_kpconv_ws_bytes(ns);
    *agg = cv->take<char>(*agg_bytes);
    *wf = cv->take<float>((size_t)(nq > 0 ? nq : 1) * PCRCG_KPOINTS * (size_t)cin);
    *inv_n = cv->take<float>((size_t)(nq > 0 ? nq : 1));
}

extern "C" int pcrcg_kpconv_forward(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx, int h,
                                    int ld_idx, const float* x, int cin, const float* kp, float extent,
                                    const float* weights, int cout, float* out, int ld_out, void* ws, size_t ws_bytes,
                                    void* stream) {
    PCRCG_CHECK_ARG(nq >= 0 && cin >= 1 && cout >= 1 && ld_out >= cout && ws);
    if (nq == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(weights && out);
    Carver cv(ws, ws_bytes);
    void* agg; size_t agg_bytes; float *wf, *inv_n;
    kpconv_ws_layout(ws, ws_bytes, nq, ns, cin, &cv, &agg, &agg_bytes, &wf, &inv_n);
    PCRCG_CHECK_WS(cv);
    PCRCG_PROPAGATE(pcrcg_kpconv_aggregate(q_pts, nq, s_pts, ns, idx, h, ld_idx, x, cin, kp, extent, wf, inv_n, agg,
                                           agg_bytes, stream));
    return pcrcg_gemm_f32(wf, PCRCG_KPOINTS * cin, weights, cout, 0, out, ld_out, nq, cout, PCRCG_KPOINTS * cin, inv_n,
                          nullptr, stream);
}

extern "C" size_t pcrcg_kpconv_backward_ws_bytes(int nq, int cin, int cout) {
    return carve_bytes((size_t)(nq > 0 ? nq : 1) * PCRCG_KPOINTS * (size_t)cin, 4) +
           carve_bytes((size_t)(nq > 0 ? nq : 1) * (size_t)cout, 4);
}

namespace pcrcg {
namespace {
__global__ void __launch_bounds__(256) k_scale_rows(const float* __restrict__ src, int ld_src, const float* __restrict__ s,
                                                     float* __restrict__ dst, long total, int cols) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long r = t / cols;
    dst[t] = src[r * ld_src + (t - r * cols)] * s[r];
}
}  // namespace
}  // namespace pcrcg

extern "C" int pcrcg_kpconv_backward(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx, int h,
                                     int ld_idx, int cin, const float* kp, float extent, const float* weights, int cout,
                                     const float* dy, int ld_dy, const void* fwd_ws, size_t fwd_ws_bytes, float* dx,
                                     float* dweights, void* ws, size_t ws_bytes, void* stream) {
    PCRCG_CHECK_ARG(nq >= 0 && ns >= 1 && cin >= 1 && cout >= 1 && ld_dy >= cout && fwd_ws && ws);
    PCRCG_CHECK_ARG(weights && dy);
    hipStream_t st = as_stream(stream);
    const int kc = PCRCG_KPOINTS * cin;
    if (nq == 0) {
        if (dweights) PCRCG_CHECK_HIP(hipMemsetAsync(dweights, 0, (size_t)kc * cout * 4, st));
        return PCRCG_OK;
    }
    Carver fcv(const_cast<void*>(fwd_ws), fwd_ws_bytes);
    void* agg; size_t agg_bytes; float *wf, *inv_n;
    kpconv_ws_layout(const_cast<void*>(fwd_ws), fwd_ws_bytes, nq, ns, cin, &fcv, &agg, &agg_bytes, &wf, &inv_n);
    PCRCG_CHECK_WS(fcv);
    Carver cv(ws, ws_bytes);
    float* d_wf = cv.take<float>((size_t)nq * kc);
    float* dys = cv.take<float>((size_t)nq * cout);
    PCRCG_CHECK_WS(cv);
    if (dweights) {                               // dW = wf^T @ (dy / n)
        const long total = (long)nq * cout;
        hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, ld_dy, inv_n, dys,
                           total, cout);
        PCRCG_PROPAGATE(pcrcg_gemm_f32_ex(wf, kc, 1, dys, cout, 0, dweights, cout, kc, cout, nq, nullptr, nullptr, stream));
    }
    if (dx) {                                     // d wf = (dy / n) @ W^T, then scatter through the influence weights
        PCRCG_PROPAGATE(pcrcg_gemm_f32(dy, ld_dy, weights, cout, 1, d_wf, kc, nq, kc, cout, inv_n, nullptr, stream));
        PCRCG_PROPAGATE(pcrcg_kpconv_backward_dx(q_pts, nq, s_pts, ns, idx, h, ld_idx, d_wf, cin, kp, extent, dx, stream));
    }
    return PCRCG_OK;
}

// ---- element-wise pieces of the C++ train step (train_runner.hip) --------------------------------------------------
namespace pcrcg {
namespace {
__global__ void __launch_bounds__(256) k_add_lrelu(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                    float slope, float* __restrict__ y, int ldy, long total, int cols) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long r = t / cols;
    const int c = (int)(t - r * cols);
    const float v = a[r * lda + c] + b[r * ldb + c];
    y[r * ldy + c] = v >= 0.f ? v : v * slope;
}
// float4 forms of the element-wise kernels below (widths and leading dimensions multiples of 4, 16-byte aligned bases):
// one thread per four channels of a row
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__global__ void __launch_bounds__(256) k_add_lrelu4(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                     float slope, float* __restrict__ y, int ldy, long total4, int c4) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const long r = t / c4;
    const int c = 4 * (int)(t - r * c4);
    const float4 u = ld4(a + r * lda + c), w = ld4(b + r * ldb + c);
    float4 v = make_float4(u.x + w.x, u.y + w.y, u.z + w.z, u.w + w.w);
    v.x = v.x >= 0.f ? v.x : v.x * slope; v.y = v.y >= 0.f ? v.y : v.y * slope;
    v.z = v.z >= 0.f ? v.z : v.z * slope; v.w = v.w >= 0.f ? v.w : v.w * slope;
    st4(y + r * ldy + c, v);
}
__global__ void __launch_bounds__(256) k_add_lrelu_bwd4(const float* __restrict__ y, int ldy, const float* __restrict__ dy,
                                                         int ld_dy, float slope, float* __restrict__ ga, int lga,
                                                         float* __restrict__ gb, int lgb, long total4, int c4) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const long r = t / c4;
    const int c = 4 * (int)(t - r * c4);
    const float4 yv = ld4(y + r * ldy + c), d = ld4(dy + r * ld_dy + c);
    const float4 g = make_float4(d.x * (yv.x > 0.f ? 1.0f : slope), d.y * (yv.y > 0.f ? 1.0f : slope),
                                 d.z * (yv.z > 0.f ? 1.0f : slope), d.w * (yv.w > 0.f ? 1.0f : slope));
    if (ga) { float4 o = ld4(ga + r * lga + c); o.x += g.x; o.y += g.y; o.z += g.z; o.w += g.w; st4(ga + r * lga + c, o); }
    if (gb) { float4 o = ld4(gb + r * lgb + c); o.x += g.x; o.y += g.y; o.z += g.z; o.w += g.w; st4(gb + r * lgb + c, o); }
}
__global__ void __launch_bounds__(256) k_add2d4(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst,
                                                 long total4, int c4) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const long r = t / c4;
    const int c = 4 * (int)(t - r * c4);
    const float4 v = ld4(src + r * ld_src + c);
    float4 o = ld4(dst + r * ld_dst + c);
    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    st4(dst + r * ld_dst + c, o);
}
// g = dy * (y > 0 ? 1 : slope), added to ga and gb (either may be NULL).  lrelu keeps the sign, so y > 0 <=> a + b > 0;
// at exactly 0 torch's leaky_relu_backward takes the slope (x > 0 ? 1 : slope), and so does this.
__global__ void __launch_bounds__(256) k_add_lrelu_bwd(const float* __restrict__ y, int ldy, const float* __restrict__ dy,
                                                        int ld_dy, float slope, float* __restrict__ ga, int lga,
                                                        float* __restrict__ gb, int lgb, long total, int cols) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long r = t / cols;
    const int c = (int)(t - r * cols);
    const float g = dy[r * ld_dy + c] * (y[r * ldy + c] > 0.f ? 1.0f : slope);
    if (ga) ga[r * lga + c] += g;
    if (gb) gb[r * lgb + c] += g;
}
__global__ void __launch_bounds__(256) k_add2d(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst,
                                                long total, int cols) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long r = t / cols;
    const int c = (int)(t - r * cols);
    dst[r * ld_dst + c] += src[r * ld_src + c];
}
// db[c] += sum_r dy[r, c]: 64 columns x 4 row groups per workgroup (fp32 partial sums of a few hundred / thousand rows)
__global__ void __launch_bounds__(256) k_bias_grad(const float* __restrict__ dy, int ld, int rows, int cols,
                                                    float* __restrict__ db) {
    __shared__ float s[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < cols)
        for (int r = blockIdx.y * 4 + g; r < rows; r += 4 * gridDim.y) acc += dy[(long)r * ld + c];
    s[g][threadIdx.x & 63] = acc;
    __syncthreads();
    if (g == 0 && c < cols) atomicAdd(&db[c], s[0][threadIdx.x & 63] + s[1][threadIdx.x & 63] + s[2][threadIdx.x & 63] +
                                                 s[3][threadIdx.x & 63]);
}
// y = x / max(|x|, 1e-12) row-wise (F.normalize): dx += (dy - y * <y, dy>) / max(|x|, 1e-12); one wavefront per row
__global__ void __launch_bounds__(256) k_l2norm_bwd(const float* __restrict__ x, int ldx, const float* __restrict__ dy,
                                                     int ld_dy, float* __restrict__ dx, int ld_dx, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float ss = 0.f, sd = 0.f;
    for (int c = lane; c < cols; c += 64) {
        const float v = x[(long)r * ldx + c];
        ss += v * v;
        sd += v * dy[(long)r * ld_dy + c];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        ss += __shfl_xor(ss, d, 64);
        sd += __shfl_xor(sd, d, 64);
    }
    const float nrm = sqrtf(ss);
    if (nrm < 1e-12f) {                     // below the clamp the divisor is the constant eps: dx = dy / eps
        for (int c = lane; c < cols; c += 64) dx[(long)r * ld_dx + c] += dy[(long)r * ld_dy + c] * 1e12f;
        return;
    }
    const float inv = 1.0f / nrm, k = sd * inv * inv * inv;        // <x, dy> / |x|^3
    for (int c = lane; c < cols; c += 64)
        dx[(long)r * ld_dx + c] += dy[(long)r * ld_dy + c] * inv - x[(long)r * ldx + c] * k;
}
__global__ void __launch_bounds__(256) k_sigmoid_bwd(const float* __restrict__ s, const float* __restrict__ ds,
                                                      float* __restrict__ dx, int ld_dx, int rows) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float v = s[r];
    dx[(long)r * ld_dx] += ds[r] * v * (1.0f - v);
}
__global__ void __launch_bounds__(256) k_dot_acc(const float* __restrict__ a, const float* __restrict__ b, long n, float scale,
                                                  float* __restrict__ out) {
    __shared__ double s[256];
    double acc = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += (double)a[i] * (double)b[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, (float)(s[0] * (double)scale));
}
inline unsigned blocks_for(long total) { return (unsigned)((total + 255) / 256); }
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace

int tr_scale_rows(const float* src, int ld_src, const float* s, float* dst, int rows, int cols, hipStream_t st) {
    const long total = (long)rows * cols;
    if (total > 0) hipLaunchKernelGGL(k_scale_rows, dim3(blocks_for(total)), dim3(256), 0, st, src, ld_src, s, dst, total, cols);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
int tr_add_lrelu(const float* a, int lda, const float* b, int ldb, float slope, float* y, int ldy, int rows, int cols,
                 hipStream_t st) {
    const long total = (long)rows * cols;
    if (total > 0 && cols % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldy % 4 == 0 && al16(a) && al16(b) && al16(y))
        hipLaunchKernelGGL(k_add_lrelu4, dim3(blocks_for(total / 4)), dim3(256), 0, st, a, lda, b, ldb, slope, y, ldy, total / 4, cols / 4);
    else if (total > 0) hipLaunchKernelGGL(k_add_lrelu, dim3(blocks_for(total)), dim3(256), 0, st, a, lda, b, ldb, slope, y, ldy, total, cols);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
int tr_add_lrelu_bwd(const float* y, int ldy, const float* dy, int ld_dy, float slope, float* ga, int lga, float* gb, int lgb,
                     int rows, int cols, hipStream_t st) {
    const long total = (long)rows * cols;
    if (total > 0 && cols % 4 == 0 && ldy % 4 == 0 && ld_dy % 4 == 0 && (!ga || lga % 4 == 0) && (!gb || lgb % 4 == 0) && al16(y) &&
        al16(dy) && al16(ga) && al16(gb))
        hipLaunchKernelGGL(k_add_lrelu_bwd4, dim3(blocks_for(total / 4)), dim3(256), 0, st, y, ldy, dy, ld_dy, slope, ga, lga, gb, lgb,
                           total / 4, cols / 4);
    else if (total > 0)
        hipLaunchKernelGGL(k_add_lrelu_bwd, dim3(blocks_for(total)), dim3(256), 0, st, y, ldy, dy, ld_dy, slope, ga, lga, gb, lgb,
                           total, cols);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
int tr_add2d(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols, hipStream_t st) {
    const long total = (long)rows * cols;
    if (total > 0 && cols % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && al16(src) && al16(dst))
        hipLaunchKernelGGL(k_add2d4, dim3(blocks_for(total / 4)), dim3(256), 0, st, src, ld_src, dst, ld_dst, total / 4, cols / 4);
    else if (total > 0) hipLaunchKernelGGL(k_add2d, dim3(blocks_for(total)), dim3(256), 0, st, src, ld_src, dst, ld_dst, total, cols);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
int tr_bias_grad(const float* dy, int ld, int rows, int cols, float* db, hipStream_t st) {
    if (rows > 0 && cols > 0) {
        int gy = (rows + 255) / 256;
        if (gy > 64) gy = 64;
        if (debug_opts().deterministic) gy = 1;      // one add per column: no order to depend on
        hipLaunchKernelGGL(k_bias_grad, dim3((cols + 63) / 64, gy), dim3(256), 0, st, dy, ld, rows, cols, db);
    }
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
int tr_l2norm_bwd(const float* x, int ldx, const float* dy, int ld_dy, float* dx, int ld_dx, int rows, int cols, hipStream_t st) {
    if (rows > 0) hipLaunchKernelGGL(k_l2norm_bwd, dim3((rows + 3) / 4), dim3(256), 0, st, x, ldx, dy, ld_dy, dx, ld_dx, rows, cols);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
int tr_sigmoid_bwd(const float* s, const float* ds, float* dx, int ld_dx, int rows, hipStream_t st) {
    if (rows > 0) hipLaunchKernelGGL(k_sigmoid_bwd, dim3((rows + 255) / 256), dim3(256), 0, st, s, ds, dx, ld_dx, rows);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
int tr_dot_acc(const float* a, const float* b, long n, float scale, float* out, hipStream_t st) {
    if (n > 0) {
        long blocks = (n + 255) / 256;
        if (blocks > 256) blocks = 256;
        if (debug_opts().deterministic) blocks = 1;
        hipLaunchKernelGGL(k_dot_acc, dim3((unsigned)blocks), dim3(256), 0, st, a, b, n, scale, out);
    }
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
}  // namespace pcrcg
