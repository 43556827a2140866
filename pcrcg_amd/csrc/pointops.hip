// pointops.hip -- HBM-bound point-wise blocks of the KPConv encoder/decoder on gfx950:
//   gather-max pooling      (max_pool,      ref:models/blocks.py:86-102)
//   nearest upsampling      (closest_pool,  ref:models/blocks.py:71-83)
//   InstanceNorm over all stacked points + LeakyReLU (+ residual)
//                           (BatchNormBlock/UnaryBlock/ResnetBottleneckBlock, ref:models/blocks.py:433-470,
//                            493-500, 650-678)
// Every kernel keeps channels on the fastest-moving lanes so that a wavefront reads and writes
// contiguous 256-byte (float) or 1-KiB (float4) segments of a feature row.
#include "common.h"

namespace pcrcg {
namespace {

constexpr int kWavesPerBlock = 4;

// one wavefront per (query row, block of 64 lanes x float4 | float channels)
// (round 5: up to four (features, table, output) triples of one width per launch -- blockIdx.y -- the pairs of a forward call)
struct GatherMulti { const float* x[4]; const long long* idx[4]; float* out[4]; int ns[4], nq[4], h[4], ld_idx[4]; };
template <bool VEC4>
__global__ void __launch_bounds__(kWavesPerBlock * 64) k_gather_max(GatherMulti mm, int c, int nchunk) {
    const int g = blockIdx.y;
    const float* __restrict__ x = mm.x[g];
    const long long* __restrict__ idx = mm.idx[g];
    float* __restrict__ out = mm.out[g];
    const int ns = mm.ns[g], nq = mm.nq[g], h = mm.h[g], ld_idx = mm.ld_idx[g];
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (item >= (long)nq * nchunk) return;
    const int q = (int)(item / nchunk), chunk = (int)(item - (long)q * nchunk);
    const long long* row = idx + (long)q * ld_idx;
    if (VEC4) {
        const int cb = chunk * 64 + lane;
        if (cb >= (c >> 2)) return;
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);   // the shadow row is all zeros  (:95)
        bool any = false;
#pragma unroll 4
        for (int j = 0; j < h; ++j) {
            const long long i = row[j];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i >= 0 && i < ns) v = reinterpret_cast<const float4*>(x + i * c)[cb];
            if (!any) { m = v; any = true; }
            else { m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w); }
        }
        reinterpret_cast<float4*>(out + (long)q * c)[cb] = m;
    } else {
        const int cc = chunk * 64 + lane;
        if (cc >= c) return;
        float m = 0.f;
        bool any = false;
#pragma unroll 4
        for (int j = 0; j < h; ++j) {
            const long long i = row[j];
            const float v = (i >= 0 && i < ns) ? x[i * c + cc] : 0.f;
            m = any ? fmaxf(m, v) : v;
            any = true;
        }
        out[(long)q * c + cc] = m;
    }
}

__global__ void __launch_bounds__(256) k_gather_first(const float* __restrict__ x, int ns, int c,
                                                       const long long* __restrict__ idx, int nq, int ld_idx,
                                                       float* __restrict__ out, int ld_out) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (q >= nq) return;
    const long long i = idx[(long)q * ld_idx];
    const bool real = i >= 0 && i < ns;
    for (int cc = lane; cc < c; cc += 64) out[(long)q * ld_out + cc] = real ? x[i * c + cc] : 0.f;
}

// ---- InstanceNorm statistics: deterministic two-stage column reduction in fp64 -----------------
constexpr int kStatChunks = 128;   // row chunks (= partial sums per channel)

struct ColsMulti { const float* x[4]; double* partial[4]; int n[4]; };      // up to four tensors per launch (blockIdx.z)
template <bool ATOMIC>      // ATOMIC: add the chunk's sums into zeroed [2][c] accumulators instead of storing partials
__global__ void __launch_bounds__(256) k_colstats_partial(ColsMulti mm, int c, int ldx) {
    const float* __restrict__ x = mm.x[blockIdx.z];
    double* __restrict__ partial = mm.partial[blockIdx.z];    /* [2][c][chunks] */
    const int n = mm.n[blockIdx.z];
    __shared__ double s_sum[4][64], s_sq[4][64];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int ch = blockIdx.y * 64 + lane;
    const int chunk = blockIdx.x, nchunks = gridDim.x;
    const long rows_per = ((long)n + nchunks - 1) / nchunks;
    const long r0 = chunk * rows_per, r1 = min((long)n, r0 + rows_per);
    double s = 0.0, sq = 0.0;
    if (ch < c) {
        long r = r0 + rl;
        for (; r + 12 < r1; r += 16) {   // four independent row reads in flight per thread
            const float v0 = x[r * ldx + ch], v1 = x[(r + 4) * ldx + ch], v2 = x[(r + 8) * ldx + ch],
                        v3 = x[(r + 12) * ldx + ch];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
            sq += ((double)v0 * v0 + (double)v1 * v1) + ((double)v2 * v2 + (double)v3 * v3);
        }
        for (; r < r1; r += 4) {
            const double v = (double)x[r * ldx + ch];
            s += v;
            sq += v * v;
        }
    }
    s_sum[rl][lane] = s;
    s_sq[rl][lane] = sq;
    __syncthreads();
    if (rl == 0 && ch < c) {
        s = (s_sum[0][lane] + s_sum[1][lane]) + (s_sum[2][lane] + s_sum[3][lane]);
        sq = (s_sq[0][lane] + s_sq[1][lane]) + (s_sq[2][lane] + s_sq[3][lane]);
        if (ATOMIC) {
            if (r0 < r1) {
                unsafeAtomicAdd(&partial[ch], s);
                unsafeAtomicAdd(&partial[(long)c + ch], sq);
            }
        } else {
            partial[(long)ch * nchunks + chunk] = s;                   // layout [2][c][nchunks]
            partial[((long)c + ch) * nchunks + chunk] = sq;
        }
    }
}

// one wavefront per channel: lanes over the (<= 128) chunk partials, fixed-order butterfly
__global__ void __launch_bounds__(256) k_colstats_final(const double* __restrict__ partial, int nchunks, int c,
                                                         double count, float eps, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int ch = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ch >= c) return;
    double s = 0.0, sq = 0.0;
    for (int k = lane; k < nchunks; k += 64) {
        s += partial[(long)ch * nchunks + k];
        sq += partial[((long)c + ch) * nchunks + k];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s += __shfl_xor(s, d, 64);
        sq += __shfl_xor(sq, d, 64);
    }
    if (lane != 0) return;
    const double mean = s / count;
    double var = sq / count - mean * mean;   // biased variance (InstanceNorm)
    if (var < 0.0) var = 0.0;
    stats[2 * ch] = (float)mean;
    stats[2 * ch + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ void __launch_bounds__(256) k_instnorm_apply(const float* __restrict__ x, int n, int c, int ldx,
                                                         const float* __restrict__ stats, const float* __restrict__ res,
                                                         int ldr, const float* __restrict__ res_stats, float slope,
                                                         float* __restrict__ y, int ldy) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)n * c) return;
    const long r = e / c;
    const int ch = (int)(e - r * c);
    float v = (x[r * ldx + ch] - stats[2 * ch]) * stats[2 * ch + 1];
    if (res) {
        float rv = res[r * ldr + ch];
        if (res_stats) rv = (rv - res_stats[2 * ch]) * res_stats[2 * ch + 1];
        v += rv;
    }
    y[r * ldy + ch] = v >= 0.f ? v : v * slope;
}

// float4 variant: c, ldx, ldr, ldy multiples of 4 and 16-byte aligned bases
__global__ void __launch_bounds__(256) k_instnorm_apply4(const float* __restrict__ x, int n, int c4, int ldx,
                                                          const float* __restrict__ stats, const float* __restrict__ res,
                                                          int ldr, const float* __restrict__ res_stats, float slope,
                                                          float* __restrict__ y, int ldy) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)n * c4) return;
    const long r = e / c4;
    const int q = (int)(e - r * c4);
    const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + 4 * q);
    const float4 s0 = *reinterpret_cast<const float4*>(stats + 8 * q), s1 = *reinterpret_cast<const float4*>(stats + 8 * q + 4);
    float4 v = make_float4((xv.x - s0.x) * s0.y, (xv.y - s0.z) * s0.w, (xv.z - s1.x) * s1.y, (xv.w - s1.z) * s1.w);
    if (res) {
        float4 rv = *reinterpret_cast<const float4*>(res + r * ldr + 4 * q);
        if (res_stats) {
            const float4 t0 = *reinterpret_cast<const float4*>(res_stats + 8 * q),
                         t1 = *reinterpret_cast<const float4*>(res_stats + 8 * q + 4);
            rv = make_float4((rv.x - t0.x) * t0.y, (rv.y - t0.z) * t0.w, (rv.z - t1.x) * t1.y, (rv.w - t1.z) * t1.w);
        }
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
    }
    v.x = v.x >= 0.f ? v.x : v.x * slope;
    v.y = v.y >= 0.f ? v.y : v.y * slope;
    v.z = v.z >= 0.f ? v.z : v.z * slope;
    v.w = v.w >= 0.f ? v.w : v.w * slope;
    *reinterpret_cast<float4*>(y + r * ldy + 4 * q) = v;
}

// (mean, rstd) of four consecutive channels from [2][c] fp64 column sums over `count` rows (as k_colstats_final)
__device__ __forceinline__ void stats4_from_sums(const double* __restrict__ sums, int c, int ch, double count, float eps,
                                                 float4& mean, float4& rstd) {
    float m[4], r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double mu = sums[ch + i] / count;
        double var = sums[c + ch + i] / count - mu * mu;   // biased variance (InstanceNorm)
        if (var < 0.0) var = 0.0;
        m[i] = (float)mu;
        r[i] = (float)(1.0 / sqrt(var + (double)eps));
    }
    mean = make_float4(m[0], m[1], m[2], m[3]);
    rstd = make_float4(r[0], r[1], r[2], r[3]);
}

// k_instnorm_apply4 with the statistics taken from fp64 column SUMS (what the GEMM epilogue's sums mode leaves): no
// finishing launch in between.  A thread keeps four channels and walks rows, so the fp64 arithmetic is paid once per
// thread.  RES: 0 none, 1 residual as is, 2 residual normalised by its own sums.
// (round 5: up to four tensors of one shape class per launch -- blockIdx.y -- the pairs of one forward call)
struct NormMulti {
    const float* x[4]; const double* sums[4]; const float* res[4]; const double* res_sums[4]; float* y[4];
    const float* s_pts[4]; float4* pk[4];
    int n[4]; double count[4];
    float* stats_out[4];
};
template <int RES>
__global__ void __launch_bounds__(256) k_instnorm_apply4_sums(NormMulti mm, int c4, int ldx, float eps, int ldr, float slope, int ldy,
                                                               int rows_per_block) {
    const int g = blockIdx.y;
    const float* __restrict__ x = mm.x[g];
    const double* __restrict__ sums = mm.sums[g];
    const float* __restrict__ res = mm.res[g];
    const double* __restrict__ res_sums = mm.res_sums[g];
    float* __restrict__ y = mm.y[g];
    const int n = mm.n[g];
    const double count = mm.count[g];
    if ((long)blockIdx.x * rows_per_block >= n) return;
    const int cb = c4 < 256 ? c4 : 256;          // channel groups per pass (c4 is a power of two times ... see host check)
    const int rp = 256 / cb;                     // rows handled at once
    const int r_lo = blockIdx.x * rows_per_block, r_hi = min(n, r_lo + rows_per_block);
    for (int qb = 0; qb < c4; qb += cb) {
        const int q = qb + (int)(threadIdx.x % cb);
        float4 m0, s0, m1 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = make_float4(1.f, 1.f, 1.f, 1.f);
        stats4_from_sums(sums, 4 * c4, 4 * q, count, eps, m0, s0);
        if (RES == 2) stats4_from_sums(res_sums, 4 * c4, 4 * q, count, eps, m1, s1);
        if (blockIdx.x == 0 && threadIdx.x < cb && mm.stats_out[g]) {       // the pairs k_colstats_final would have left
            float4* so = reinterpret_cast<float4*>(mm.stats_out[g] + 8 * q);
            so[0] = make_float4(m0.x, s0.x, m0.y, s0.y);
            so[1] = make_float4(m0.z, s0.z, m0.w, s0.w);
        }
        auto finish = [&](const float4& xv, float4 rv, long r) {
            float4 v = make_float4((xv.x - m0.x) * s0.x, (xv.y - m0.y) * s0.y, (xv.z - m0.z) * s0.z, (xv.w - m0.w) * s0.w);
            if (RES) {
                if (RES == 2) rv = make_float4((rv.x - m1.x) * s1.x, (rv.y - m1.y) * s1.y, (rv.z - m1.z) * s1.z, (rv.w - m1.w) * s1.w);
                v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            }
            v.x = v.x >= 0.f ? v.x : v.x * slope;
            v.y = v.y >= 0.f ? v.y : v.y * slope;
            v.z = v.z >= 0.f ? v.z : v.z * slope;
            v.w = v.w >= 0.f ? v.w : v.w * slope;
            *reinterpret_cast<float4*>(y + r * ldy + 4 * q) = v;
        };
        long r = r_lo + (int)(threadIdx.x / cb);
        for (; r + 3 * rp < r_hi; r += 4 * rp) {             // four rows in flight per thread
            float4 xv[4], rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xv[u] = *reinterpret_cast<const float4*>(x + (r + u * rp) * ldx + 4 * q);
                rv[u] = RES ? *reinterpret_cast<const float4*>(res + (r + u * rp) * ldr + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) finish(xv[u], rv[u], r + u * rp);
        }
        for (; r < r_hi; r += rp) {
            const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + 4 * q);
            const float4 rv = RES ? *reinterpret_cast<const float4*>(res + r * ldr + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            finish(xv, rv, r);
        }
    }
}

// The two normalisation kernels (no residual) that ALSO leave the KPConv support records of their output rows:
// pk[r] = (s_pts[r], sum_c y[r,c] > 0 ? 1 : 0) -- what k_row_positive (kpconv.hip) computes from y in a launch of its
// own when y is the input of a KPConv (ref:models/blocks.py:350-356: neighbours whose feature row sums to zero do not
// count).  The c4 <= 64 lanes that hold a row are contiguous in a wavefront: their partial sums meet in a butterfly.
__global__ void __launch_bounds__(256) k_instnorm_apply4_pack(const float* __restrict__ x, int n, int c4, int ldx,
                                                               const float* __restrict__ stats, float slope,
                                                               float* __restrict__ y, int ldy,
                                                               const float* __restrict__ s_pts, float4* __restrict__ pk) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)n * c4) return;                     // n * c4 is a multiple of c4: whole rows leave together
    const long r = e / c4;
    const int q = (int)(e - r * c4);
    const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + 4 * q);
    const float4 s0 = *reinterpret_cast<const float4*>(stats + 8 * q), s1 = *reinterpret_cast<const float4*>(stats + 8 * q + 4);
    float4 v = make_float4((xv.x - s0.x) * s0.y, (xv.y - s0.z) * s0.w, (xv.z - s1.x) * s1.y, (xv.w - s1.z) * s1.w);
    v.x = v.x >= 0.f ? v.x : v.x * slope;
    v.y = v.y >= 0.f ? v.y : v.y * slope;
    v.z = v.z >= 0.f ? v.z : v.z * slope;
    v.w = v.w >= 0.f ? v.w : v.w * slope;
    *reinterpret_cast<float4*>(y + r * ldy + 4 * q) = v;
    float part = (v.x + v.y) + (v.z + v.w);
    for (int sh = c4 >> 1; sh >= 1; sh >>= 1) part += __shfl_xor(part, sh, 64);
    if (q == 0) pk[r] = make_float4(s_pts[3 * r], s_pts[3 * r + 1], s_pts[3 * r + 2], part > 0.0f ? 1.f : 0.f);
}

__global__ void __launch_bounds__(256) k_instnorm_apply4_sums_pack(NormMulti mm, int c4, int ldx, float eps, float slope, int ldy,
                                                                    int rows_per_block) {
    const int g = blockIdx.y;
    const float* __restrict__ x = mm.x[g];
    const double* __restrict__ sums = mm.sums[g];
    float* __restrict__ y = mm.y[g];
    const float* __restrict__ s_pts = mm.s_pts[g];
    float4* __restrict__ pk = mm.pk[g];
    const int n = mm.n[g];
    const double count = mm.count[g];
    if ((long)blockIdx.x * rows_per_block >= n) return;
    const int rp = 256 / c4;                           // c4 <= 64 divides 64: one pass over the channel groups
    const int r_lo = blockIdx.x * rows_per_block, r_hi = min(n, r_lo + rows_per_block);
    const int q = (int)(threadIdx.x % c4);
    float4 m0, s0;
    stats4_from_sums(sums, 4 * c4, 4 * q, count, eps, m0, s0);
    for (long r = r_lo + (int)(threadIdx.x / c4); r < r_hi; r += rp) {
        const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + 4 * q);
        float4 v = make_float4((xv.x - m0.x) * s0.x, (xv.y - m0.y) * s0.y, (xv.z - m0.z) * s0.z, (xv.w - m0.w) * s0.w);
        v.x = v.x >= 0.f ? v.x : v.x * slope;
        v.y = v.y >= 0.f ? v.y : v.y * slope;
        v.z = v.z >= 0.f ? v.z : v.z * slope;
        v.w = v.w >= 0.f ? v.w : v.w * slope;
        *reinterpret_cast<float4*>(y + r * ldy + 4 * q) = v;
        float part = (v.x + v.y) + (v.z + v.w);
        for (int sh = c4 >> 1; sh >= 1; sh >>= 1) part += __shfl_xor(part, sh, 64);
        if (q == 0) pk[r] = make_float4(s_pts[3 * r], s_pts[3 * r + 1], s_pts[3 * r + 2], part > 0.0f ? 1.f : 0.f);
    }
}

}  // namespace

int instnorm_apply_sums_multi(const NormJob* jobs, int count, int c, int ldx, float eps, int ldr, float slope, int ldy, bool pack,
                              hipStream_t st);

// Normalise + LeakyReLU (no residual) and leave the KPConv support records of the output rows in `pk` (see the kernels).
// Statistics as (mean, rstd) pairs (`stats`) or as fp64 column sums (`sums`, `count`); exactly one of the two.
bool instnorm_pack_ok(int c, int ldx, int ldy) {
    const int c4 = c / 4;
    return c % 4 == 0 && c4 >= 1 && c4 <= 64 && (c4 & (c4 - 1)) == 0 && ldx % 4 == 0 && ldy % 4 == 0;
}
int instnorm_apply_pack(const float* x, int n, int c, int ldx, const float* stats, const double* sums, double count, float eps,
                        float slope, float* y, int ldy, const float* s_pts, float4* pk, hipStream_t st) {
    PCRCG_CHECK_ARG(n >= 0 && instnorm_pack_ok(c, ldx, ldy) && ldx >= c && ldy >= c && (stats != nullptr) != (sums != nullptr));
    if (n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(x && y && s_pts && pk && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0);
    const int c4 = c / 4;
    if (stats) {
        const long total = (long)n * c4;
        hipLaunchKernelGGL(k_instnorm_apply4_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, n, c4, ldx, stats,
                           slope, y, ldy, s_pts, pk);
    } else {
        NormJob one{x, sums, nullptr, nullptr, y, s_pts, pk, n, count};
        return instnorm_apply_sums_multi(&one, 1, c, ldx, eps, 0, slope, ldy, true, st);
    }
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

// lrelu(IN(x) [+ res | + IN(res)]) for up to four tensors of one width in ONE launch (the pairs of a forward call); pack: also
// leave the KPConv support records (instnorm_apply_pack).  The caller has checked the layout rules of the single-tensor entry.
int instnorm_apply_sums_multi(const NormJob* jobs, int count, int c, int ldx, float eps, int ldr, float slope, int ldy, bool pack,
                              hipStream_t st) {
    PCRCG_CHECK_ARG(jobs && count >= 1 && count <= 4 && c >= 4 && c % 4 == 0);
    const int c4 = c / 4;
    NormMulti mm;
    int nmax = 0;
    for (int g = 0; g < 4; ++g) {
        const NormJob& j = jobs[g < count ? g : 0];
        mm.x[g] = j.x; mm.sums[g] = j.sums; mm.res[g] = j.res; mm.res_sums[g] = j.res_sums; mm.y[g] = j.y;
        mm.s_pts[g] = j.s_pts; mm.pk[g] = j.pk; mm.n[g] = g < count ? j.n : 0; mm.count[g] = j.count;
        mm.stats_out[g] = g < count ? j.stats_out : nullptr;
        if (g < count) {
            PCRCG_CHECK_ARG(j.n >= 0 && j.x && j.sums && j.y && j.count >= 1.0);
            PCRCG_CHECK_ARG((j.res != nullptr) == (jobs[0].res != nullptr) && (j.res_sums != nullptr) == (jobs[0].res_sums != nullptr));
            nmax = j.n > nmax ? j.n : nmax;
        }
    }
    if (nmax == 0) return PCRCG_OK;
    const int rp = 256 / (c4 < 256 ? c4 : 256);
    // ~8 rows per thread, but at least ~256 workgroups' worth of parallelism on small inputs
    int rows_per_block = rp * 8;
    while (rows_per_block > rp && (long)(nmax + rows_per_block - 1) / rows_per_block * count < 256) rows_per_block -= rp;
    const dim3 grid((unsigned)((nmax + rows_per_block - 1) / rows_per_block), count);
    if (pack)
        hipLaunchKernelGGL(k_instnorm_apply4_sums_pack, grid, dim3(256), 0, st, mm, c4, ldx, eps, slope, ldy, rows_per_block);
    else if (!jobs[0].res)
        hipLaunchKernelGGL(k_instnorm_apply4_sums<0>, grid, dim3(256), 0, st, mm, c4, ldx, eps, ldr, slope, ldy, rows_per_block);
    else if (!jobs[0].res_sums)
        hipLaunchKernelGGL(k_instnorm_apply4_sums<1>, grid, dim3(256), 0, st, mm, c4, ldx, eps, ldr, slope, ldy, rows_per_block);
    else
        hipLaunchKernelGGL(k_instnorm_apply4_sums<2>, grid, dim3(256), 0, st, mm, c4, ldx, eps, ldr, slope, ldy, rows_per_block);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

size_t colstats_ws_bytes(int c) { return carve_bytes((size_t)kStatChunks * 2 * (size_t)(c > 0 ? c : 1), sizeof(double)); }

// shared with gnn.hip: finish a [chunks][2][c] fp64 partial buffer into (mean, rstd) pairs
int colstats_finalize(const double* partial, int nchunks, int c, double count, float eps, float* stats,
                      hipStream_t st) {
    hipLaunchKernelGGL(k_colstats_final, dim3((c + 3) / 4), dim3(256), 0, st, partial, nchunks, c, count, eps, stats);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
int colstats_chunks() { return kStatChunks; }

// out_g[q, :] = max over the table row's neighbours of x_g[., :] for up to four triples of one width in ONE launch
int gather_max_multi(const GatherJob* jobs, int count, int c, hipStream_t st) {
    PCRCG_CHECK_ARG(jobs && count >= 1 && count <= 4 && c >= 1);
    GatherMulti mm;
    bool vec = c % 4 == 0;
    int nq_max = 0;
    for (int g = 0; g < 4; ++g) {
        const GatherJob& j = jobs[g < count ? g : 0];
        mm.x[g] = j.x; mm.idx[g] = reinterpret_cast<const long long*>(j.idx); mm.out[g] = j.out;
        mm.ns[g] = j.ns; mm.nq[g] = g < count ? j.nq : 0; mm.h[g] = j.h; mm.ld_idx[g] = j.ld_idx;
        if (g < count) {
            PCRCG_CHECK_ARG(j.ns >= 0 && j.nq >= 0 && j.h >= 1 && j.ld_idx >= j.h && (j.nq == 0 || (j.x && j.idx && j.out)));
            vec = vec && ((reinterpret_cast<uintptr_t>(j.x) | reinterpret_cast<uintptr_t>(j.out)) & 15) == 0;
            nq_max = j.nq > nq_max ? j.nq : nq_max;
        }
    }
    if (nq_max == 0) return PCRCG_OK;
    const int per_wave = vec ? 256 : 64;
    const int nchunk = (c + per_wave - 1) / per_wave;
    const long items = (long)nq_max * nchunk;
    const dim3 grid((unsigned)((items + kWavesPerBlock - 1) / kWavesPerBlock), count);
    if (vec) hipLaunchKernelGGL(k_gather_max<true>, grid, dim3(kWavesPerBlock * 64), 0, st, mm, c, nchunk);
    else hipLaunchKernelGGL(k_gather_max<false>, grid, dim3(kWavesPerBlock * 64), 0, st, mm, c, nchunk);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

// column sums (sum, sum of squares) of up to four [n_g, c] tensors added into their zeroed [2][c] fp64 accumulators, one launch
int instnorm_colsums_multi(const float* const* x, double* const* sums, const int* n, int count, int c, int ldx, hipStream_t st) {
    PCRCG_CHECK_ARG(count >= 1 && count <= 4 && c >= 1 && ldx >= c);
    ColsMulti mm;
    int nmax = 0;
    for (int g = 0; g < 4; ++g) {
        const int k = g < count ? g : 0;
        PCRCG_CHECK_ARG(n[k] >= 1 && x[k] && sums[k]);
        mm.x[g] = x[k]; mm.partial[g] = sums[k]; mm.n[g] = n[k];
        if (g < count) nmax = n[k] > nmax ? n[k] : nmax;
    }
    // enough row chunks to fill the chip on tall inputs, few on short ones (each adds 2 atomics per channel)
    int chunks = (nmax + 63) / 64;
    if (chunks > kStatChunks) chunks = kStatChunks;
    hipLaunchKernelGGL(k_colstats_partial<true>, dim3(chunks, (c + 63) / 64, count), dim3(256), 0, st, mm, c, ldx);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

}  // namespace pcrcg

using namespace pcrcg;

extern "C" {

int pcrcg_gather_max(const float* x, int ns, int c, const int64_t* idx, int nq, int h, int ld_idx,
                     float* out, void* stream) {
    PCRCG_CHECK_ARG(ns >= 0 && c >= 1 && nq >= 0 && h >= 1 && ld_idx >= h);
    if (nq == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(x && idx && out);
    const GatherJob one{x, idx, out, ns, nq, h, ld_idx};
    return gather_max_multi(&one, 1, c, as_stream(stream));
}

int pcrcg_gather_first(const float* x, int ns, int c, const int64_t* idx, int nq, int ld_idx, float* out,
                       int ld_out, void* stream) {
    PCRCG_CHECK_ARG(ns >= 0 && c >= 1 && nq >= 0 && ld_idx >= 1 && ld_out >= c);
    if (nq == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(x && idx && out);
    hipLaunchKernelGGL(k_gather_first, dim3((nq + 3) / 4), dim3(256), 0, as_stream(stream), x, ns, c,
                       reinterpret_cast<const long long*>(idx), nq, ld_idx, out, ld_out);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

size_t pcrcg_instnorm_ws_bytes(int c) { return colstats_ws_bytes(c); }

int pcrcg_instnorm_stats_from_partials(const void* partials, int chunks, int c, double count, float eps,
                                       float* stats, void* stream) {
    PCRCG_CHECK_ARG(partials && stats && chunks >= 1 && c >= 1 && count >= 1.0);
    return colstats_finalize(static_cast<const double*>(partials), chunks, c, count, eps, stats, as_stream(stream));
}

int pcrcg_instnorm_stats(const float* x, int n, int c, int ldx, float eps, float* stats, void* ws,
                         size_t ws_bytes, void* stream) {
    PCRCG_CHECK_ARG(n >= 1 && c >= 1 && ldx >= c && x && stats && ws);
    Carver cv(ws, ws_bytes);
    double* partial = cv.take<double>((size_t)kStatChunks * 2 * c);
    PCRCG_CHECK_WS(cv);
    hipStream_t st = as_stream(stream);
    {
        ColsMulti mm;
        for (int g = 0; g < 4; ++g) { mm.x[g] = x; mm.partial[g] = partial; mm.n[g] = n; }
        hipLaunchKernelGGL(k_colstats_partial<false>, dim3(kStatChunks, (c + 63) / 64, 1), dim3(256), 0, st, mm, c, ldx);
    }
    return colstats_finalize(partial, kStatChunks, c, (double)n, eps, stats, st);
}

int pcrcg_instnorm_colsums(const float* x, int n, int c, int ldx, void* sums, void* stream) {
    PCRCG_CHECK_ARG(n >= 1 && c >= 1 && ldx >= c && x && sums);
    const float* xs[1] = {x};
    double* ss[1] = {static_cast<double*>(sums)};
    return instnorm_colsums_multi(xs, ss, &n, 1, c, ldx, as_stream(stream));
}

int pcrcg_instnorm_apply_sums(const float* x, int n, int c, int ldx, const void* sums, double count, float eps,
                              const float* res, int ldr, const void* res_sums, float slope, float* y, int ldy,
                              void* stream) {
    PCRCG_CHECK_ARG(n >= 0 && c >= 4 && c % 4 == 0 && ldx >= c && ldy >= c && count >= 1.0);
    if (n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(x && sums && y && (!res || ldr >= c) && (!res_sums || res));
    const int c4 = c / 4;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    // the kernel's thread -> channel-group map needs the groups to tile 256 threads (or be a multiple of 256)
    PCRCG_CHECK_ARG((c4 <= 256 ? 256 % c4 == 0 : c4 % 256 == 0) && ldx % 4 == 0 && ldy % 4 == 0 && al16(x) && al16(y) &&
                    (!res || (ldr % 4 == 0 && al16(res))));
    NormJob one{x, static_cast<const double*>(sums), res, static_cast<const double*>(res_sums), y, nullptr, nullptr, n, count};
    return instnorm_apply_sums_multi(&one, 1, c, ldx, eps, ldr, slope, ldy, false, as_stream(stream));
}

int pcrcg_instnorm_apply(const float* x, int n, int c, int ldx, const float* stats, const float* res,
                         int ldr, const float* res_stats, float slope, float* y, int ldy, void* stream) {
    PCRCG_CHECK_ARG(n >= 0 && c >= 1 && ldx >= c && ldy >= c);
    if (n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(x && stats && y);
    PCRCG_CHECK_ARG(!res || ldr >= c);
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = c % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && al16(x) && al16(y) && al16(stats) &&
                     (!res || (ldr % 4 == 0 && al16(res))) && (!res_stats || al16(res_stats));
    if (vec) {
        const long total = (long)n * (c / 4);
        hipLaunchKernelGGL(k_instnorm_apply4, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), x,
                           n, c / 4, ldx, stats, res, ldr, res_stats, slope, y, ldy);
    } else {
        const long total = (long)n * c;
        hipLaunchKernelGGL(k_instnorm_apply, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), x,
                           n, c, ldx, stats, res, ldr, res_stats, slope, y, ldy);
    }
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
}
