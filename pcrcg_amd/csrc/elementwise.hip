// elementwise.hip -- small element-wise helpers of the network runner: strided row copies (the pieces
// of torch.cat), residual add, row L2 normalisation (F.normalize, ref:models/architectures.py:541,582)
// and the score head (sigmoid + clamp + NaN/Inf scrub, ref:models/architectures.py:176-179,576-579).
#include "common.h"
#include "pcrcg_train.h"

namespace pcrcg {
namespace {

struct CopyMulti { const float* src[4]; float* dst[4]; int rows[4]; };      // up to four copies of one width per launch (blockIdx.y)
__global__ void __launch_bounds__(256) k_copy2d(CopyMulti mm, int ld_src, int ld_dst, int cols) {
    const float* __restrict__ src = mm.src[blockIdx.y];
    float* __restrict__ dst = mm.dst[blockIdx.y];
    const int rows = mm.rows[blockIdx.y];
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)rows * cols) return;
    const long r = e / cols;
    const int c = (int)(e - r * cols);
    dst[r * ld_dst + c] = src[r * ld_src + c];
}

__global__ void __launch_bounds__(256) k_add(const float* __restrict__ a, const float* __restrict__ b,
                                              float* __restrict__ dst, long n) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) dst[e] = a[e] + b[e];
}

// one wavefront per row
__global__ void __launch_bounds__(256) k_l2norm_rows(const float* __restrict__ src, int ld_src, float* __restrict__ dst,
                                                      int ld_dst, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) {
        const float v = src[(long)r * ld_src + c];
        s += v * v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);   // F.normalize: x / max(||x||, eps)
    for (int c = lane; c < cols; c += 64) dst[(long)r * ld_dst + c] = src[(long)r * ld_src + c] * inv;
}

__global__ void __launch_bounds__(256) k_sigmoid_scores(const float* __restrict__ src, int ld_src,
                                                         float* __restrict__ dst, int rows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float v = 1.0f / (1.0f + expf(-src[(long)r * ld_src]));
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    if (isnan(v) || isinf(v)) v = 0.0f;
    dst[r] = v;
}

// The three heads of the network's last matrix for up to four pairs in ONE launch (round 5; ref:models/architectures.py:571-582):
// feats_f = F.normalize(x[:, :fd]), scores = clamp(sigmoid(x[:, fd]), 0, 1) and the same of x[:, fd + 1], non-finite -> 0 --
// the arithmetic of k_l2norm_rows and k_sigmoid_scores, one wavefront per row
struct HeadsMulti { const float* x[4]; float* feats[4]; float* s_ov[4]; float* s_sal[4]; int rows[4]; };
__global__ void __launch_bounds__(256) k_heads(HeadsMulti mm, int ld, int fd) {
    const int g = blockIdx.y, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= mm.rows[g]) return;
    const float* src = mm.x[g] + (long)r * ld;
    float s = 0.f;
    for (int c = lane; c < fd; c += 64) {
        const float v = src[c];
        s += v * v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    float* dst = mm.feats[g] + (long)r * fd;
    for (int c = lane; c < fd; c += 64) dst[c] = src[c] * inv;
    if (lane < 2) {
        float v = 1.0f / (1.0f + expf(-src[fd + lane]));
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        if (isnan(v) || isinf(v)) v = 0.0f;
        (lane == 0 ? mm.s_ov[g] : mm.s_sal[g])[r] = v;
    }
}

__global__ void __launch_bounds__(256) k_fill2d(float* __restrict__ dst, int ld, int rows, int cols, float v) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)rows * cols) return;
    const long r = e / cols;
    dst[r * ld + (int)(e - r * cols)] = v;
}

// x[inds3d[j] + row_offset, 0:c] = fmap[:, inds2d[j,1], inds2d[j,0]] * valid[inds2d[j,0], inds2d[j,1]];  x[.., c] = 1
// one wavefront per projected point: lanes run over the channels (fmap is [c, h, w]: a strided gather)
__global__ void __launch_bounds__(256) k_inject_image(const float* __restrict__ fmap, int c, int h, int w,
                                                       const float* __restrict__ valid, const long long* __restrict__ inds2d,
                                                       const long long* __restrict__ inds3d, int n, long row_offset,
                                                       long n_rows, float* __restrict__ x, int ldx) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    const long px = inds2d[2 * (long)j], py = inds2d[2 * (long)j + 1], row = inds3d[j] + row_offset;
    if (px < 0 || px >= w || py < 0 || py >= h || row < 0 || row >= n_rows) return;
    const float m = valid ? valid[px * h + py] : 1.0f;      // valid is stored [w, h] (the reference transposes it)
    float* dst = x + row * ldx;
    for (int ch = lane; ch < c; ch += 64) dst[ch] = fmap[((long)ch * h + py) * w + px] * m;
    if (lane == 0) dst[c] = 1.0f;
}

}  // namespace

int heads_multi(const float* const* x, const int* rows, int count, int ld, int fd, float* const* feats, float* const* s_ov,
                float* const* s_sal, hipStream_t st) {
    PCRCG_CHECK_ARG(count >= 1 && count <= 4 && fd >= 1 && ld >= fd + 2);
    HeadsMulti mm;
    int rmax = 0;
    for (int g = 0; g < 4; ++g) {
        const int k = g < count ? g : 0;
        PCRCG_CHECK_ARG(x[k] && feats[k] && s_ov[k] && s_sal[k] && rows[k] >= 0);
        mm.x[g] = x[k]; mm.feats[g] = feats[k]; mm.s_ov[g] = s_ov[k]; mm.s_sal[g] = s_sal[k]; mm.rows[g] = g < count ? rows[k] : 0;
        if (g < count) rmax = rows[k] > rmax ? rows[k] : rmax;
    }
    if (rmax == 0) return PCRCG_OK;
    hipLaunchKernelGGL(k_heads, dim3((rmax + 3) / 4, count), dim3(256), 0, st, mm, ld, fd);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

// dst_g[r, :cols] = src_g[r, :cols] for up to four (src, dst, rows) of one width and one pair of leading dimensions: one launch
int copy2d_multi(const float* const* src, float* const* dst, const int* rows, int count, int ld_src, int ld_dst, int cols,
                 hipStream_t st) {
    PCRCG_CHECK_ARG(count >= 1 && count <= 4 && cols >= 0 && ld_src >= cols && ld_dst >= cols);
    CopyMulti mm;
    int rmax = 0;
    for (int g = 0; g < 4; ++g) {
        const int k = g < count ? g : 0;
        mm.src[g] = src[k]; mm.dst[g] = dst[k]; mm.rows[g] = g < count ? rows[k] : 0;
        if (g < count) {
            PCRCG_CHECK_ARG(rows[k] >= 0 && (rows[k] == 0 || (src[k] && dst[k])));
            rmax = rows[k] > rmax ? rows[k] : rmax;
        }
    }
    if (rmax == 0 || cols == 0) return PCRCG_OK;
    const long total = (long)rmax * cols;
    hipLaunchKernelGGL(k_copy2d, dim3((unsigned)((total + 255) / 256), count), dim3(256), 0, st, mm, ld_src, ld_dst, cols);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
// Re-packed weight layouts of the train step and the way back for their gradients (include/pcrcg_train.h pcrcg_gather_jobs):
// job j fills dst[i] = src[m1[i]] + s2 src[m2[i]] (an index of -1 contributes zero; m2 may be NULL), or adds that to dst[i].
// m1 == NULL: a plain transpose, src [n / cols][cols] -> dst [cols][n / cols], through 32 x 32 LDS tiles (a map whose
// consecutive entries are a whole row apart would send every lane of a load to the same memory channel).
struct GatherJobs { const float* src; float* dst; const int* m1; const int* m2; int n; float s2; int accumulate; int cols; };
__global__ void __launch_bounds__(256) k_gather_jobs(const GatherJobs* __restrict__ jobs) {
    const GatherJobs jb = jobs[blockIdx.y];
    if (!jb.m1) {
        __shared__ float tile[32][33];
        const int cols = jb.cols, rows = jb.n / cols;
        const int tx = (cols + 31) / 32, ty = (rows + 31) / 32;
        const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;          // 32 x 8 threads, four rows each
        for (int t = blockIdx.x; t < tx * ty; t += gridDim.x) {
            const int r0 = (t / tx) * 32, c0 = (t % tx) * 32;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + ly + 8 * u, c = c0 + lx;
                tile[ly + 8 * u][lx] = (r < rows && c < cols) ? jb.src[(long)r * cols + c] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + ly + 8 * u, r = r0 + lx;              // dst row c, column r
                if (c < cols && r < rows) {
                    float* d = jb.dst + (long)c * rows + r;
                    *d = jb.accumulate ? *d + tile[lx][ly + 8 * u] : tile[lx][ly + 8 * u];
                }
            }
        }
        return;
    }
    auto one = [&](long i) {
        const int a = jb.m1[i];
        float v = a >= 0 ? jb.src[a] : 0.f;
        if (jb.m2) {
            const int b = jb.m2[i];
            if (b >= 0) v += jb.s2 * jb.src[b];
        }
        jb.dst[i] = jb.accumulate ? jb.dst[i] + v : v;
    };
    // four elements per thread and pass: their index loads, then their gathers, are in flight together
    const bool vec = ((reinterpret_cast<uintptr_t>(jb.dst) | reinterpret_cast<uintptr_t>(jb.m1) | reinterpret_cast<uintptr_t>(jb.m2)) & 15) == 0;
    if (!vec) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < jb.n; i += (long)gridDim.x * 256) one(i);
        return;
    }
    const long n4 = jb.n & ~3;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n4; i += (long)gridDim.x * 1024) {
        const int4 a = *reinterpret_cast<const int4*>(jb.m1 + i);
        int4 b = make_int4(-1, -1, -1, -1);
        if (jb.m2) b = *reinterpret_cast<const int4*>(jb.m2 + i);
        float4 v = make_float4(a.x >= 0 ? jb.src[a.x] : 0.f, a.y >= 0 ? jb.src[a.y] : 0.f, a.z >= 0 ? jb.src[a.z] : 0.f,
                               a.w >= 0 ? jb.src[a.w] : 0.f);
        if (b.x >= 0) v.x += jb.s2 * jb.src[b.x];
        if (b.y >= 0) v.y += jb.s2 * jb.src[b.y];
        if (b.z >= 0) v.z += jb.s2 * jb.src[b.z];
        if (b.w >= 0) v.w += jb.s2 * jb.src[b.w];
        float4* d = reinterpret_cast<float4*>(jb.dst + i);
        if (jb.accumulate) { const float4 o = *d; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *d = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (jb.n & 3)) one(n4 + threadIdx.x);
}
}  // namespace pcrcg

using namespace pcrcg;

extern "C" {

int pcrcg_fill2d(float* dst, int ld, int rows, int cols, float value, void* stream) {
    PCRCG_CHECK_ARG(rows >= 0 && cols >= 0 && ld >= cols);
    if (rows == 0 || cols == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(dst != nullptr);
    const long total = (long)rows * cols;
    hipLaunchKernelGGL(k_fill2d, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), dst, ld, rows,
                       cols, value);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

int pcrcg_inject_image_features(const float* fmap, int c, int h, int w, const float* valid, const int64_t* inds2d,
                                const int64_t* inds3d, int n, long row_offset, long n_rows, float* x, int ldx,
                                void* stream) {
    PCRCG_CHECK_ARG(c >= 1 && h >= 1 && w >= 1 && n >= 0 && n_rows >= 0 && ldx >= c + 1);
    if (n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(fmap && inds2d && inds3d && x);
    hipLaunchKernelGGL(k_inject_image, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), fmap, c, h, w, valid,
                       reinterpret_cast<const long long*>(inds2d), reinterpret_cast<const long long*>(inds3d), n, row_offset,
                       n_rows, x, ldx);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

int pcrcg_copy2d(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols, void* stream) {
    PCRCG_CHECK_ARG(rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= cols);
    if (rows == 0 || cols == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(src && dst);
    return pcrcg::copy2d_multi(&src, &dst, &rows, 1, ld_src, ld_dst, cols, as_stream(stream));
}

int pcrcg_gather_jobs(const void* jobs, int n_jobs, int max_n, void* stream) {
    static_assert(sizeof(GatherJobs) == sizeof(pcrcg_gather_job), "pcrcg_gather_job layout");
    PCRCG_CHECK_ARG(n_jobs >= 0 && max_n >= 0);
    if (n_jobs == 0 || max_n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(jobs != nullptr);
    const int gx = (max_n + 1023) / 1024 < 1024 ? (max_n + 1023) / 1024 : 1024;
    hipLaunchKernelGGL(k_gather_jobs, dim3(gx, n_jobs), dim3(256), 0, as_stream(stream), static_cast<const GatherJobs*>(jobs));
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

int pcrcg_add(const float* a, const float* b, float* dst, long n, void* stream) {
    PCRCG_CHECK_ARG(n >= 0);
    if (n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(a && b && dst);
    hipLaunchKernelGGL(k_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), a, b, dst, n);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

int pcrcg_l2norm_rows(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols, void* stream) {
    PCRCG_CHECK_ARG(rows >= 0 && cols >= 1 && ld_src >= cols && ld_dst >= cols);
    if (rows == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(src && dst);
    hipLaunchKernelGGL(k_l2norm_rows, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), src, ld_src, dst, ld_dst,
                       rows, cols);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

int pcrcg_sigmoid_scores(const float* src, int ld_src, float* dst, int rows, void* stream) {
    PCRCG_CHECK_ARG(rows >= 0 && ld_src >= 1);
    if (rows == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(src && dst);
    hipLaunchKernelGGL(k_sigmoid_scores, dim3((rows + 255) / 256), dim3(256), 0, as_stream(stream), src, ld_src, dst,
                       rows);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
}
