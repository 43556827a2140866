// kpconv.hip -- KPConv gather + kernel-point aggregation on gfx950 (stage 1 of KPConv.forward,
// ref:models/blocks.py:264-354,369-372).
//
// For every query q (one 64-lane wavefront per query):
//   geometry  lanes = neighbours: load idx[q,h], the support point, and evaluate the 15 linear
//             influence weights  w[h,k] = max(0, 1 - |s_h - q - kp_k| / extent)  (:272-289,328);
//             weights and indices are staged in LDS
//   aggregate lanes = channels: for each neighbour read the feature row x[idx,:] (coalesced 256-byte
//             rows) once and accumulate all 15 kernel points from registers:
//             wf[q,k,c] += w[h,k] * x[idx[q,h],c]   (:351-354)
//   count     n_q = max(1, #{h : sum_c x[idx[q,h],c] > 0})   (:369-371) from per-support flags that a
//             small pre-pass derives from x (the flag depends on the support only).
// Shadow neighbours (idx == ns) have weight 0 and feature 0 in the reference (:269,:348) and are
// skipped.  Output wf [nq, 15*cin] feeds the dense contraction (pcrcg_gemm_f32, row_scale = 1/n_q).
#include "common.h"

namespace pcrcg {
namespace {

constexpr int K = PCRCG_KPOINTS;
constexpr int kWavesPerBlock = 4;

// pos[s] = (sum_c x[s,c] > 0); one wavefront per support row.
__global__ void __launch_bounds__(256) k_row_positive(const float* __restrict__ x, int ns, int cin,
                                                       unsigned char* __restrict__ pos) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= ns) return;
    float s = 0.0f;
    for (int c = lane; c < cin; c += 64) s += x[(long)row * cin + c];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) pos[row] = s > 0.0f ? 1 : 0;
}

template <int J>  // channels handled per lane and pass: 64*J
__global__ void __launch_bounds__(kWavesPerBlock * 64) k_kpconv_aggregate(
    const float* __restrict__ q_pts, int nq, const float* __restrict__ s_pts, int ns,
    const long long* __restrict__ idx, int H, int ld_idx, const float* __restrict__ x, int cin,
    const float* __restrict__ kp, float extent, const unsigned char* __restrict__ pos, float* __restrict__ wf,
    float* __restrict__ inv_n) {
    __shared__ __attribute__((aligned(16))) float s_w[kWavesPerBlock][64][16];
    __shared__ int s_idx[kWavesPerBlock][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * kWavesPerBlock + wave, nw = gridDim.x * kWavesPerBlock;
    float kpx[K], kpy[K], kpz[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { kpx[k] = kp[3 * k]; kpy[k] = kp[3 * k + 1]; kpz[k] = kp[3 * k + 2]; }
    const float inv_extent = 1.0f / extent;

    for (int q = gw; q < nq; q += nw) {
        const float qx = q_pts[3 * (long)q], qy = q_pts[3 * (long)q + 1], qz = q_pts[3 * (long)q + 2];
        int npos = 0;
        for (int cbase = 0; cbase < cin; cbase += 64 * J) {
            float acc[K][J];
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int j = 0; j < J; ++j) acc[k][j] = 0.0f;
            for (int hc = 0; hc < H; hc += 64) {
                // ---- geometry: lanes = neighbours
                const int h = hc + lane;
                int i = ns;
                if (h < H) i = (int)idx[(long)q * ld_idx + h];
                const bool real = i >= 0 && i < ns;
                float w[16];
                if (real) {
                    const float nx = s_pts[3 * (long)i] - qx, ny = s_pts[3 * (long)i + 1] - qy,
                                nz = s_pts[3 * (long)i + 2] - qz;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const float dx = nx - kpx[k], dy = ny - kpy[k], dz = nz - kpz[k];
                        const float d2 = dx * dx + dy * dy + dz * dz;
                        w[k] = fmaxf(1.0f - sqrtf(d2) * inv_extent, 0.0f);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < K; ++k) w[k] = 0.0f;
                }
                w[15] = 0.0f;
                __builtin_amdgcn_wave_barrier();
                float4* dst = reinterpret_cast<float4*>(&s_w[wave][lane][0]);
                dst[0] = make_float4(w[0], w[1], w[2], w[3]);
                dst[1] = make_float4(w[4], w[5], w[6], w[7]);
                dst[2] = make_float4(w[8], w[9], w[10], w[11]);
                dst[3] = make_float4(w[12], w[13], w[14], w[15]);
                s_idx[wave][lane] = real ? i : -1;
                if (cbase == 0) npos += __popcll(__ballot(real && pos[real ? i : 0] != 0));
                __builtin_amdgcn_wave_barrier();
                // ---- aggregate: lanes = channels
                const int hn = H - hc < 64 ? H - hc : 64;
                for (int hh = 0; hh < hn; ++hh) {
                    const int ii = s_idx[wave][hh];
                    if (ii < 0) continue;  // wave-uniform
                    const float4* src = reinterpret_cast<const float4*>(&s_w[wave][hh][0]);
                    const float4 w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
                    const float wk[K] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w,
                                         w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z};
                    const float* xr = x + (long)ii * cin + cbase;
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const int c = lane + 64 * j;
                        const float xv = cbase + c < cin ? xr[c] : 0.0f;
#pragma unroll
                        for (int k = 0; k < K; ++k) acc[k][j] = fmaf(wk[k], xv, acc[k][j]);
                    }
                }
            }
            float* o = wf + (long)q * K * cin + cbase;
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const int c = lane + 64 * j;
                    if (cbase + c < cin) o[(long)k * cin + c] = acc[k][j];
                }
        }
        if (lane == 0) inv_n[q] = 1.0f / (float)(npos > 1 ? npos : 1);
    }
}

}  // namespace
}  // namespace pcrcg

using namespace pcrcg;

extern "C" {

size_t pcrcg_kpconv_ws_bytes(int ns) { return carve_bytes((size_t)(ns > 0 ? ns : 0) + 1, 1); }

int pcrcg_kpconv_aggregate(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx,
                           int h, int ld_idx, const float* x, int cin, const float* kp, float extent,
                           float* wf, float* inv_n, void* ws, size_t ws_bytes, void* stream) {
    PCRCG_CHECK_ARG(nq >= 0 && ns >= 0 && h >= 1 && ld_idx >= h && cin >= 1);
    PCRCG_CHECK_ARG(extent > 0.0f);
    if (nq == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(q_pts && s_pts && idx && x && kp && wf && inv_n && ws);
    Carver cv(ws, ws_bytes);
    unsigned char* pos = cv.take<unsigned char>((size_t)ns + 1);
    PCRCG_CHECK_WS(cv);
    hipStream_t st = as_stream(stream);
    if (ns > 0) hipLaunchKernelGGL(k_row_positive, dim3((ns + 3) / 4), dim3(256), 0, st, x, ns, cin, pos);
    int blocks = (nq + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > 256 * 32) blocks = 256 * 32;
    const long long* idx_ll = reinterpret_cast<const long long*>(idx);
#define LAUNCH(J)                                                                                             \
    hipLaunchKernelGGL(k_kpconv_aggregate<J>, dim3(blocks), dim3(kWavesPerBlock * 64), 0, st, q_pts, nq, s_pts, \
                       ns, idx_ll, h, ld_idx, x, cin, kp, extent, pos, wf, inv_n)
    if (cin <= 64) LAUNCH(1);
    else if (cin <= 128) LAUNCH(2);
    else LAUNCH(4);
#undef LAUNCH
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
}
