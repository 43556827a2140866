// kpconv_fused.hip -- whole KPConv.forward (ref:models/blocks.py:264-372) in one kernel for the fine
// levels of the pyramid, where the [Nq, 15*Cin] intermediate of the two-stage path (kpconv.hip + GEMM)
// is the largest HBM stream of the network (0.23 GB written and read back for one 60k-point layer).
//
// One workgroup (4 wavefronts) owns 16 queries.  For every chunk of 64 input channels:
//   phase A  each wavefront aggregates 4 of the queries exactly like k_kpconv_mfma (one influence
//            weight and one float4 of a neighbour's feature row per lane, v_mfma_f32_16x16x4_f32 over
//            [16 kernel points] x [4 neighbours] x [16 channel groups]) and parks the [15 x 64] result
//            in LDS (rows padded to 964 floats: conflict-free b128 reads across the 16 queries);
//   phase B  the 16 x 960 LDS tile is contracted with the matching rows of the layer's weights on the
//            matrix cores (M = 16 queries, K = 960, N = Cout; each wavefront owns Cout/4 columns).  The
//            weights are read straight from L2 as float4s along K from a K-contiguous copy
//            Wt [Cout, 15*Cin] (no LDS staging: every element is used once per workgroup).
// Output rows are scaled by 1/n_q (neighbour count with positive feature sum, :369-372).
#include <hip/hip_ext.h>

#include "common.h"

namespace pcrcg {
namespace {

constexpr int K = PCRCG_KPOINTS;
constexpr int TQ = 16;                 // queries per workgroup
constexpr int CC = 64;                 // input channels per chunk
constexpr int ROW = K * CC + 4;        // LDS row stride in floats (964 = 4 mod 64)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT>   // 16-column output tiles per wavefront: Cout = 64 * NT
__global__ void __launch_bounds__(256) k_kpconv_fused(
    const float* __restrict__ q_pts, int nq, const float* __restrict__ s_pts, int ns,
    const long long* __restrict__ idx, int H, int ld_idx, const float* __restrict__ x, int cin,
    const float* __restrict__ kp, float extent, const unsigned char* __restrict__ pos,
    const float* __restrict__ wt /* [cout, 15*cin] */, float* __restrict__ out, int ld_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [TQ][ROW] + [TQ]
    float* inv_s = smem + TQ * ROW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hsub = lane >> 4, j = lane & 15;
    const int q0 = blockIdx.x * TQ;
    const bool jvalid = j < K;
    const float kpx = jvalid ? kp[3 * j] : 0.f, kpy = jvalid ? kp[3 * j + 1] : 0.f, kpz = jvalid ? kp[3 * j + 2] : 0.f;
    const float inv_extent = 1.0f / extent;
    const int kdim = K * cin;

    f32x4 acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    for (int c0 = 0; c0 < cin; c0 += CC) {
        // ---------------- phase A: aggregate this wavefront's 4 queries into LDS
        for (int qq = 0; qq < TQ / 4; ++qq) {
            const int ql = wave * (TQ / 4) + qq;
            const int q = q0 + ql;
            float* row = smem + ql * ROW;
            f32x4 a4[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) a4[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            int npos = 0;
            if (q < nq) {   // wave-uniform
                const float qx = q_pts[3 * (long)q], qy = q_pts[3 * (long)q + 1], qz = q_pts[3 * (long)q + 2];
                for (int hc = 0; hc < H; hc += 64) {
                    const int h = hc + lane;
                    const long long iv = idx[(long)q * ld_idx + (h < H ? h : H - 1)];
                    const int i = (h < H && iv >= 0 && iv < ns) ? (int)iv : -1;
                    const long ic = i >= 0 ? i : 0;
                    const float px = s_pts[3 * ic] - qx, py = s_pts[3 * ic + 1] - qy, pz = s_pts[3 * ic + 2] - qz;
                    if (c0 == 0) npos += __popcll(__ballot(i >= 0 && pos[ic] != 0));
                    const int hn = H - hc < 64 ? H - hc : 64;
                    for (int h0 = 0; h0 < hn; h0 += 16) {
                        float w[4];
                        float4 v[4];
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const int src = h0 + 4 * s + hsub;
                            const int ii = __shfl(i, src, 64);
                            const float nx = __shfl(px, src, 64), ny = __shfl(py, src, 64), nz = __shfl(pz, src, 64);
                            const bool real = ii >= 0 && h0 + 4 * s < hn;
                            w[s] = 0.f;
                            if (real && jvalid) {
                                const float dx = nx - kpx, dy = ny - kpy, dz = nz - kpz;
                                w[s] = fmaxf(1.0f - __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz) * inv_extent, 0.0f);
                            }
                            const float4 t = *reinterpret_cast<const float4*>(x + (long)(real ? ii : 0) * cin + c0 + 4 * j);
                            v[s] = make_float4(real ? t.x : 0.f, real ? t.y : 0.f, real ? t.z : 0.f, real ? t.w : 0.f);
                        }
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            a4[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], v[s].x, a4[0], 0, 0, 0);
                            a4[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], v[s].y, a4[1], 0, 0, 0);
                            a4[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], v[s].z, a4[2], 0, 0, 0);
                            a4[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], v[s].w, a4[3], 0, 0, 0);
                        }
                    }
                }
            }
            // D layout: register r of lane (hsub, j) = kernel point 4*hsub + r, channels 4j .. 4j+3
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * hsub + r;
                if (k < K)
                    *reinterpret_cast<float4*>(row + k * CC + 4 * j) = make_float4(a4[0][r], a4[1][r], a4[2][r], a4[3][r]);
            }
            if (c0 == 0 && lane == 0) inv_s[ql] = 1.0f / (float)(npos > 1 ? npos : 1);
        }
        __syncthreads();
        // ---------------- phase B: out[16, cout] += wf[16, 960] @ W[k*cin + c0 + c, :]
        // A[i = lane&15 (query)][k = lane>>4]; B[k = lane>>4][j = lane&15 (column)]; four consecutive
        // K per lane and load (MFMA t takes component t on both sides)
        const float* arow = smem + j * ROW + 4 * hsub;
        constexpr int UB = 6;                                      // K-steps whose operands are in flight together
        for (int kc0 = 0; kc0 < K * CC; kc0 += 16 * UB) {          // 960 = 10 * 96
            float4 a[UB], b[UB][NT];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int kk = kc0 + 16 * u + 4 * hsub;                // this lane's first K inside the chunk
                const long kg = (long)(kk >> 6) * cin + c0 + (kk & 63);   // ... and in the [15*cin] axis
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    b[u][t] = *reinterpret_cast<const float4*>(wt + (long)(wave * 16 * NT + t * 16 + j) * kdim + kg);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) a[u] = *reinterpret_cast<const float4*>(arow + kc0 + 16 * u);
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u][t].x, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u][t].y, acc[t][1], 0, 0, 0);
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u][t].z, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u][t].w, acc[t][1], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    // epilogue: D register r of lane (hsub, j) = query 4*hsub + r, column (wave*NT + t)*16 + j
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ql = 4 * hsub + r;
        const int q = q0 + ql;
        if (q >= nq) continue;
        const float sc = inv_s[ql];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            out[(long)q * ld_out + wave * 16 * NT + t * 16 + j] = (acc[t][0][r] + acc[t][1][r]) * sc;
    }
}

__global__ void __launch_bounds__(256) k_row_positive2(const float* __restrict__ x, int ns, int cin,
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

}  // namespace
}  // namespace pcrcg

using namespace pcrcg;

extern "C" {

int pcrcg_kpconv_fused_supported(int nq, int cin, int cout) {
    return cin >= 64 && cin % 64 == 0 && (cout == 64 || cout == 128 || cout == 256) && nq >= 1;
}

int pcrcg_kpconv_fused(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx, int h,
                       int ld_idx, const float* x, int cin, const float* kp, float extent, const float* wt,
                       int cout, float* out, int ld_out, void* ws, size_t ws_bytes, void* stream) {
    PCRCG_CHECK_ARG(nq >= 0 && ns >= 1 && h >= 1 && ld_idx >= h && ld_out >= cout && extent > 0.0f);
    PCRCG_CHECK_ARG(pcrcg_kpconv_fused_supported(nq > 0 ? nq : 1, cin, cout));
    if (nq == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(q_pts && s_pts && idx && x && kp && wt && out && ws);
    PCRCG_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(wt) & 15) == 0);
    Carver cv(ws, ws_bytes);
    unsigned char* pos = cv.take<unsigned char>((size_t)ns + 1);
    PCRCG_CHECK_WS(cv);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(k_row_positive2, dim3((ns + 3) / 4), dim3(256), 0, st, x, ns, cin, pos);
    KpProfScope prof_scope(st, nq, h, cin, cout, 1);   // start / stop events of the fused kernel itself
    const size_t lds = sizeof(float) * (TQ * ROW + TQ);
    const int blocks = (nq + TQ - 1) / TQ;
    const long long* idx_ll = reinterpret_cast<const long long*>(idx);
#define LAUNCH(NTV)                                                                                         \
    do {                                                                                                    \
        auto kern = k_kpconv_fused<NTV>;                                                                    \
        static bool configured = false;                                                                     \
        if (!configured) {                                                                                  \
            PCRCG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                        \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
            configured = true;                                                                              \
        }                                                                                                   \
        hipExtLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, prof_scope.a, prof_scope.b, 0, q_pts, nq, s_pts, \
                              ns, idx_ll, h, ld_idx, x, cin, kp, extent, (const unsigned char*)pos, wt, out, ld_out); \
    } while (0)
    if (cout == 64) LAUNCH(1);
    else if (cout == 128) LAUNCH(2);
    else LAUNCH(4);
#undef LAUNCH
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
}
