// kpconv.hip -- KPConv neighbour gather + kernel-point aggregation on gfx950 (stage 1 of
// KPConv.forward, ref:models/blocks.py:264-354,369-372).  This is the dominant HBM-bound kernel of the
// path (bench.py `roofline`).
//
//   wf[q,k,c] = sum_h w[q,h,k] * x[idx[q,h],c],   w[q,h,k] = max(0, 1 - |s[idx[q,h]] - q - kp[k]| / extent)
//   n_q       = max(1, #{h : sum_c x[idx[q,h],c] > 0})                       (:369-371)
//
// Main kernel (k_kpconv_mfma): one wavefront per (query, channel chunk).  The aggregation is a
// [16 kernel points] x [H neighbours] x [channels] contraction, done on the matrix cores with
// v_mfma_f32_16x16x4_f32 (exact fp32).  Per step of 4 neighbours:
//   * lane (hsub = lane>>4, j = lane&15) evaluates ONE influence weight -- neighbour h0+hsub against
//     kernel point j -- which is exactly the A-operand element A[i=j][k=hsub] it must supply;
//   * the same lane loads one float4 of feature row idx[h0+hsub] at channel 4j: 16 lanes read a whole
//     256-byte row segment (coalesced), and the 4 components feed 4 MFMAs as B[k=hsub][col=j], so MFMA n
//     accumulates channels {4j+n};
//   * neighbour indices and centred neighbour coordinates are loaded once per 64 neighbours
//     (lanes = neighbours) and broadcast with wave shuffles; nothing is staged through LDS.
// The D layout (row = 4*(lane>>4)+r = kernel point, col = j) lets each lane store float4s of 4
// consecutive channels of wf.
// Shadow neighbours (idx == ns) have weight 0 and feature 0 in the reference (:269,:348): their weight
// is forced to 0 and no row is read.
//
// Fallbacks: k_kpconv_c1 for Cin == 1 (first layer, features are a column of ones; four lanes per query), and the scalar
// k_kpconv_generic for channel counts that are not a multiple of 4.
#include <mutex>
#include <vector>

#include <hip/hip_ext.h>

#include "common.h"

namespace pcrcg {

// optional HIP-event timing of every KPConv launch (bench.py roofline)
struct ProfRec { hipEvent_t a, b; int nq, h, cin, cout, kind; };
static int g_prof_on = 0;         // bit 0: KPConv kernels (kinds 0-2), bit 1: the GEMM family (kind 3)
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;   // forwards may be enqueued from several host threads

KpProfScope::KpProfScope(hipStream_t s, int nq_, int h_, int cin_, int cout_, int kind_)
    : st(s), a(nullptr), b(nullptr), nq(nq_), h(h_), cin(cin_), cout(cout_), kind(kind_),
      on((g_prof_on & (kind_ == 3 ? 2 : kind_ == 4 ? 4 : 1)) != 0) {
    if (!on) return;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
}
KpProfScope::~KpProfScope() {
    if (!on) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof.push_back({a, b, nq, h, cin, cout, kind});
}

namespace {

constexpr int K = PCRCG_KPOINTS;
constexpr int kWavesPerBlock = 4;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short bf16_rne(float f) {
    const unsigned u = __float_as_uint(f);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// pos[s] = (sum_c x[s,c] > 0); one wavefront per support row.  Also writes the packed record
// pk[s] = (x, y, z, pos ? 1 : 0): the gather kernels fetch a neighbour's coordinates and flag with ONE 16-byte
// gather (one cache line) instead of a 12-byte and a 1-byte gather (two lines) -- random line fetches, not
// feature bytes, are what loads L2 in this kernel.
__global__ void __launch_bounds__(256) k_row_positive(const float* __restrict__ x, int ns, int cin,
                                                       const float* __restrict__ s_pts, unsigned char* __restrict__ pos,
                                                       float4* __restrict__ pk, unsigned short* __restrict__ xb) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= ns) return;
    float s = 0.0f;
    for (int c = lane; c < cin; c += 64) {
        const float v = x[(long)row * cin + c];
        s += v;
        if (xb) xb[(long)row * cin + c] = bf16_rne(v);      // bf16 feature-storage variant: the copy the gathers read
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) {
        pos[row] = s > 0.0f ? 1 : 0;
        pk[row] = make_float4(s_pts[3 * (long)row], s_pts[3 * (long)row + 1], s_pts[3 * (long)row + 2], s > 0.0f ? 1.f : 0.f);
    }
}

// Cin == 1: pk[s] = (x, y, z, feature)
__global__ void __launch_bounds__(256) k_pack_c1(const float* __restrict__ x, int ns, const float* __restrict__ s_pts,
                                                  float4* __restrict__ pk) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row < ns) pk[row] = make_float4(s_pts[3 * (long)row], s_pts[3 * (long)row + 1], s_pts[3 * (long)row + 2], x[row]);
}

struct __attribute__((packed, aligned(4))) P3 { float x, y, z; };

// BF16: the bf16 feature-storage variant (BASELINE.json configs[1] "bf16/fp32"): x and wf are bf16 in memory (half the
// gather and half the wf bytes), the aggregation still runs on the fp32 MFMA with fp32 accumulation; wf is rounded to
// nearest-even on the way out.
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const unsigned short* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}

// TAIL (round 6, PCR-CG's 129-channel input in rows of 132 floats): cin = 64 NB + 4 with ONE wavefront per query -- the last
// float4 of a row does not get a 64-channel block of its own (a third wavefront per query that repeats the index loads,
// the shuffles and the 15 square roots for four channels: what the plain plan gave this width) but rides along on the vector
// units: every lane multiplies its influence weight with its neighbour's last float4 (the 16 lanes of a neighbour read one
// address), the four neighbour groups meet by two shuffles at the end.
template <int NB, bool NT_STORE, typename FT, bool TAIL = false>  // 64-channel blocks handled per wavefront (one float4 per lane and block)
__global__ void __launch_bounds__(kWavesPerBlock * 64) k_kpconv_mfma(
    const float* __restrict__ q_pts, int nq, const float* __restrict__ s_pts, int ns,
    const long long* __restrict__ idx, int H, int ld_idx, const FT* __restrict__ x, int cin,
    const float* __restrict__ kp, float extent, const float4* __restrict__ pk, FT* __restrict__ wf,
    float* __restrict__ inv_n, int nchunk) {
    constexpr bool BF16 = sizeof(FT) == 2;
    // groups of 4 neighbours whose row reads are in flight together.  The kernel is bound by its dependent load
    // chain, so occupancy beats deeper batches: 4 groups (76 VGPRs, 6 wavefronts/SIMD) measured best for NB = 1
    // (8 groups: 96 VGPRs / 5 waves, -9 %; 12 groups: 136 / 3 waves, -40 %; 2 groups: 68 / 7 waves, -4 %)
    constexpr int STEPS = NB >= 4 ? 2 : 4;
    const int lane = threadIdx.x & 63;
    const int hsub = lane >> 4, j = lane & 15;
    // the wavefront index is uniform: telling the compiler so moves the item's query, its coordinates and the row
    // base addresses into SGPRs (fewer VGPRs -> more wavefronts per SIMD, which is what this kernel is short of)
    const long gw = (long)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long nw = (long)gridDim.x * kWavesPerBlock;
    const long items = (long)nq * nchunk;
    const bool jvalid = j < K;
    const float kpx = jvalid ? kp[3 * j] : 0.f, kpy = jvalid ? kp[3 * j + 1] : 0.f, kpz = jvalid ? kp[3 * j + 2] : 0.f;
    const float inv_extent = 1.0f / extent;

    for (long item = gw; item < items; item += nw) {
        const int q = (int)(item / nchunk), chunk = (int)(item - (long)q * nchunk);
        const int c0 = chunk * 64 * NB;
        const float qx = q_pts[3 * (long)q], qy = q_pts[3 * (long)q + 1], qz = q_pts[3 * (long)q + 2];
        f32x4 acc[NB][4];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[b][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int npos = 0;
        float ta0 = 0.f, ta1 = 0.f, ta2 = 0.f, ta3 = 0.f;             // TAIL: this lane's (kernel point j, neighbour group hsub) share
        for (int hc = 0; hc < H; hc += 64) {
            // lanes = neighbours: index + centred coordinates, once per 64 neighbours
            const int h = hc + lane;
            const long long iv = idx[(long)q * ld_idx + (h < H ? h : H - 1)];   // branch-free, clamped
            const int i = (h < H && iv >= 0 && iv < ns) ? (int)iv : -1;
            const long ic = i >= 0 ? i : 0;
            const float4 sp = pk[ic];                                      // coordinates + row-positive flag
            const float px = sp.x - qx, py = sp.y - qy, pz = sp.z - qz;
            if (chunk == 0) npos += __popcll(__ballot(i >= 0 && sp.w != 0.f));
            // the table pads a short row with shadow entries at its END: the groups behind the last real neighbour would
            // load (a clamped row), take 15 square roots and multiply by zero weights -- the loop stops at the last real one
            // (round 6; a row usually holds fewer neighbours than the table is wide: the limits sit at a high percentile)
            const unsigned long long real_lanes = __ballot(i >= 0);
            const int hn_all = H - hc < 64 ? H - hc : 64;
            const int hn_real = real_lanes ? 64 - (int)__clzll(real_lanes) : 0;
            const int hn = hn_real < hn_all ? hn_real : hn_all;
            // STEPS groups of 4 neighbours at a time: all their row reads are issued before the first
            // MFMA needs one (STEPS x NB KiB in flight per wavefront)
            for (int h0 = 0; h0 < hn; h0 += 4 * STEPS) {
                float w[STEPS];
                float4 v[STEPS][NB];
                float4 vt[STEPS];
#pragma unroll
                for (int s = 0; s < STEPS; ++s) {
                    const int src = h0 + 4 * s + hsub;            // <= 63
                    const int ii = __shfl(i, src, 64);
                    const float nx = __shfl(px, src, 64), ny = __shfl(py, src, 64), nz = __shfl(pz, src, 64);
                    const bool real = ii >= 0 && h0 + 4 * s < hn;
                    w[s] = 0.f;
                    if (real && jvalid) {
                        const float dx = nx - kpx, dy = ny - kpy, dz = nz - kpz;
                        w[s] = fmaxf(1.0f - __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz) * inv_extent, 0.0f);
                    }
                    // unconditional loads from clamped (always valid) addresses, zeroed by select: a load
                    // inside a divergent branch would be waited for one by one
                    const FT* xrow = x + (long)(real ? ii : 0) * cin;
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const int c = c0 + 64 * b + 4 * j;
                        const float4 t = load4(xrow + (c < cin ? c : cin - 4));
                        const bool ok = real && c < cin;
                        v[s][b] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
                    }
                    if constexpr (TAIL) {
                        const float4 t = load4(xrow + 64 * NB);
                        vt[s] = make_float4(real ? t.x : 0.f, real ? t.y : 0.f, real ? t.z : 0.f, real ? t.w : 0.f);
                    }
                }
#pragma unroll
                for (int s = 0; s < STEPS; ++s)
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], v[s][b].x, acc[b][0], 0, 0, 0);
                        acc[b][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], v[s][b].y, acc[b][1], 0, 0, 0);
                        acc[b][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], v[s][b].z, acc[b][2], 0, 0, 0);
                        acc[b][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], v[s][b].w, acc[b][3], 0, 0, 0);
                    }
                if constexpr (TAIL) {
#pragma unroll
                    for (int s = 0; s < STEPS; ++s) {
                        ta0 += w[s] * vt[s].x;
                        ta1 += w[s] * vt[s].y;
                        ta2 += w[s] * vt[s].z;
                        ta3 += w[s] * vt[s].w;
                    }
                }
            }
        }
        // D layout: register r of lane (hsub, j) = kernel point 4*hsub + r, channel group j
        FT* o = wf + (long)q * K * cin + c0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int c = 64 * b + 4 * j;
            if (c0 + c >= cin) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * hsub + r;
                if (k < K) {
                    // streamed once (the contraction GEMM reads it back): non-temporal, so that 230 MB of wf do not
                    // push the feature rows the gathers re-read out of L2
                    if constexpr (BF16) {
                        typedef unsigned v2u __attribute__((ext_vector_type(2)));
                        const v2u val = {(unsigned)bf16_rne(acc[b][0][r]) | ((unsigned)bf16_rne(acc[b][1][r]) << 16),
                                         (unsigned)bf16_rne(acc[b][2][r]) | ((unsigned)bf16_rne(acc[b][3][r]) << 16)};
                        __builtin_nontemporal_store(val, reinterpret_cast<v2u*>(o + (long)k * cin + c));
                    } else {
                        typedef float v4f __attribute__((ext_vector_type(4)));
                        const v4f val = {acc[b][0][r], acc[b][1][r], acc[b][2][r], acc[b][3][r]};
                        if (NT_STORE) __builtin_nontemporal_store(val, reinterpret_cast<v4f*>(o + (long)k * cin + c));
                        else *reinterpret_cast<v4f*>(o + (long)k * cin + c) = val;
                    }
                }
            }
        }
        if constexpr (TAIL) {
            ta0 += __shfl_xor(ta0, 16, 64); ta1 += __shfl_xor(ta1, 16, 64); ta2 += __shfl_xor(ta2, 16, 64); ta3 += __shfl_xor(ta3, 16, 64);
            ta0 += __shfl_xor(ta0, 32, 64); ta1 += __shfl_xor(ta1, 32, 64); ta2 += __shfl_xor(ta2, 32, 64); ta3 += __shfl_xor(ta3, 32, 64);
            if (hsub == 0 && jvalid) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f val = {ta0, ta1, ta2, ta3};
                v4f* dst = reinterpret_cast<v4f*>(reinterpret_cast<float*>(wf) + (long)q * K * cin + (long)j * cin + 64 * NB);
                if (NT_STORE) __builtin_nontemporal_store(val, dst);
                else *dst = val;
            }
        }
        if (chunk == 0 && lane == 0) inv_n[q] = 1.0f / (float)(npos > 1 ? npos : 1);
    }
}

// Cin == 1 (first layer: the feature is a column of ones): wf[q,k] = sum_h w[q,h,k] * x[idx[q,h]].
// There is no channel dimension to spread over lanes and only Nq/64 wavefronts of thread-per-query work
// (fewer than the chip has SIMDs), so the kernel is bound by the idx -> coordinate load chain.  Four
// lanes share a query: each takes every fourth neighbour, C1_BATCH of them at a time with all index
// loads, then all gathers, in flight together; the 15 partial sums are combined with two butterfly
// shuffles.  Kernel points live in SGPRs, loads are branch-free (clamped addresses + select).
constexpr int C1_PARTS = 4, C1_BATCH = 6;
__global__ void __launch_bounds__(256) k_kpconv_c1(
    const float* __restrict__ q_pts, int nq, const float* __restrict__ s_pts, int ns,
    const long long* __restrict__ idx, int H, int ld_idx, const float4* __restrict__ pk, const float* __restrict__ kp,
    float extent, float* __restrict__ wf, float* __restrict__ inv_n, int ld_wf) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int part = t & (C1_PARTS - 1);
    const int qraw = t / C1_PARTS;
    const int q = qraw < nq ? qraw : nq - 1;          // surplus lanes shadow the last query (shuffles stay uniform)
    float kpx[K], kpy[K], kpz[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { kpx[k] = kp[3 * k]; kpy[k] = kp[3 * k + 1]; kpz[k] = kp[3 * k + 2]; }
    const float inv_extent = 1.0f / extent;
    const float qx = q_pts[3 * (long)q], qy = q_pts[3 * (long)q + 1], qz = q_pts[3 * (long)q + 2];
    float acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.f;
    int npos = 0;
    const long long* row = idx + (long)q * ld_idx;
    for (int h0 = 0; h0 < H; h0 += C1_PARTS * C1_BATCH) {
        long long iv[C1_BATCH];
#pragma unroll
        for (int b = 0; b < C1_BATCH; ++b) {
            const int h = h0 + b * C1_PARTS + part;
            iv[b] = row[h < H ? h : H - 1];
        }
        float nx[C1_BATCH], ny[C1_BATCH], nz[C1_BATCH], xv[C1_BATCH];
#pragma unroll
        for (int b = 0; b < C1_BATCH; ++b) {
            const int h = h0 + b * C1_PARTS + part;
            const bool real = h < H && iv[b] >= 0 && iv[b] < ns;
            const long ic = real ? iv[b] : 0;
            const float4 p = pk[ic];                                     // one 16-byte gather: coordinates + feature
            nx[b] = p.x - qx;
            ny[b] = p.y - qy;
            nz[b] = p.z - qz;
            const float v = p.w;
            xv[b] = real ? v : 0.f;
        }
#pragma unroll
        for (int b = 0; b < C1_BATCH; ++b) {
            npos += xv[b] > 0.0f ? 1 : 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float dx = nx[b] - kpx[k], dy = ny[b] - kpy[k], dz = nz[b] - kpz[k];
                acc[k] = fmaf(fmaxf(1.0f - __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz) * inv_extent, 0.0f),
                              xv[b], acc[k]);
            }
        }
    }
#pragma unroll
    for (int d = 1; d < C1_PARTS; d <<= 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] += __shfl_xor(acc[k], d, 64);
        npos += __shfl_xor(npos, d, 64);
    }
    if (qraw >= nq) return;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if ((k & (C1_PARTS - 1)) == part) wf[(long)q * ld_wf + k] = acc[k];
    if (ld_wf > K && part == C1_PARTS - 1) wf[(long)q * ld_wf + K] = 0.f;      // rows of 16 floats: the contraction's k-steps are whole
    if (part == 0) inv_n[q] = 1.0f / (float)(npos > 1 ? npos : 1);
}

// Scalar fallback for channel counts that are not a multiple of 4: weights staged in LDS, lanes =
// channels.
__global__ void __launch_bounds__(kWavesPerBlock * 64) k_kpconv_generic(
    const float* __restrict__ q_pts, int nq, const float* __restrict__ s_pts, int ns,
    const long long* __restrict__ idx, int H, int ld_idx, const float* __restrict__ x, int cin,
    const float* __restrict__ kp, float extent, const unsigned char* __restrict__ pos, float* __restrict__ wf,
    float* __restrict__ inv_n) {
    __shared__ __attribute__((aligned(16))) float s_w[kWavesPerBlock][64][16];
    __shared__ int s_idx[kWavesPerBlock][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * kWavesPerBlock + wave, nw = gridDim.x * kWavesPerBlock;
    const float inv_extent = 1.0f / extent;
    for (int q = gw; q < nq; q += nw) {
        const float qx = q_pts[3 * (long)q], qy = q_pts[3 * (long)q + 1], qz = q_pts[3 * (long)q + 2];
        int npos = 0;
        for (int cbase = 0; cbase < cin; cbase += 64) {
            float acc[K];
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] = 0.0f;
            for (int hc = 0; hc < H; hc += 64) {
                const int h = hc + lane;
                int i = -1;
                if (h < H) {
                    const long long v = idx[(long)q * ld_idx + h];
                    i = (v >= 0 && v < ns) ? (int)v : -1;
                }
                __builtin_amdgcn_wave_barrier();
                if (i >= 0) {
                    const float nx = s_pts[3 * (long)i] - qx, ny = s_pts[3 * (long)i + 1] - qy,
                                nz = s_pts[3 * (long)i + 2] - qz;
                    for (int k = 0; k < K; ++k) {
                        const float dx = nx - kp[3 * k], dy = ny - kp[3 * k + 1], dz = nz - kp[3 * k + 2];
                        s_w[wave][lane][k] = fmaxf(1.0f - sqrtf(dx * dx + dy * dy + dz * dz) * inv_extent, 0.0f);
                    }
                }
                s_idx[wave][lane] = i;
                if (cbase == 0) npos += __popcll(__ballot(i >= 0 && pos[i >= 0 ? i : 0] != 0));
                __builtin_amdgcn_wave_barrier();
                const int hn = H - hc < 64 ? H - hc : 64;
                const int c = cbase + lane;
                for (int hh = 0; hh < hn; ++hh) {
                    const int ii = s_idx[wave][hh];
                    if (ii < 0) continue;
                    const float xv = c < cin ? x[(long)ii * cin + c] : 0.0f;
#pragma unroll
                    for (int k = 0; k < K; ++k) acc[k] = fmaf(s_w[wave][hh][k], xv, acc[k]);
                }
            }
            const int c = cbase + lane;
            if (c < cin)
#pragma unroll
                for (int k = 0; k < K; ++k) wf[(long)q * K * cin + (long)k * cin + c] = acc[k];
        }
        if (lane == 0) inv_n[q] = 1.0f / (float)(npos > 1 ? npos : 1);
    }
}

}  // namespace

float4* kpconv_pk_ptr(void* ws, size_t ws_bytes, int ns) {
    Carver cv(ws, ws_bytes);
    cv.take<unsigned char>((size_t)ns + 1);
    float4* pk = cv.take<float4>((size_t)ns + 1);
    return cv.ok() ? pk : nullptr;
}

// pos / pk records of `x` (see k_row_positive) into the workspace layout of pcrcg_kpconv_ws_bytes
int kpconv_pack(const float* x, int ns, int cin, const float* s_pts, void* ws, size_t ws_bytes, hipStream_t st,
                unsigned short* x_bf16) {
    Carver cv(ws, ws_bytes);
    unsigned char* pos = cv.take<unsigned char>((size_t)ns + 1);
    float4* pk = cv.take<float4>((size_t)ns + 1);
    PCRCG_CHECK_WS(cv);
    if (ns > 0) hipLaunchKernelGGL(k_row_positive, dim3((ns + 3) / 4), dim3(256), 0, st, x, ns, cin, s_pts, pos, pk, x_bf16);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

}  // namespace pcrcg

using namespace pcrcg;

extern "C" {

void pcrcg_profile_kpconv(int enable) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    g_prof_on = enable;
}

int pcrcg_profile_kpconv_read(float* ms, int* nq, int* h, int* cin, int* cout, int* kind, int cap) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    int n = 0;
    for (auto& r : g_prof) {
        if (n >= cap) break;
        if (hipEventSynchronize(r.b) != hipSuccess) return -1;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return -1;
        ms[n] = t; nq[n] = r.nq; h[n] = r.h; cin[n] = r.cin; cout[n] = r.cout; kind[n] = r.kind;
        ++n;
    }
    return n;
}

size_t pcrcg_kpconv_ws_bytes(int ns) {
    const size_t n = (size_t)(ns > 0 ? ns : 0) + 1;
    return carve_bytes(n, 1) + carve_bytes(n, sizeof(float4));     // row-positive flags + packed (x, y, z, flag) records
}

int pcrcg_kpconv_aggregate(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx,
                           int h, int ld_idx, const float* x, int cin, const float* kp, float extent,
                           float* wf, float* inv_n, void* ws, size_t ws_bytes, void* stream) {
    return kpconv_aggregate_rows(q_pts, nq, s_pts, ns, idx, h, ld_idx, x, cin, kp, extent, wf, inv_n, ws, ws_bytes,
                                 as_stream(stream), true, true, 0);
}

// The bf16 feature-storage variant of pcrcg_kpconv_aggregate: x stays fp32 at the boundary; `x_bf16` ([ns, cin] u16
// scratch) receives its round-to-nearest-even copy, which is what the neighbour gathers read, and `wf_bf16`
// ([nq, 15*cin] u16) the aggregated features rounded the same way.  Geometry, influences, the neighbour-count
// normaliser and the accumulation are fp32 as in the fp32 path.
int pcrcg_kpconv_aggregate_bf16(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx, int h,
                                int ld_idx, const float* x, int cin, const float* kp, float extent, void* x_bf16,
                                void* wf_bf16, float* inv_n, void* ws, size_t ws_bytes, void* stream) {
    return kpconv_aggregate_bf16(q_pts, nq, s_pts, ns, idx, h, ld_idx, x, static_cast<unsigned short*>(x_bf16), cin, kp,
                                 extent, static_cast<unsigned short*>(wf_bf16), inv_n, ws, ws_bytes, as_stream(stream));
}
}

namespace pcrcg {

// pcrcg_kpconv_aggregate for a slice of the queries; `pack` = (re)build the support records in `ws` first (once per
// layer), `stream_out` = non-temporal wf stores (the whole layer's wf is streamed once through HBM) or plain stores
// (a row chunk whose wf the contraction reads back from L2 / Infinity Cache right away)
int kpconv_aggregate_rows(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx, int h, int ld_idx,
                          const float* x, int cin, const float* kp, float extent, float* wf, float* inv_n, void* ws,
                          size_t ws_bytes, hipStream_t st, bool pack, bool stream_out, int c1_ld) {
    PCRCG_CHECK_ARG(nq >= 0 && ns >= 0 && h >= 1 && ld_idx >= h && cin >= 1 && (c1_ld == 0 || c1_ld == K || c1_ld == K + 1));
    PCRCG_CHECK_ARG(extent > 0.0f);
    if (nq == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(q_pts && s_pts && idx && x && kp && wf && inv_n && ws);
    PCRCG_CHECK_ARG(ns >= 1);
    Carver cv(ws, ws_bytes);
    unsigned char* pos = cv.take<unsigned char>((size_t)ns + 1);
    float4* pk = cv.take<float4>((size_t)ns + 1);
    PCRCG_CHECK_WS(cv);
    const long long* idx_ll = reinterpret_cast<const long long*>(idx);
    const int max_blocks = 256 * 32;
    auto blocks_for = [&](long waves) {
        long b = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
        return (int)(b > max_blocks ? max_blocks : b);
    };
    if (cin == 1) {
        if (pack) hipLaunchKernelGGL(k_pack_c1, dim3((ns + 255) / 256), dim3(256), 0, st, x, ns, s_pts, pk);
        KpProfScope prof_scope(st, nq, h, cin, 0, 0);
        hipExtLaunchKernelGGL(k_kpconv_c1, dim3((int)(((long)nq * C1_PARTS + 255) / 256)), dim3(256), 0, st, prof_scope.a,
                              prof_scope.b, 0, q_pts, nq, s_pts, ns, idx_ll, h, ld_idx, (const float4*)pk, kp, extent, wf,
                              inv_n, c1_ld > 0 ? c1_ld : K);
        PCRCG_CHECK_LAUNCH();
        return PCRCG_OK;
    }
    if (pack && ns > 0)
        hipLaunchKernelGGL(k_row_positive, dim3((ns + 3) / 4), dim3(256), 0, st, x, ns, cin, s_pts, pos, pk,
                           (unsigned short*)nullptr);
    KpProfScope prof_scope(st, nq, h, cin, 0, 0);   // start / stop events of the gather/aggregate kernel itself
    const bool aligned = (cin % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(wf) & 15) == 0);
    if (!aligned) {
        hipLaunchKernelGGL(k_kpconv_generic, dim3(blocks_for(nq)), dim3(kWavesPerBlock * 64), 0, st, q_pts, nq, s_pts,
                           ns, idx_ll, h, ld_idx, x, cin, kp, extent, pos, wf, inv_n);
        PCRCG_CHECK_LAUNCH();
        return PCRCG_OK;
    }
    // cin = 64 nb + 4 (PCR-CG's 129 channels in rows of 132): one wavefront per query, the last float4 on the vector units
    if (cin == 68 || cin == 132) {
        const int blocks = blocks_for((long)nq);
#define LAUNCH_T(NBV, NT)                                                                                                        \
        hipExtLaunchKernelGGL((k_kpconv_mfma<NBV, NT, float, true>), dim3(blocks), dim3(kWavesPerBlock * 64), 0, st, prof_scope.a, \
                              prof_scope.b, 0, q_pts, nq, s_pts, ns, idx_ll, h, ld_idx, x, cin, kp, extent,                      \
                              (const float4*)pk, wf, inv_n, 1)
        if (cin == 132) { if (stream_out) LAUNCH_T(2, true); else LAUNCH_T(2, false); }
        else { if (stream_out) LAUNCH_T(1, true); else LAUNCH_T(1, false); }
#undef LAUNCH_T
        PCRCG_CHECK_LAUNCH();
        return PCRCG_OK;
    }
    // channel blocks of 64 per wavefront: as many as possible while keeping >= ~16k wavefronts in the grid
    const int nblk = (cin + 63) / 64;
    int nb = nblk >= 4 ? 4 : (nblk >= 2 ? 2 : 1);
    while (nb > 1 && ((long)nq * ((nblk + nb - 1) / nb) < 16384 || nblk % nb != 0)) nb >>= 1;
    const int nchunk = (nblk + nb - 1) / nb;
    const int blocks = blocks_for((long)nq * nchunk);
#define LAUNCH(NBV, NT)                                                                                                 \
    hipExtLaunchKernelGGL((k_kpconv_mfma<NBV, NT, float>), dim3(blocks), dim3(kWavesPerBlock * 64), 0, st, prof_scope.a, \
                          prof_scope.b, 0, q_pts, nq, s_pts, ns, idx_ll, h, ld_idx, x, cin, kp, extent,                 \
                          (const float4*)pk, wf, inv_n, nchunk)
    if (stream_out) {
        if (nb == 4) LAUNCH(4, true);
        else if (nb == 2) LAUNCH(2, true);
        else LAUNCH(1, true);
    } else {
        if (nb == 4) LAUNCH(4, false);
        else if (nb == 2) LAUNCH(2, false);
        else LAUNCH(1, false);
    }
#undef LAUNCH
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

// The bf16 feature-storage variant: x_bf16 [ns, cin] and wf_bf16 [nq, 15*cin] are bf16; `x` (fp32) is only read by the
// support-record prelude (the n_q normaliser counts rows with a positive fp32 sum, exactly as the fp32 path does).
int kpconv_aggregate_bf16(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx, int h, int ld_idx,
                          const float* x, unsigned short* x_bf16, int cin, const float* kp, float extent,
                          unsigned short* wf_bf16, float* inv_n, void* ws, size_t ws_bytes, hipStream_t st) {
    PCRCG_CHECK_ARG(nq >= 0 && ns >= 1 && h >= 1 && ld_idx >= h && cin >= 4 && cin % 4 == 0 && extent > 0.0f);
    if (nq == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(q_pts && s_pts && idx && x && x_bf16 && kp && wf_bf16 && inv_n && ws);
    PCRCG_CHECK_ARG((reinterpret_cast<uintptr_t>(x_bf16) & 7) == 0 && (reinterpret_cast<uintptr_t>(wf_bf16) & 7) == 0);
    PCRCG_PROPAGATE(kpconv_pack(x, ns, cin, s_pts, ws, ws_bytes, st, x_bf16));   // also writes the bf16 copy of x
    Carver cv(ws, ws_bytes);
    cv.take<unsigned char>((size_t)ns + 1);
    float4* pk = cv.take<float4>((size_t)ns + 1);
    const long long* idx_ll = reinterpret_cast<const long long*>(idx);
    const int nblk = (cin + 63) / 64;
    int nb = nblk >= 4 ? 4 : (nblk >= 2 ? 2 : 1);
    while (nb > 1 && ((long)nq * ((nblk + nb - 1) / nb) < 16384 || nblk % nb != 0)) nb >>= 1;
    const int nchunk = (nblk + nb - 1) / nb;
    long blocks = ((long)nq * nchunk + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > 256 * 32) blocks = 256 * 32;
    KpProfScope prof_scope(st, nq, h, cin, 0, 2);      // kind 2: bf16 storage
#define LAUNCHB(NBV)                                                                                                      \
    hipExtLaunchKernelGGL((k_kpconv_mfma<NBV, true, unsigned short>), dim3((int)blocks), dim3(kWavesPerBlock * 64), 0, st, \
                          prof_scope.a, prof_scope.b, 0, q_pts, nq, s_pts, ns, idx_ll, h, ld_idx, x_bf16, cin, kp, extent, \
                          (const float4*)pk, wf_bf16, inv_n, nchunk)
    if (nb == 4) LAUNCHB(4);
    else if (nb == 2) LAUNCHB(2);
    else LAUNCHB(1);
#undef LAUNCHB
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

}  // namespace pcrcg
