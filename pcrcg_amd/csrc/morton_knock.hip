// morton_knock.hip -- MEASUREMENT AID ONLY (DebugOpts::pyr_morton, PCRCG_DEBUG=pyr_morton=1; off by default).
//
// Round 5's review asked what an internal, spatially coherent point order would be worth before anybody builds it: the rows
// of a subsampled level come out in libstdc++'s hash order, i.e. spatially random, and every gather of the levels below
// (KPConv, max-pool, the upsampling products) pays for that.  This knock-out sorts every subsampled level along a Z-order
// curve (per cloud) right after the subsampling, BEFORE anything reads the level: grids, searches, tables and the network
// then run on the sorted rows, self-consistently -- the outputs are those of the same network on the same points, but the
// level rows (and with them tie choices and summation orders) are not the reference's, so this is a timing experiment,
// not a product path (tests never set the switch; profiles/r06_knock_internal_morton_order.txt holds what it measured).
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace pcrcg {
namespace {

typedef unsigned long long u64;

__device__ __forceinline__ u64 spread3(unsigned v) {       // 19 bits -> every third bit
    u64 x = v & 0x7FFFFu;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

// key = (cloud << 57) | morton(19 bits per axis of (p + 512 m) * 512 / m); rows behind the level's end sort last
__global__ void __launch_bounds__(256) k_morton_keys(const float* __restrict__ pts, const int* __restrict__ len, int nb, int cap,
                                                      u64* __restrict__ key, int* __restrict__ val) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    int b = 0, acc = 0;
    while (b < nb && i >= acc + len[b]) { acc += len[b]; ++b; }
    val[i] = i;
    if (b >= nb) { key[i] = ~0ull; return; }
    auto q = [](float p) { const float v = (p + 512.f) * 512.f; return (unsigned)(v < 0.f ? 0.f : (v > 524287.f ? 524287.f : v)); };
    const u64 m = spread3(q(pts[3 * (long)i])) | (spread3(q(pts[3 * (long)i + 1])) << 1) | (spread3(q(pts[3 * (long)i + 2])) << 2);
    key[i] = ((u64)b << 57) | m;
}
__global__ void __launch_bounds__(256) k_permute_rows(const float* __restrict__ src, const int* __restrict__ perm,
                                                       const int* __restrict__ len, int nb, float* __restrict__ dst) {
    int n = 0;
    for (int b = 0; b < nb; ++b) n += len[b];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long s = perm[i];
    dst[3 * (long)i] = src[3 * s];
    dst[3 * (long)i + 1] = src[3 * s + 1];
    dst[3 * (long)i + 2] = src[3 * s + 2];
}

}  // namespace

size_t morton_knock_ws_bytes(int cap) {
    size_t temp = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const u64*)nullptr, (u64*)nullptr, (const int*)nullptr, (int*)nullptr,
                                             cap > 0 ? cap : 1, 0, 64, (hipStream_t) nullptr);
    const size_t N = (size_t)(cap > 0 ? cap : 1);
    return carve_bytes(temp, 1) + 2 * carve_bytes(N, sizeof(u64)) + 2 * carve_bytes(N, sizeof(int)) + carve_bytes(3 * N, sizeof(float));
}

// pts [<= cap rows of nb clouds, lengths on the device] -> the same rows, every cloud in Z order, in place
int morton_knock_level(float* pts, int cap, const int* len, int nb, void* ws, size_t ws_bytes, hipStream_t st) {
    if (cap <= 0) return PCRCG_OK;
    size_t temp = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const u64*)nullptr, (u64*)nullptr, (const int*)nullptr, (int*)nullptr, cap,
                                             0, 64, st);
    Carver cv(ws, ws_bytes);
    void* tmp = cv.take<char>(temp);
    u64* k0 = cv.take<u64>((size_t)cap);
    u64* k1 = cv.take<u64>((size_t)cap);
    int* v0 = cv.take<int>((size_t)cap);
    int* v1 = cv.take<int>((size_t)cap);
    float* copy = cv.take<float>(3 * (size_t)cap);
    PCRCG_CHECK_WS(cv);
    const int blocks = (cap + 255) / 256;
    hipLaunchKernelGGL(k_morton_keys, dim3(blocks), dim3(256), 0, st, pts, len, nb, cap, k0, v0);
    PCRCG_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, temp, k0, k1, v0, v1, cap, 0, 64, st));
    PCRCG_CHECK_HIP(hipMemcpyAsync(copy, pts, sizeof(float) * 3 * (size_t)cap, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_permute_rows, dim3(blocks), dim3(256), 0, st, copy, v1, len, nb, pts);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

}  // namespace pcrcg
