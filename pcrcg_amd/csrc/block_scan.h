// block_scan.h -- wave/block prefix-sum helpers for 64-wide wavefronts (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace pcrcg {

__device__ __forceinline__ int wave_incl_scan_i32(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive scan of one int per thread over a block of THREADS threads (multiple of 64).
// smem must hold THREADS/64 ints.  Contains two __syncthreads(); all threads must call it.
template <int THREADS>
__device__ __forceinline__ int block_excl_scan_i32(int v, int* total, int* smem) {
    constexpr int W = THREADS / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = wave_incl_scan_i32(v, lane);
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    int wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        int s = smem[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return wave_off + inc - v;
}

__device__ __forceinline__ int aload(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace pcrcg
