// mfma_rate.hip -- what limits v_mfma_f32_32x32x2_f32 issue on gfx950?  Standalone microbenchmark:
//   mode 0: N dependent MFMAs on ONE accumulator per wave            (the 64x64-tile GEMM inner loop shape)
//   mode 1: MFMAs alternating over TWO accumulators
//   mode 2: as 0, with one ds_read_b128 + waitcnt between MFMA groups of 4  (operands fed from LDS)
//   mode 3: as 2, with a __syncthreads() every 16 MFMAs                (one barrier per k-step)
//   mode 4: as 3, B operand read as 4 x ds_read_b32 with row stride 68 floats (the [K,N] weight image)
//   mode 5: as 3, plus 4 x ds_write_b128 per thread and step (register -> LDS staging of the next tile)
//   mode 6: as 5, plus 4 x global_load_dwordx4 per thread and step from a 230 MB array (the A / B stream)
//   mode 7: as 3, plus 4 x global_load_lds_dwordx4 per thread and step (LDS-DMA: same bytes as mode 6, no
//           VGPR round trip and no ds_write)
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, const float4* __restrict__ src, long nsrc) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 64 * 36 + 2 * 4608];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4 * 64 * 36; i += 256) lds[i] = 1.0f + (i & 7);
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float a = 1.0f + tid, b = 2.0f;
    const float* p = &lds[(tid & 63) * 36];
    const float* pb = &lds[(tid >> 5 & 1) * 4 * 68 + (tid & 31)];
    float4 stage[4] = {make_float4(a, b, a, b), make_float4(a, b, a, b), make_float4(a, b, a, b), make_float4(a, b, a, b)};
    long gpos = ((long)blockIdx.x * 977 + tid) % (nsrc - 4 * 256 * 2);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 7) {
            float* dst = &lds[4 * 64 * 36 + (it & 1) * 4608];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + gpos + u * 256),
                                                 (__attribute__((address_space(3))) void*)(dst + ((tid >> 6) * 4 + u) * 256), 16, 0, 0);
            gpos += 1024 * 64;
            if (gpos >= nsrc - 2048) gpos -= nsrc - 4096;
        }
        if (MODE == 5 || MODE == 6) {
            float* dst = &lds[4 * 64 * 36 + (it & 1) * 4608];
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(&dst[(tid + u * 256) * 4 % 4604]) = stage[u];
        }
        if (MODE == 6) {
#pragma unroll
            for (int u = 0; u < 4; ++u) stage[u] = src[gpos + u * 256];
            gpos += 1024 * 64;
            if (gpos >= nsrc - 2048) gpos -= nsrc - 4096;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 va = make_float4(a, a, a, a), vb = make_float4(b, b, b, b);
            if (MODE >= 2) {
                va = *reinterpret_cast<const float4*>(p + g * 8);
                if (MODE == 4) vb = make_float4(pb[g * 8 * 68], pb[g * 8 * 68 + 68], pb[g * 8 * 68 + 136], pb[g * 8 * 68 + 204]);
                else vb = *reinterpret_cast<const float4*>(p + g * 8 + 4);
            }
            if (MODE == 1) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va.x, vb.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va.y, vb.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va.z, vb.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va.w, vb.w, acc1, 0, 0, 0);
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va.x, vb.x, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va.y, vb.y, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va.z, vb.z, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va.w, vb.w, acc0, 0, 0, 0);
            }
        }
        if (MODE >= 3) __syncthreads();
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    s += stage[0].x + stage[1].y + stage[2].z + stage[3].w;
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(int blocks_per_cu, int iters) {
    const int blocks = 256 * blocks_per_cu;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    static float4* src = nullptr;
    const long nsrc = 230l * 1000 * 1000 / 16;
    if (!src) { hipMalloc(&src, nsrc * 16); hipMemset(src, 0, nsrc * 16); }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, src, nsrc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, src, nsrc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * iters * 16;
    const double tf = mfmas * 4096 / (ms * 1e-3) / 1e12;
    printf("mode %d  %d block(s)/CU (%d waves/SIMD): %.3f ms  %.1f TFLOP/s  (%.0f%% of 157)\n", MODE, blocks_per_cu,
           blocks_per_cu, ms, tf, tf / 157.3 * 100);
    hipFree(out);
}

int main() {
    for (int bpc : {1, 2, 4}) {
        run<0>(bpc, 2000);
        run<1>(bpc, 2000);
        run<2>(bpc, 2000);
        run<3>(bpc, 2000);
        run<4>(bpc, 2000);
        run<5>(bpc, 2000);
        run<6>(bpc, 2000);
        run<7>(bpc, 2000);
    }
    return 0;
}
