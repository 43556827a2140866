// load_latency.hip -- measurement aid (GPU box): what does ONE dependent global load cost a wavefront as a function of
// how many workgroups are resident, when nothing else loads the memory system?  (k_radius_cells measured 18 000 cycles
// per load with 1536 workgroups of 256 threads resident and 2 000 with 256: this isolates it.)
//   hipcc --offload-arch=gfx950 -O3 load_latency.hip -o /tmp/ll && /tmp/ll
// Each wavefront: ITER times { one 16-byte load per lane at a pseudo-random, data-dependent address; [barrier] }.
// Prints the average shader cycles (s_memtime) per iteration for grids of 256..2048 workgroups, for vector and scalar
// (wave-uniform address) loads, with and without a workgroup barrier, with 24 KB and 0 KB of LDS per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE, int LDS>   // MODE 0: vector load, no barrier; 1: vector load + __syncthreads; 2: uniform (scalar) load + __syncthreads
__global__ void __launch_bounds__(256) k_lat(const uint4* __restrict__ buf, unsigned mask, int iters, unsigned long long* out) {
    __shared__ char pad[LDS > 0 ? LDS : 4];
    if (LDS > 0 && threadIdx.x == 0) pad[blockIdx.x % LDS] = 1;
    unsigned idx = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    if (MODE == 2) idx = __builtin_amdgcn_readfirstlane(blockIdx.x * 2654435761u);
    const long long t0 = (long long)__builtin_readcyclecounter();
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (MODE != 3 || threadIdx.x < 27) v = buf[idx & mask];      // MODE 3: 27 lanes of wave 0 only, everybody waits
        acc += v.x;
        idx = idx * 1664525u + 1013904223u + v.y;     // v.y is 0: the address depends on the data without leaving the buffer pattern
        if (MODE == 2) idx = __builtin_amdgcn_readfirstlane(idx);
        if (MODE >= 1) __syncthreads();
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)(t1 - t0));
    if (acc == 0x12345678u) out[1] = pad[0];
}

template <int MODE, int LDS>
void run(const uint4* buf, unsigned mask, unsigned long long* d_out, const char* tag) {
    const int iters = 64;
    printf("%-44s", tag);
    for (int wgs : {256, 512, 1024, 1536, 2048}) {
        hipMemset(d_out, 0, 16);
        hipLaunchKernelGGL((k_lat<MODE, LDS>), dim3(wgs), dim3(256), 0, 0, buf, mask, iters, d_out);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipMemset(d_out, 0, 16);
        hipEventRecord(a);
        hipLaunchKernelGGL((k_lat<MODE, LDS>), dim3(wgs), dim3(256), 0, 0, buf, mask, iters, d_out);
        hipEventRecord(b);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        unsigned long long h = 0;
        hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
        printf("  %4d wg: %6.0f cyc/iter (%5.1f us)", wgs, (double)h / ((double)wgs * 4 * iters), ms * 1e3);
    }
    printf("\n");
}

int main() {
    const size_t n = 1 << 22;                    // 4 M records of 16 B = 64 MB
    uint4* buf;
    unsigned long long* d_out;
    hipMalloc(&buf, n * sizeof(uint4));
    hipMemset(buf, 0, n * sizeof(uint4));
    hipMalloc(&d_out, 16);
    for (unsigned mask : {(1u << 16) - 1, (1u << 22) - 1}) {      // 1 MB (L2-resident) and 64 MB working sets
        printf("== working set %u MB\n", (mask + 1) / 65536);
        run<0, 0>(buf, mask, d_out, "vector load, no barrier, no LDS");
        run<1, 0>(buf, mask, d_out, "vector load + barrier, no LDS");
        run<1, 24576>(buf, mask, d_out, "vector load + barrier, 24 KB LDS (6 wg/CU)");
        run<2, 0>(buf, mask, d_out, "uniform load + barrier, no LDS");
        run<2, 24576>(buf, mask, d_out, "uniform load + barrier, 24 KB LDS (6 wg/CU)");
        run<3, 24576>(buf, mask, d_out, "27 lanes of wave 0 load + barrier, 24 KB LDS");
    }
    return 0;
}
