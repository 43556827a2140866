// cu_mask.hip -- what hipExtStreamCreateWithCUMask does on gfx950 (256 CUs in 8 XCDs): a compute-bound kernel of 8192 one-wave
// workgroups on streams whose mask enables the lowest K bits / every second bit / the bits of one 32-bit word; reports the
// run time, the number of distinct (XCC, SE, CU) places the workgroups ran on and how they spread over the XCDs.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/micro/cu_mask.hip -o scripts/micro/cu_mask
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>

__global__ void k_spin(unsigned* where, int spin) {
    unsigned x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1664525u + 1013904223u;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) where[blockIdx.x] = (hw & 0xffffu) | ((xcc & 0xfu) << 16) | (x == 12345u ? 1u << 31 : 0u);
}
static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int blocks = 8192;
    unsigned* d;
    CK(hipMalloc(&d, blocks * 4));
    std::vector<unsigned> h(blocks);
    auto run = [&](const char* name, const std::vector<uint32_t>& mask) -> int {
        hipStream_t st;
        if (mask.empty()) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        else CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
        hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(64), 0, st, d, 1000);
        CK(hipStreamSynchronize(st));
        const double t0 = now_us();
        hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(64), 0, st, d, 200000);
        CK(hipStreamSynchronize(st));
        const double us = now_us() - t0;
        CK(hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost));
        std::set<unsigned> places;
        int per_xcc[16] = {};
        for (unsigned v : h) {
            const unsigned cu = (v >> 8) & 0xf, sh = (v >> 12) & 1, se = (v >> 13) & 7, xcc = (v >> 16) & 0xf;
            places.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
            per_xcc[xcc]++;
        }
        printf("%-34s %9.0f us  %4zu places  per XCC:", name, us, places.size());
        for (int x = 0; x < 8; ++x) printf(" %5d", per_xcc[x]);
        printf("\n");
        CK(hipStreamDestroy(st));
        return 0;
    };
    if (run("no mask", {})) return 1;
    for (int k : {256, 224, 192, 128, 64, 32, 16, 8}) {
        std::vector<uint32_t> m(8, 0);
        for (int i = 0; i < k; ++i) m[i / 32] |= 1u << (i % 32);
        char name[64];
        snprintf(name, sizeof name, "lowest %d bits", k);
        if (run(name, m)) return 1;
    }
    { std::vector<uint32_t> m(8, 0x55555555u); if (run("every second bit (128)", m)) return 1; }
    { std::vector<uint32_t> m(8, 0); m[3] = 0xffffffffu; if (run("word 3 only (32)", m)) return 1; }
    { std::vector<uint32_t> m(8, 0xffffffffu); m[0] = 0; if (run("all but word 0 (224)", m)) return 1; }
    { std::vector<uint32_t> m(8, 0xffffff00u); if (run("all but bits 0-7 of every word", m)) return 1; }
    return 0;
}
