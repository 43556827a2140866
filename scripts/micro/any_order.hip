// any_order.hip -- probe (GPU box): does hipExtAnyOrderLaunch let two independent kernels of ONE stream overlap on gfx950?
// Two small-grid spin kernels back to back: ~1x the single time if they overlap, ~2x if the second waits for the first.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long cycles, int* out) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
int main() {
    int* d; hipMalloc(&d, 4);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const long cyc = 100000 * 100;   // wall_clock64 ticks at 100 MHz: 100 ms?  (calibrated below by the single run)
    for (int mode = 0; mode < 3; ++mode) {
        hipStreamSynchronize(st);
        hipEventRecord(a, st);
        if (mode == 0) {
            hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, st, nullptr, nullptr, 0, 200000L, d);
        } else {
            hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, st, nullptr, nullptr, 0, 200000L, d);
            hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, st, nullptr, nullptr, mode == 2 ? hipExtAnyOrderLaunch : 0, 200000L, d);
        }
        hipEventRecord(b, st);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s: %.3f ms\n", mode == 0 ? "one kernel" : mode == 1 ? "two kernels, in order" : "two kernels, second any-order", ms);
    }
    (void)cyc;
    return 0;
}
