// queue_pipes.hip -- which HIP streams share a hardware dispatcher (a compute pipe of the command processor)?
// For every ordered pair (i, j) of S streams: a dispatch-bound kernel (very many tiny workgroups: the dispatcher hands them
// out for several milliseconds) runs on stream i; 200 us later a one-workgroup kernel is launched on stream j and the host
// times how long it takes to complete.  A pair whose tiny kernel waits for the big dispatch shares a dispatcher.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/micro/queue_pipes.hip -o scripts/micro/queue_pipes ; run: queue_pipes [S]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_many(int* sink, int spin) {
    int x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1664525 + 1013904223;
    if (x == 0x7fffffff) sink[0] = x;
}
__global__ void k_tiny(int* sink) { if (threadIdx.x == 1234567) sink[1] = 1; }
// a kernel that OCCUPIES the chip: few, long workgroups (dispatch finishes at once)
__global__ void k_long(int* sink, int spin) {
    int x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1664525 + 1013904223;
    if (x == 0x7fffffff) sink[2] = x;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 8;
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES=%s, %d streams\n", q ? q : "(unset: 4)", S);
    int* sink;
    CK(hipMalloc(&sink, 64));
    std::vector<hipStream_t> st(S);
    for (int i = 0; i < S; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    for (int i = 0; i < S; ++i) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st[i], sink); }
    CK(hipDeviceSynchronize());
    // calibrate: the big dispatch alone
    const int blocks = 2000000;
    double t0 = now_us();
    hipLaunchKernelGGL(k_many, dim3(blocks), dim3(64), 0, st[0], sink, 64);
    CK(hipStreamSynchronize(st[0]));
    const double big_us = now_us() - t0;
    t0 = now_us();
    hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st[0], sink);
    CK(hipStreamSynchronize(st[0]));
    printf("dispatch-bound kernel alone: %.0f us (%d workgroups of 64); tiny kernel alone: %.0f us\n", big_us, blocks, now_us() - t0);
    printf("rows: stream of the dispatch-bound kernel; columns: stream of the tiny kernel; entry: tiny kernel's completion time in us\n     ");
    for (int j = 0; j < S; ++j) printf("%7d", j);
    printf("\n");
    for (int i = 0; i < S; ++i) {
        printf("%3d: ", i);
        for (int j = 0; j < S; ++j) {
            if (i == j) { printf("      -"); continue; }
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(k_many, dim3(blocks), dim3(64), 0, st[i], sink, 64);
                std::this_thread::sleep_for(std::chrono::microseconds(300));
                const double a = now_us();
                hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st[j], sink);
                CK(hipStreamSynchronize(st[j]));
                const double d = now_us() - a;
                best = d < best ? d : best;
                CK(hipDeviceSynchronize());
            }
            printf("%7.0f", best);
        }
        printf("\n");
    }
    // the same with a chip-filling kernel of few long workgroups (its dispatch is over at once): does a tiny kernel on another
    // stream get in?  (resources, not the dispatcher, would be what it waits for)
    printf("a chip-filling kernel of 2048 long workgroups on stream 0 (%s), tiny kernel on stream j:\n", "dispatch over at once");
    for (int j = 1; j < S; ++j) {
        hipLaunchKernelGGL(k_long, dim3(2048), dim3(256), 0, st[0], sink, 400000);
        std::this_thread::sleep_for(std::chrono::microseconds(300));
        const double a = now_us();
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st[j], sink);
        CK(hipStreamSynchronize(st[j]));
        printf("  j=%d: %.0f us", j, now_us() - a);
        CK(hipDeviceSynchronize());
    }
    printf("\n");
    return 0;
}
