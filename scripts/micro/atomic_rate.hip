// GPU box: how fast does the L2 take float atomic adds?  The KPConv backward scatters 80 M of them per level-0 product
// (60 000 queries x 43 neighbours x 32 channels) and takes 380 us whatever computes the addends (VALU or MFMA kernel).
// A wavefront adds `width` consecutive floats to each of a list of pseudo-random rows of a [rows][width] table: the same
// access shape.  Prints lane-atomics per second for atomics, and the same pattern as plain loads / stores for scale.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip && ./atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(256) k(float* table, const int* rows, int n_rows_list, int width, int per_wave) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int sub = lane / width, c = lane % width;           // 64 / width rows per instruction
    const int rpi = 64 / width;
    float acc = 0.f;
    for (int i = 0; i < per_wave; i += rpi) {
        const long at = wave * per_wave + i + sub;
        if (at >= n_rows_list) break;
        const long off = (long)rows[at] * width + c;
        if (MODE == 0) atomicAdd(table + off, 1.0f);
        else if (MODE == 1) table[off] = 1.0f;
        else acc += table[off];
    }
    if (MODE == 2 && acc == 12345.f) table[0] = acc;
}
int main() {
    const int n_table = 60000;
    const long n_list = 60000L * 43;
    for (int width : {16, 32, 64}) {
        float* table; int* rows;
        hipMalloc(&table, sizeof(float) * n_table * width);
        hipMemset(table, 0, sizeof(float) * n_table * width);
        std::vector<int> h(n_list);
        unsigned s = 12345;
        for (long i = 0; i < n_list; ++i) {
            // rows near the "query" (i / 43), as neighbour lists are: a window of +-2000 rows
            s = s * 1664525u + 1013904223u;
            long q = i / 43, r = q + (long)(s >> 8) % 4000 - 2000;
            h[i] = (int)(r < 0 ? 0 : r >= n_table ? n_table - 1 : r);
        }
        hipMalloc(&rows, sizeof(int) * n_list);
        hipMemcpy(rows, h.data(), sizeof(int) * n_list, hipMemcpyHostToDevice);
        const int per_wave = 43;
        const long waves = (n_list + per_wave - 1) / per_wave;
        const dim3 grid((unsigned)((waves + 3) / 4));
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, grid, dim3(256), 0, 0, table, rows, (int)n_list, width, per_wave);
                if (mode == 1) hipLaunchKernelGGL(k<1>, grid, dim3(256), 0, 0, table, rows, (int)n_list, width, per_wave);
                if (mode == 2) hipLaunchKernelGGL(k<2>, grid, dim3(256), 0, 0, table, rows, (int)n_list, width, per_wave);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
            }
            const double ops = (double)n_list * width;
            printf("width %2d %-7s %8.1f us  %7.1f G lane-ops/s  %6.2f TB/s\n", width, mode == 0 ? "atomic" : mode == 1 ? "store" : "load",
                   best * 1e3, ops / best * 1e-6, ops * 4 / best * 1e-9);
        }
        hipFree(table); hipFree(rows);
    }
    return 0;
}
