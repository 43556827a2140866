// stream_rows.hip -- measurement aid (GPU box): how fast can 64-row workgroup tiles stream a row-major fp32 matrix
// [M, K] when every k-step takes RB bytes of each of its 64 rows (the A-operand pattern of k_gemm_x6: RB = 128), with
// D steps in flight?  Prints TB/s per (RB, D).  hipcc --offload-arch=gfx950 -O3 stream_rows.hip -o /tmp/sr && /tmp/sr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

// PAT 1: the GEMM's lane map -- a thread owns 32 contiguous bytes of a row and fetches them with two 16-byte loads, so one
// wave-level load instruction touches every other 16-byte chunk of 16 rows (half of each 128-byte line per instruction)
template <int D>
__global__ void __launch_bounds__(256, 4) k_stream_pairs(const float* __restrict__ A, int M, int K, float* __restrict__ out) {
    const int m0 = blockIdx.x * 64, tid = threadIdx.x;
    const int ksteps = K / 32;
    f4 r[D][2];
    f4 acc = {0, 0, 0, 0};
    const float* base = A + (long)min(m0 + (tid >> 2), M - 1) * K + (tid & 3) * 8;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float* p = base + (long)min(d, ksteps - 1) * 32;
        r[d][0] = *reinterpret_cast<const f4*>(p);
        r[d][1] = *reinterpret_cast<const f4*>(p + 4);
    }
    for (int s = 0; s < ksteps; s += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            acc += r[d][0] + r[d][1];
            const float* p = base + (long)min(s + d + D, ksteps - 1) * 32;
            r[d][0] = *reinterpret_cast<const f4*>(p);
            r[d][1] = *reinterpret_cast<const f4*>(p + 4);
            __syncthreads();
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

// L2-resident variant: 1024+ workgroups re-read the same 512 rows (3.9 MB at K = 1920: every line is an L2 hit after the
// first touch): what the L2 -> L1 fill path delivers per CU with D tile-steps of 64 rows x 128 B in flight per workgroup
template <int D, int WGS_PER_CU>
__global__ void __launch_bounds__(256, WGS_PER_CU) k_stream_l2(const float* __restrict__ A, int K, float* __restrict__ out) {
    const int m0 = (blockIdx.x & 7) * 64, tid = threadIdx.x;
    const int ksteps = K / 32;
    f4 r[D][2];
    f4 acc = {0, 0, 0, 0};
    const float* base = A + (long)(m0 + (tid >> 2)) * K + (tid & 3) * 8;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float* p = base + (long)min(d, ksteps - 1) * 32;
        r[d][0] = *reinterpret_cast<const f4*>(p);
        r[d][1] = *reinterpret_cast<const f4*>(p + 4);
    }
    for (int rep = 0; rep < 4; ++rep)
        for (int s = 0; s < ksteps; s += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                acc += r[d][0] + r[d][1];
                const float* p = base + (long)((s + d + D) % ksteps) * 32;
                r[d][0] = *reinterpret_cast<const f4*>(p);
                r[d][1] = *reinterpret_cast<const f4*>(p + 4);
            }
        }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

template <int D, int W>
void run_l2(const float* A, int K, float* out) {
    const int grid = 256 * W;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k_stream_l2<D, W><<<grid, 256>>>(A, K, out);
    hipEventRecord(a);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) k_stream_l2<D, W><<<grid, 256>>>(A, K, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps, bytes = (double)grid * 4 * K * 64 * 4;
    printf("L2-resident rows, %d workgroups per CU, depth %d: %7.1f us  %6.2f TB/s of L1 fills = %5.1f bytes per clock and CU (2.4 GHz)\n",
           W, D, us, bytes / us / 1e6, bytes / us / 1e6 * 1e12 / 256 / 2.4e9);
}

template <int RB, int D>
__global__ void __launch_bounds__(256, 4) k_stream(const float* __restrict__ A, int M, int K, float* __restrict__ out) {
    constexpr int F4_PER_ROW = RB / 16;               // float4 per row and step
    constexpr int ITEMS = 64 * F4_PER_ROW / 256;      // float4 per thread and step
    static_assert(ITEMS >= 1, "tile too small");
    const int m0 = blockIdx.x * 64, tid = threadIdx.x;
    const int ksteps = K * 4 / RB;
    f4 r[D][ITEMS];
    f4 acc = {0, 0, 0, 0};
    const float* base[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int e = tid + i * 256, row = e / F4_PER_ROW, c = e % F4_PER_ROW;
        base[i] = A + (long)min(m0 + row, M - 1) * K + c * 4;
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) r[d][i] = *reinterpret_cast<const f4*>(base[i] + (long)min(d, ksteps - 1) * (RB / 4));
    for (int s = 0; s < ksteps; s += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) acc += r[d][i];
            const int nx = min(s + d + D, ksteps - 1);
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) r[d][i] = *reinterpret_cast<const f4*>(base[i] + (long)nx * (RB / 4));
            __syncthreads();     // the GEMM has (two) barriers per step
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

template <int RB, int D>
void run(const float* A, int M, int K, float* out) {
    const int grid = (M + 63) / 64;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k_stream<RB, D><<<grid, 256>>>(A, M, K, out);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) k_stream<RB, D><<<grid, 256>>>(A, M, K, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps;
    printf("M %6d K %5d  RB %4d B/row/step  depth %d: %7.1f us  %5.2f TB/s\n", M, K, RB, D, us, (double)M * K * 4 / us / 1e6);
}

template <int D>
void run_pairs(const float* A, int M, int K, float* out) {
    const int grid = (M + 63) / 64;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k_stream_pairs<D><<<grid, 256>>>(A, M, K, out);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) k_stream_pairs<D><<<grid, 256>>>(A, M, K, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps;
    printf("M %6d K %5d  GEMM lane map (2 x 16 B per thread) depth %d: %7.1f us  %5.2f TB/s\n", M, K, D, us, (double)M * K * 4 / us / 1e6);
}

int main() {
    const int shapes[3][2] = {{15456, 1920}, {60000, 960}, {3934, 3840}};
    float *A, *out;
    hipMalloc(&A, (size_t)60000 * 3840 * 4);
    hipMalloc(&out, 4);
    hipMemset(A, 0, (size_t)60000 * 3840 * 4);
    run_l2<2, 1>(A, 1920, out); run_l2<2, 2>(A, 1920, out); run_l2<2, 4>(A, 1920, out); run_l2<4, 4>(A, 1920, out);
    run_l2<2, 8>(A, 1920, out); run_l2<4, 8>(A, 1920, out);
    for (auto& s : shapes) {
        run_pairs<2>(A, s[0], s[1], out);
        run_pairs<3>(A, s[0], s[1], out);
        run<128, 2>(A, s[0], s[1], out);
        run<128, 3>(A, s[0], s[1], out);
        run<128, 4>(A, s[0], s[1], out);
        run<256, 2>(A, s[0], s[1], out);
        run<256, 3>(A, s[0], s[1], out);
        run<512, 2>(A, s[0], s[1], out);
    }
    return 0;
}
