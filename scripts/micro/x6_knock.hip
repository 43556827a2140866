// x6_knock.hip -- measurement aid (GPU box): k_gemm_x6 with one of its parts removed (template parameter KNOCK of
// pcrcg_amd/csrc/gemm_x6.hip; results wrong, timing meaningful) on the path's big shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I pcrcg_amd/csrc scripts/micro/x6_knock.hip -o scripts/micro/x6_knock
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../pcrcg_amd/csrc/gemm_x6.hip"

namespace pcrcg {
void set_error(const char*, ...) {}
const DebugOpts& debug_opts() { static DebugOpts d; return d; }
KpProfScope::KpProfScope(hipStream_t s, int, int, int, int, int) : st(s), a(nullptr), b(nullptr), on(false) {}
KpProfScope::~KpProfScope() {}
}  // namespace pcrcg
using namespace pcrcg;

template <int BM, int BN, int MINB, int KNOCK>
float run(const float* a, const float* b, float* c, int m, int n, int k, int splits) {
    const int ktiles = (k + 31) / 32;
    const int kps = ((ktiles + splits - 1) / splits) * 32;
    dim3 grid((n + BN - 1) / BN, (m + BM - 1) / BM, (k + kps - 1) / kps);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_x6<BM, BN, MINB, 3, 0, 0, 0, KNOCK>(grid, 0, a, k, b, k, c, n, m, n, k, nullptr, nullptr, kps, 1, 1, splits > 1, nullptr, 0);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch_x6<BM, BN, MINB, 3, 0, 0, 0, KNOCK>(grid, 0, a, k, b, k, c, n, m, n, k, nullptr, nullptr, kps, 1, 1, splits > 1, nullptr, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

template <int BM, int BN, int MINB>
void sweep(const char* tag, const float* a, const float* b, float* c, int m, int n, int k, int splits) {
    printf("%-8s %6dx%4dx%5d split %d | full %6.1f | -loads %6.1f | -split %6.1f | -ldsW %6.1f | -mfma %6.1f | -ldsR %6.1f | -bar %6.1f | "
           "-loads-split-ldsW %6.1f | only loads(-split-ldsW-mfma-ldsR) %6.1f | -ldsR-mfma %6.1f | loads+split(-ldsW-mfma-ldsR) %6.1f | only loads, no barriers %6.1f\n", tag, m, n, k, splits,
           run<BM, BN, MINB, 0>(a, b, c, m, n, k, splits), run<BM, BN, MINB, 1>(a, b, c, m, n, k, splits),
           run<BM, BN, MINB, 2>(a, b, c, m, n, k, splits), run<BM, BN, MINB, 4>(a, b, c, m, n, k, splits),
           run<BM, BN, MINB, 8>(a, b, c, m, n, k, splits), run<BM, BN, MINB, 16>(a, b, c, m, n, k, splits),
           run<BM, BN, MINB, 32>(a, b, c, m, n, k, splits), run<BM, BN, MINB, 7>(a, b, c, m, n, k, splits),
           run<BM, BN, MINB, 30>(a, b, c, m, n, k, splits), run<BM, BN, MINB, 24>(a, b, c, m, n, k, splits),
           run<BM, BN, MINB, 28>(a, b, c, m, n, k, splits), run<BM, BN, MINB, 62>(a, b, c, m, n, k, splits));
    fflush(stdout);
}

int main() {
    const int shapes[][4] = {{15456, 128, 1920, 2}, {60000, 64, 960, 1}, {3934, 256, 3840, 4}, {60000, 256, 128, 1}, {15456, 128, 512, 1}};
    for (auto& s : shapes) {
        const int m = s[0], n = s[1], k = s[2];
        float *a, *b, *c;
        hipMalloc(&a, sizeof(float) * (size_t)m * k);
        hipMalloc(&b, sizeof(float) * (size_t)n * k);
        hipMalloc(&c, sizeof(float) * (size_t)m * n);
        std::vector<float> h((size_t)m * k);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
        hipMemcpy(a, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice);
        hipMemcpy(b, h.data(), sizeof(float) * (size_t)n * k, hipMemcpyHostToDevice);
        hipMemset(c, 0, sizeof(float) * (size_t)m * n);
        sweep<64, 64, 4>("64x64", a, b, c, m, n, k, s[3]);
        sweep<128, 64, 2>("128x64", a, b, c, m, n, k, s[3]);
        if (n >= 128) sweep<128, 128, 2>("128x128", a, b, c, m, n, k, s[3]);
        if (n >= 128) sweep<128, 128, 2>("128x128", a, b, c, m, n, k, s[3] * 2);
        hipFree(a);
        hipFree(b);
        hipFree(c);
    }
    return 0;
}
