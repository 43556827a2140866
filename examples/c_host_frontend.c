/* c_host_frontend.c -- a plain C host driving libpcrcg_hip.so through include/pcrcg.h (no Python, no torch):
 * two synthetic clouds -> grid subsampling -> radius neighbours of the subsampled level; then the same clouds snapped
 * to a 1/32 lattice (exactly equal distances everywhere) -> radius neighbours in the REFERENCE's order inside tie
 * groups (pcrcg_radius_query_ex + pcrcg_kdforest_build + pcrcg_radius_reorder).  Prints the numbers
 * tests/test_c_host_gpu.py compares with the Python binding on the same input.
 *
 * Build:  gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host_frontend.c -o c_host_frontend \
 *             -Lpcrcg_amd -l:libpcrcg_hip.so -L/opt/rocm/lib -lamdhip64
 * Run:              LD_LIBRARY_PATH=pcrcg_amd ./c_host_frontend */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "pcrcg.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_PCRCG(x) do { int r_ = (x); if (r_ != PCRCG_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, pcrcg_last_error()); return 3; } } while (0)

int main(void) {
    const int n_per = 4000, nb = 2, n = n_per * nb;
    const float dl = 0.05f, radius = 0.125f;
    const int cols = 40;
    float* h_pts = (float*)malloc(sizeof(float) * 3 * n);
    unsigned int s = 12345u;                                  /* LCG shared with the Python test */
    for (int i = 0; i < 3 * n; ++i) {
        s = s * 1664525u + 1013904223u;
        h_pts[i] = (float)(s >> 8) * (1.0f / 16777216.0f);    /* uniform in [0,1) */
    }
    int h_len[2] = {n_per, n_per};

    float *d_pts, *d_sub;
    int *d_len, *d_sub_len, *d_m, *d_max, *d_status, *d_count;
    CHECK_HIP(hipMalloc((void**)&d_pts, sizeof(float) * 3 * n));
    CHECK_HIP(hipMalloc((void**)&d_sub, sizeof(float) * 3 * n));
    CHECK_HIP(hipMalloc((void**)&d_len, sizeof(int) * nb));
    CHECK_HIP(hipMalloc((void**)&d_sub_len, sizeof(int) * nb));
    CHECK_HIP(hipMalloc((void**)&d_m, sizeof(int)));
    CHECK_HIP(hipMalloc((void**)&d_max, sizeof(int)));
    CHECK_HIP(hipMalloc((void**)&d_status, sizeof(int)));
    CHECK_HIP(hipMemcpy(d_pts, h_pts, sizeof(float) * 3 * n, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_len, h_len, sizeof(int) * nb, hipMemcpyHostToDevice));

    /* 1. grid subsampling (replaces cpp_subsampling.subsample_batch) */
    size_t ws_bytes = pcrcg_grid_subsample_ws_bytes(n, nb);
    void* ws;
    CHECK_HIP(hipMalloc(&ws, ws_bytes));
    CHECK_PCRCG(pcrcg_grid_subsample_batch(d_pts, n, d_len, nb, dl, 0, d_sub, d_sub_len, d_m, ws, ws_bytes, NULL));
    int m = 0, sub_len[2];
    CHECK_HIP(hipMemcpy(&m, d_m, sizeof(int), hipMemcpyDeviceToHost));     /* synchronises the null stream */
    CHECK_HIP(hipMemcpy(sub_len, d_sub_len, sizeof(int) * nb, hipMemcpyDeviceToHost));

    /* 2. radius neighbours of the subsampled level (replaces cpp_neighbors.batch_query + [:, :cols].long()) */
    size_t nws_bytes = pcrcg_radius_neighbors_ws_bytes(m, nb);
    void* nws;
    int64_t* d_idx;
    CHECK_HIP(hipMalloc(&nws, nws_bytes));
    CHECK_HIP(hipMalloc((void**)&d_idx, sizeof(int64_t) * (size_t)m * cols));
    CHECK_HIP(hipMalloc((void**)&d_count, sizeof(int) * m));
    CHECK_PCRCG(pcrcg_radius_neighbors_batch(d_sub, m, d_sub, m, d_sub_len, d_sub_len, nb, radius, cols, d_idx, d_count,
                                             d_max, d_status, nws, nws_bytes, NULL));
    int max_count = 0, status = 0;
    CHECK_HIP(hipMemcpy(&max_count, d_max, sizeof(int), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&status, d_status, sizeof(int), hipMemcpyDeviceToHost));
    int64_t* h_idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)m * cols);
    float* h_sub = (float*)malloc(sizeof(float) * 3 * m);
    CHECK_HIP(hipMemcpy(h_idx, d_idx, sizeof(int64_t) * (size_t)m * cols, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_sub, d_sub, sizeof(float) * 3 * m, hipMemcpyDeviceToHost));
    unsigned long long idx_sum = 0, shadow = 0;
    for (size_t i = 0; i < (size_t)m * cols; ++i) {
        idx_sum += (unsigned long long)h_idx[i];
        shadow += h_idx[i] == m;
    }
    double coord_sum = 0.0;
    for (int i = 0; i < 3 * m; ++i) coord_sum += (double)h_sub[i];
    printf("abi=%d m=%d len0=%d len1=%d max_count=%d status=%d idx_sum=%llu shadow=%llu coord_sum=%.6f\n",
           pcrcg_abi_version(), m, sub_len[0], sub_len[1], max_count, status, idx_sum, shadow, coord_sum);

    /* 3. lattice-snapped clouds: the table in the reference's own order inside groups of equal distance */
    {
        const int tn = 1500 * nb, tcols = 48;
        const float tr = 0.11f;
        int t_len[2] = {1500, 1500};
        float* h_t = (float*)malloc(sizeof(float) * 3 * tn);
        for (int i = 0; i < 3 * tn; ++i) h_t[i] = (float)(int)(h_pts[i] * 32.0f) * (1.0f / 32.0f);
        float* d_t;
        int *d_tlen, *d_tcount, *d_ties, *d_meta;     /* meta: max_count, status, tie_count, reorder status */
        int64_t* d_tidx;
        CHECK_HIP(hipMalloc((void**)&d_t, sizeof(float) * 3 * tn));
        CHECK_HIP(hipMalloc((void**)&d_tlen, sizeof(int) * nb));
        CHECK_HIP(hipMalloc((void**)&d_tcount, sizeof(int) * tn));
        CHECK_HIP(hipMalloc((void**)&d_ties, sizeof(int) * tn));
        CHECK_HIP(hipMalloc((void**)&d_meta, sizeof(int) * 4));
        CHECK_HIP(hipMalloc((void**)&d_tidx, sizeof(int64_t) * (size_t)tn * tcols));
        CHECK_HIP(hipMemcpy(d_t, h_t, sizeof(float) * 3 * tn, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_tlen, t_len, sizeof(int) * nb, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemset(d_meta, 0, sizeof(int) * 4));
        size_t gbytes = pcrcg_cellgrid_ws_bytes(tn, nb), fbytes = pcrcg_kdforest_ws_bytes(tn, nb);
        void *grid, *forest;
        CHECK_HIP(hipMalloc(&grid, gbytes));
        CHECK_HIP(hipMalloc(&forest, fbytes));
        CHECK_PCRCG(pcrcg_cellgrid_build(d_t, tn, d_tlen, nb, tr, grid, gbytes, NULL));
        CHECK_PCRCG(pcrcg_radius_query_ex(d_t, tn, d_tlen, tn, d_tlen, nb, tr, grid, tcols, d_tidx, d_tcount, d_meta,
                                          d_meta + 1, d_ties, d_meta + 2, NULL));
        int meta[4];
        CHECK_HIP(hipMemcpy(meta, d_meta, sizeof(int) * 4, hipMemcpyDeviceToHost));
        CHECK_PCRCG(pcrcg_kdforest_build(d_t, tn, d_tlen, nb, forest, fbytes, NULL));
        CHECK_PCRCG(pcrcg_radius_reorder(d_t, tn, d_tlen, nb, d_t, tn, nb, forest, 0, tr, d_ties, meta[2], d_tcount,
                                         meta[0], tcols, d_tidx, d_meta + 3, NULL));
        int64_t* h_tidx = (int64_t*)malloc(sizeof(int64_t) * (size_t)tn * tcols);
        CHECK_HIP(hipMemcpy(h_tidx, d_tidx, sizeof(int64_t) * (size_t)tn * tcols, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(meta, d_meta, sizeof(int) * 4, hipMemcpyDeviceToHost));
        unsigned long long order_sum = 0;          /* order-sensitive checksum of the kept columns */
        const int keep = meta[0] < tcols ? meta[0] : tcols;
        for (int i = 0; i < tn; ++i)
            for (int j = 0; j < keep; ++j) order_sum += (unsigned long long)(j + 1) * (unsigned long long)h_tidx[(size_t)i * tcols + j];
        printf("tie_max_count=%d tie_status=%d tie_rows=%d reorder_status=%d order_sum=%llu\n", meta[0], meta[1], meta[2],
               meta[3], order_sum);
    }

    /* argument validation happens before any launch */
    if (pcrcg_grid_subsample_batch(NULL, n, d_len, nb, dl, 0, d_sub, d_sub_len, d_m, ws, ws_bytes, NULL) != PCRCG_EBADARG) return 4;
    if (pcrcg_grid_subsample_batch(d_pts, n, d_len, nb, dl, 0, d_sub, d_sub_len, d_m, ws, 16, NULL) != PCRCG_EWORKSPACE) return 5;
    return 0;
}
