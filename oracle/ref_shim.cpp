// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE, not product code.
//
// A thin extern "C" shim (written for this repo) over the two entry points of the
// reference's native front end, so that the UNMODIFIED reference sources can be
// called through ctypes as a checker for oracle/front_end.c and for the HIP path:
//
//   batch_nanoflann_neighbors  ref:cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-333
//   batch_grid_subsampling     zip:cpp_subsampling/grid_subsampling/grid_subsampling.cpp:109-211
//
// It replaces the reference's CPython/NumPy glue (ref:cpp_wrappers/cpp_neighbors/wrapper.cpp:58-238,
// zip:cpp_subsampling/wrapper.cpp:62-330), which no longer compiles against NumPy 2.x.
// The reference sources are never copied into this repository: oracle/Makefile unzips
// /root/reference/cpp_wrappers.zip into a temporary directory, compiles them together with this
// file into oracle/_ref/libpcrcg_ref.so (git-ignored) and deletes the temporary directory.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cpp_neighbors/neighbors/neighbors.h"
#include "cpp_subsampling/grid_subsampling/grid_subsampling.h"

extern "C" {

// Returns a malloc'ed int32 [nq, *cols] matrix (caller frees with ref_free), or NULL when the
// reference would raise RuntimeError("Error") (ref wrapper.cpp:201-205).
int* ref_batch_query(const float* q, int nq, const float* s, int ns, const int* qb, const int* sb,
                     int nb, float radius, int* cols) {
    std::vector<PointXYZ> queries((const PointXYZ*)q, (const PointXYZ*)q + nq);
    std::vector<PointXYZ> supports((const PointXYZ*)s, (const PointXYZ*)s + ns);
    std::vector<int> q_batches(qb, qb + nb), s_batches(sb, sb + nb);
    std::vector<int> out;
    batch_nanoflann_neighbors(queries, supports, q_batches, s_batches, out, radius);
    if (out.size() < 1 || nq < 1) { *cols = 0; return NULL; }
    *cols = (int)(out.size() / (size_t)nq);
    int* r = (int*)malloc(out.size() * sizeof(int));
    memcpy(r, out.data(), out.size() * sizeof(int));
    return r;
}

// Returns a malloc'ed float32 [*m, 3] matrix; out_b receives nb per-cloud counts.
float* ref_subsample_batch(const float* p, int n, const int* b, int nb, float dl, int max_p, int* m,
                           int* out_b) {
    std::vector<PointXYZ> pts((const PointXYZ*)p, (const PointXYZ*)p + n);
    std::vector<PointXYZ> sub;
    std::vector<float> f, sf;
    std::vector<int> c, sc;
    std::vector<int> ob(b, b + nb), sb;
    batch_grid_subsampling(pts, sub, f, sf, c, sc, ob, sb, dl, max_p);
    *m = (int)sub.size();
    for (int i = 0; i < nb && i < (int)sb.size(); ++i) out_b[i] = sb[i];
    float* r = (float*)malloc(sub.size() * 3 * sizeof(float) + 4);
    memcpy(r, sub.data(), sub.size() * 3 * sizeof(float));
    return r;
}

// Iteration order of this toolchain's std::unordered_map<size_t,int> after emplacing n distinct
// keys in order (standard-library behaviour, used to pin oracle_umap_order in oracle/front_end.c).
void ref_umap_order(const unsigned long long* keys, int n, int* order) {
    std::unordered_map<size_t, int> m;
    for (int i = 0; i < n; ++i)
        if (m.count((size_t)keys[i]) < 1) m.emplace((size_t)keys[i], i);
    int j = 0;
    for (auto& v : m) order[j++] = v.second;
}

void ref_free(void* p) { free(p); }
}
