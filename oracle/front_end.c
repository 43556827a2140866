/* oracle/front_end.c -- TEST INFRASTRUCTURE ONLY (CPU oracle), not product code.
 *
 * Plain-C restatement of the reference's native front end.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load the library built from this file; the product path
 * (pcrcg_amd/, libpcrcg_hip.so) never does.
 *
 * Parity status: PINNED.  The reference holds no tests / golden vectors for this path
 * (SURVEY.md section 4), so the oracle is pinned against outputs of the reference itself run in
 * the build container: oracle/_ref/libpcrcg_ref.so (unmodified reference C++ behind
 * oracle/ref_shim.cpp, recipe in oracle/Makefile) and the committed fixtures under tests/golden/
 * generated from it by scripts/make_golden_frontend.py.  See tests/test_oracle_frontend.py.
 *
 * Functions and the reference code they restate:
 *
 *   oracle_umap_order            libstdc++ std::unordered_map<size_t,...> iteration order (identity
 *                                hash) as used by zip:cpp_subsampling/grid_subsampling/
 *                                grid_subsampling.cpp:48,59-61,85 (`data.emplace`, `for (auto& v : data)`).
 *                                This is standard-library behaviour (g++ 11.4), restated explicitly
 *                                because it decides the row order of every subsampled level.
 *   oracle_grid_subsample_batch  zip:.../grid_subsampling.cpp:5-106 (single cloud) and :109-211
 *                                (batch loop, max_p cap); SampledData::update_points
 *                                zip:.../grid_subsampling.h:74-79; min_point/max_point/floor
 *                                zip:cpp_utils/cloud/cloud.cpp:27-66, cloud.h:140-143.
 *   oracle_radius_neighbors_batch ref:cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-333
 *                                (batch_nanoflann_neighbors) with nanoflann's metric
 *                                zip:cpp_utils/nanoflann/nanoflann.hpp:432-440 (L2_Simple_Adaptor),
 *                                strict `d2 < r2` test :249-253 and ascending-distance sort :208-214.
 *                                The KD-tree is replaced by a uniform cell grid: the result SET is
 *                                defined as {s : d2(q,s) < r*r} in fp32 and the ORDER as ascending
 *                                (d2, index).  Inside groups of exactly equal d2 the reference's
 *                                order is an artefact of KD-tree traversal + introsort (SURVEY.md
 *                                8a-2); this function uses ascending index there.
 *   oracle_radius_neighbors_batch_reforder  the same tables in the REFERENCE's order, tie groups
 *                                included: nanoflann 1.3.0's tree build and traversal and libstdc++'s
 *                                std::sort restated step by step (see the comment above the kd_*
 *                                functions below).  Equal to the reference entry for entry
 *                                (tests/test_oracle_frontend.py::test_reference_tie_order_*).
 *                                Outside the contract: an EMPTY query cloud followed by a non-empty
 *                                one -- the reference advances one cloud per query (`if`, :270), so
 *                                it searches the wrong cloud there; collate never produces that.
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off (no FMA contraction: the reference is built by
 * distutils with -O2 and no -march, so every fp32 product and sum is rounded separately).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------- */
/* libstdc++ prime bucket counts reached by repeated doubling from 1 (probed with g++ 11.4:
 * insert into std::unordered_map<size_t,int> and print bucket_count() at every change). */
static const uint64_t GROW[] = {1ull,        13ull,       29ull,       59ull,      127ull,     257ull,
                                541ull,      1109ull,     2357ull,     5087ull,    10273ull,   20753ull,
                                42043ull,    85229ull,    172933ull,   351061ull,  712697ull,  1447153ull,
                                2938679ull,  5967347ull,  12117689ull, 24607243ull, 49969847ull};
#define NGROW ((int)(sizeof(GROW) / sizeof(GROW[0])))

/* Iteration order of a std::unordered_map<size_t,T> after emplacing the DISTINCT keys k[0..m-1]
 * in that order.  order[j] = insertion rank of the j-th element visited by `for (auto& v : map)`.
 * Returns 0, or -1 if m exceeds the probed growth table / allocation fails.
 *
 * libstdc++ rules restated (hashtable.h: _M_insert_bucket_begin, _M_rehash_aux(unique keys),
 * hashtable_policy.h: _Prime_rehash_policy::_M_need_rehash):
 *   - one singly linked list of all nodes; bucket b remembers the node BEFORE its first node;
 *   - insert into a non-empty bucket: splice right after that before-node;
 *     into an empty bucket: splice at the list head, and the bucket of the old head node now
 *     has the new node as its before-node;
 *   - when size()+1 would exceed the bucket count, rehash first to the next prime >= 2*count:
 *     walk the old list from its head and re-insert every node with the same two rules. */
int oracle_umap_order(const uint64_t* k, int m, int* order) {
    if (m <= 0) return 0;
    int gi = 0;
    uint64_t B = GROW[0];
    int HEAD = m;
    int* next = (int*)malloc(sizeof(int) * (size_t)(m + 1));
    int* before = NULL;
    size_t before_cap = 0;
    if (!next) return -1;
    next[HEAD] = -1;
    before = (int*)malloc(sizeof(int));
    before_cap = 1;
    before[0] = -1;
    for (int i = 0; i < m; ++i) {
        if ((uint64_t)i + 1 > B) { /* _M_need_rehash: n_elt + n_ins > next_resize (== B) */
            if (gi + 1 >= NGROW) { free(next); free(before); return -1; }
            B = GROW[++gi];
            if (B > before_cap) {
                free(before);
                before = (int*)malloc(sizeof(int) * B);
                before_cap = B;
                if (!before) { free(next); return -1; }
            }
            for (uint64_t b = 0; b < B; ++b) before[b] = -1;
            int p = next[HEAD];
            next[HEAD] = -1;
            uint64_t bbegin_bkt = 0;
            while (p != -1) {
                int nx = next[p];
                uint64_t b = k[p] % B;
                if (before[b] == -1) {
                    next[p] = next[HEAD];
                    next[HEAD] = p;
                    before[b] = HEAD;
                    if (next[p] != -1) before[bbegin_bkt] = p;
                    bbegin_bkt = b;
                } else {
                    next[p] = next[before[b]];
                    next[before[b]] = p;
                }
                p = nx;
            }
        }
        uint64_t b = k[i] % B;
        if (before[b] != -1) {
            next[i] = next[before[b]];
            next[before[b]] = i;
        } else {
            next[i] = next[HEAD];
            next[HEAD] = i;
            if (next[i] != -1) before[k[next[i]] % B] = i;
            before[b] = HEAD;
        }
    }
    int j = 0;
    for (int p = next[HEAD]; p != -1; p = next[p]) order[j++] = p;
    free(next);
    free(before);
    return j == m ? 0 : -1;
}

/* ------------------------------------------------------------------------------------------- */
/* small open-addressing map key -> first-occurrence rank, used only to find cells */
typedef struct {
    uint64_t* keys;
    int* vals;
    size_t cap; /* power of two */
} cellmap;

static int cellmap_init(cellmap* m, size_t n) {
    size_t cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    m->cap = cap;
    m->keys = (uint64_t*)malloc(sizeof(uint64_t) * cap);
    m->vals = (int*)malloc(sizeof(int) * cap);
    if (!m->keys || !m->vals) return -1;
    memset(m->vals, 0xff, sizeof(int) * cap); /* -1 = empty */
    return 0;
}
static void cellmap_free(cellmap* m) { free(m->keys); free(m->vals); }
static inline size_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (size_t)x;
}
/* returns slot; *found tells whether key was present */
static inline size_t cellmap_find(const cellmap* m, uint64_t key, int* found) {
    size_t s = mix64(key) & (m->cap - 1);
    while (m->vals[s] != -1) {
        if (m->keys[s] == key) { *found = 1; return s; }
        s = (s + 1) & (m->cap - 1);
    }
    *found = 0;
    return s;
}

/* (size_t)floor(v) as the reference's x86-64 build evaluates it for the values that occur
 * (non-negative; a value that rounding pushed just below zero converts through int64). */
static inline uint64_t to_size_t(float v) { return (uint64_t)(int64_t)v; }

/* One cloud: zip:.../grid_subsampling.cpp:5-106.  out must hold 3*n floats; returns cell count. */
static int grid_subsample_one(const float* p, int n, float dl, float* out) {
    if (n <= 0) return 0;
    /* min_point / max_point: cloud.cpp:27-66 */
    float mnx = p[0], mny = p[1], mnz = p[2], mxx = p[0], mxy = p[1], mxz = p[2];
    for (int i = 0; i < n; ++i) {
        float x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
        if (x < mnx) mnx = x;
        if (y < mny) mny = y;
        if (z < mnz) mnz = z;
        if (x > mxx) mxx = x;
        if (y > mxy) mxy = y;
        if (z > mxz) mxz = z;
    }
    /* originCorner = floor(minCorner * (1/sampleDl)) * sampleDl   (:27; all fp32) */
    float inv = 1 / dl;
    float ox = floorf(mnx * inv) * dl, oy = floorf(mny * inv) * dl, oz = floorf(mnz * inv) * dl;
    /* sampleNX/NY (:30-31) */
    uint64_t nX = to_size_t(floorf((mxx - ox) / dl)) + 1;
    uint64_t nY = to_size_t(floorf((mxy - oy) / dl)) + 1;
    (void)mxz;

    cellmap map;
    if (cellmap_init(&map, (size_t)n) != 0) return -1;
    uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
    float* sum = (float*)calloc((size_t)n * 3, sizeof(float));
    int* cnt = (int*)calloc((size_t)n, sizeof(int));
    int* order = (int*)malloc(sizeof(int) * (size_t)n);
    int m = 0;
    for (int i = 0; i < n; ++i) {
        float x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
        /* :53-56 */
        uint64_t iX = to_size_t(floorf((x - ox) / dl));
        uint64_t iY = to_size_t(floorf((y - oy) / dl));
        uint64_t iZ = to_size_t(floorf((z - oz) / dl));
        uint64_t key = iX + nX * iY + nX * nY * iZ;
        int found;
        size_t s = cellmap_find(&map, key, &found);
        int c;
        if (!found) {
            map.keys[s] = key;
            map.vals[s] = m;
            keys[m] = key;
            c = m++;
        } else
            c = map.vals[s];
        /* SampledData::update_points (grid_subsampling.h:74-79): count += 1; point += p */
        cnt[c] += 1;
        sum[3 * c] += x;
        sum[3 * c + 1] += y;
        sum[3 * c + 2] += z;
    }
    int rc = oracle_umap_order(keys, m, order);
    if (rc == 0) {
        for (int j = 0; j < m; ++j) {
            int c = order[j];
            /* :87  v.second.point * (1.0 / v.second.count): double reciprocal narrowed to float by
             * operator*(PointXYZ, const float) cloud.h:120-123 */
            float a = (float)(1.0 / (double)cnt[c]);
            out[3 * j] = sum[3 * c] * a;
            out[3 * j + 1] = sum[3 * c + 1] * a;
            out[3 * j + 2] = sum[3 * c + 2] * a;
        }
    }
    cellmap_free(&map);
    free(keys); free(sum); free(cnt); free(order);
    return rc == 0 ? m : -1;
}

/* zip:.../grid_subsampling.cpp:109-211.  out_pts must hold 3*n floats, out_len nb ints.
 * Returns the total number of subsampled points, or -1 on failure. */
int oracle_grid_subsample_batch(const float* pts, int n, const int* len, int nb, float dl, int max_p,
                                float* out_pts, int* out_len) {
    if (max_p < 1) max_p = n; /* :134-135 */
    int sum_b = 0, total = 0;
    float* tmp = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    if (!tmp) return -1;
    for (int b = 0; b < nb; ++b) {
        int m = grid_subsample_one(pts + 3 * (size_t)sum_b, len[b], dl, tmp);
        if (m < 0) { free(tmp); return -1; }
        if (m > max_p) m = max_p; /* :185-205 keep the first max_p */
        memcpy(out_pts + 3 * (size_t)total, tmp, sizeof(float) * 3 * (size_t)m);
        out_len[b] = m;
        total += m;
        sum_b += len[b];
    }
    free(tmp);
    return total;
}

/* ------------------------------------------------------------------------------------------- */
typedef struct { float d2; int idx; } hit;
static int hit_cmp(const void* a, const void* b) {
    const hit* x = (const hit*)a; const hit* y = (const hit*)b;
    if (x->d2 < y->d2) return -1;
    if (x->d2 > y->d2) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

/* ref:.../neighbors.cpp:211-333.  Returns a malloc'ed int32 [nq, *cols] matrix (free with
 * oracle_free) padded with ns (:324), *cols = longest list (:296-297).  NULL if nq*cols == 0
 * (the reference wrapper then raises RuntimeError("Error"), wrapper.cpp:201-205) or on failure. */
int* oracle_radius_neighbors_batch(const float* q, int nq, const float* s, int ns, const int* qlen,
                                   const int* slen, int nb, float radius, int* cols) {
    *cols = 0;
    if (nq <= 0) return NULL;
    float r2 = radius * radius; /* :226 */
    int* counts = (int*)calloc((size_t)nq, sizeof(int));
    size_t* starts = (size_t*)malloc(sizeof(size_t) * ((size_t)nq + 1));
    size_t hcap = (size_t)nq * 32 + 1024, hn = 0;
    hit* hits = (hit*)malloc(sizeof(hit) * hcap);
    if (!counts || !starts || !hits) return NULL;
    double cell = (double)radius * (1.0 + 1e-5);
    if (!(cell > 0)) cell = 1.0;
    int qoff = 0, soff = 0, max_count = 0;
    for (int b = 0; b < nb; ++b) {
        int nsb = slen[b], nqb = qlen[b];
        /* uniform grid over this cloud's supports */
        int64_t gx = 1, gy = 1, gz = 1;
        double mn[3] = {0, 0, 0};
        int *cstart = NULL, *cpts = NULL;
        if (nsb > 0) {
            double mx[3];
            for (int d = 0; d < 3; ++d) mn[d] = mx[d] = s[3 * (size_t)soff + d];
            for (int i = 0; i < nsb; ++i)
                for (int d = 0; d < 3; ++d) {
                    double v = s[3 * (size_t)(soff + i) + d];
                    if (v < mn[d]) mn[d] = v;
                    if (v > mx[d]) mx[d] = v;
                }
            gx = (int64_t)floor((mx[0] - mn[0]) / cell) + 1;
            gy = (int64_t)floor((mx[1] - mn[1]) / cell) + 1;
            gz = (int64_t)floor((mx[2] - mn[2]) / cell) + 1;
            /* keep the dense grid bounded: coarsen (cells may only grow, never shrink below r) */
            while ((double)gx * (double)gy * (double)gz > 4.0e7) {
                cell *= 2.0;
                gx = (int64_t)floor((mx[0] - mn[0]) / cell) + 1;
                gy = (int64_t)floor((mx[1] - mn[1]) / cell) + 1;
                gz = (int64_t)floor((mx[2] - mn[2]) / cell) + 1;
            }
            size_t ncell = (size_t)(gx * gy * gz);
            cstart = (int*)calloc(ncell + 1, sizeof(int));
            cpts = (int*)malloc(sizeof(int) * (size_t)nsb);
            int* cof = (int*)malloc(sizeof(int) * (size_t)nsb);
            for (int i = 0; i < nsb; ++i) {
                const float* p = s + 3 * (size_t)(soff + i);
                int64_t cx = (int64_t)floor(((double)p[0] - mn[0]) / cell);
                int64_t cy = (int64_t)floor(((double)p[1] - mn[1]) / cell);
                int64_t cz = (int64_t)floor(((double)p[2] - mn[2]) / cell);
                cof[i] = (int)(cx + gx * (cy + gy * cz));
                cstart[cof[i] + 1]++;
            }
            for (size_t c = 0; c < ncell; ++c) cstart[c + 1] += cstart[c];
            int* fill = (int*)calloc(ncell, sizeof(int));
            for (int i = 0; i < nsb; ++i) cpts[cstart[cof[i]] + fill[cof[i]]++] = i; /* ascending i */
            free(fill); free(cof);
        }
        for (int i = 0; i < nqb; ++i) {
            const float* p0 = q + 3 * (size_t)(qoff + i);
            starts[qoff + i] = hn;
            int c = 0;
            if (nsb > 0) {
                int64_t cx = (int64_t)floor(((double)p0[0] - mn[0]) / cell);
                int64_t cy = (int64_t)floor(((double)p0[1] - mn[1]) / cell);
                int64_t cz = (int64_t)floor(((double)p0[2] - mn[2]) / cell);
                for (int64_t z = cz - 1; z <= cz + 1; ++z) {
                    if (z < 0 || z >= gz) continue;
                    for (int64_t y = cy - 1; y <= cy + 1; ++y) {
                        if (y < 0 || y >= gy) continue;
                        for (int64_t x = cx - 1; x <= cx + 1; ++x) {
                            if (x < 0 || x >= gx) continue;
                            size_t ci = (size_t)(x + gx * (y + gy * z));
                            for (int t = cstart[ci]; t < cstart[ci + 1]; ++t) {
                                int j = cpts[t];
                                const float* ps = s + 3 * (size_t)(soff + j);
                                /* L2_Simple_Adaptor::evalMetric nanoflann.hpp:432-440:
                                 * result starts at 0 and adds diff*diff per dimension */
                                float d0 = p0[0] - ps[0], d1 = p0[1] - ps[1], d2v = p0[2] - ps[2];
                                float d2 = 0.0f;
                                d2 += d0 * d0;
                                d2 += d1 * d1;
                                d2 += d2v * d2v;
                                if (d2 < r2) { /* RadiusResultSet::addPoint :249-253 */
                                    if (hn == hcap) {
                                        hcap *= 2;
                                        hits = (hit*)realloc(hits, sizeof(hit) * hcap);
                                        if (!hits) return NULL;
                                    }
                                    hits[hn].d2 = d2;
                                    hits[hn].idx = j + soff; /* :319-321 */
                                    ++hn; ++c;
                                }
                            }
                        }
                    }
                }
            }
            counts[qoff + i] = c;
            if (c > max_count) max_count = c;
            qsort(hits + starts[qoff + i], (size_t)c, sizeof(hit), hit_cmp); /* sorted=true :266 */
        }
        free(cstart); free(cpts);
        qoff += nqb;
        soff += nsb;
    }
    int* out = NULL;
    if (max_count > 0) {
        out = (int*)malloc(sizeof(int) * (size_t)nq * (size_t)max_count);
        if (out) {
            for (int i = 0; i < nq; ++i) {
                int* row = out + (size_t)i * max_count;
                for (int j = 0; j < max_count; ++j)
                    row[j] = j < counts[i] ? hits[starts[i] + j].idx : ns; /* :319-325 */
            }
            *cols = max_count;
        }
    }
    free(counts); free(starts); free(hits);
    return out;
}

void oracle_free(void* p) { free(p); }

/* ------------------------------------------------------------------------------------------- */
/* Reference ORDER inside groups of exactly equal distance.
 *
 * oracle_radius_neighbors_batch above defines that order as ascending index.  The reference's order is an
 * artefact of two third-party pieces, restated here so that the oracle can reproduce the reference tables
 * entry for entry (used to count how many rows a `[:, :limit]` cut makes ambiguous, tests/ and DESIGN.md):
 *
 *   - nanoflann 1.3.0 (vendored, zip:cpp_utils/nanoflann/nanoflann.hpp): KDTreeSingleIndexAdaptor with leaf
 *     size 10 (ref:.../neighbors.cpp:245): buildIndex :1190-1203 (vind = iota, root bbox :1318-1338),
 *     divideTree :857-905, middleSplit_ :909-957, planeSplit :967-1003, computeMinMax :836-848;
 *     findNeighbors :1221-1243, computeInitialDistances :1005-1022, searchLevel :1348-1410 (hits are appended
 *     in traversal order, RadiusResultSet::addPoint :246-250);
 *   - libstdc++ std::sort (GCC 11 bits/stl_algo.h: __sort, __introsort_loop, __unguarded_partition_pivot,
 *     __move_median_to_first, __unguarded_partition, __final_insertion_sort, heap fallback __partial_sort) with
 *     IndexDist_Sorter (:208-214: compares the distance only) -- unstable, so the order of equal distances
 *     depends on the traversal order above.
 * All arithmetic is fp32 exactly as written there (this file is built with -ffp-contract=off). */
typedef struct { float low, high; } kd_iv;
typedef struct kd_node {
    int leaf, left, right;       /* leaf: vind range [left, right) */
    int divfeat;
    float divlow, divhigh;
    int child1, child2;          /* node indices */
} kd_node;
typedef struct {
    const float* pts;            /* this cloud's supports [n,3] */
    int n;
    int* vind;
    kd_node* nodes;
    int nnodes, cap;
    kd_iv root[3];
} kd_tree;

static float kd_get(const kd_tree* t, int idx, int d) { return t->pts[3 * (size_t)idx + d]; }

static void kd_minmax(const kd_tree* t, const int* ind, int count, int d, float* mn, float* mx) {
    *mn = *mx = kd_get(t, ind[0], d);
    for (int i = 1; i < count; ++i) {
        float v = kd_get(t, ind[i], d);
        if (v < *mn) *mn = v;
        if (v > *mx) *mx = v;
    }
}

static void kd_plane_split(const kd_tree* t, int* ind, int count, int cutfeat, float cutval, int* lim1, int* lim2) {
    /* IndexType is size_t in the reference: `right` never goes below 0 thanks to the `right &&` tests */
    size_t left = 0, right = (size_t)count - 1;
    for (;;) {
        while (left <= right && kd_get(t, ind[left], cutfeat) < cutval) ++left;
        while (right && left <= right && kd_get(t, ind[right], cutfeat) >= cutval) --right;
        if (left > right || !right) break;
        int tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
        ++left; --right;
    }
    *lim1 = (int)left;
    right = (size_t)count - 1;
    for (;;) {
        while (left <= right && kd_get(t, ind[left], cutfeat) <= cutval) ++left;
        while (right && left <= right && kd_get(t, ind[right], cutfeat) > cutval) --right;
        if (left > right || !right) break;
        int tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
        ++left; --right;
    }
    *lim2 = (int)left;
}

static int kd_divide(kd_tree* t, int left, int right, kd_iv* bbox) {
    if (t->nnodes == t->cap) {
        t->cap = t->cap ? 2 * t->cap : 1024;
        t->nodes = (kd_node*)realloc(t->nodes, sizeof(kd_node) * (size_t)t->cap);
    }
    const int me = t->nnodes++;
    if (right - left <= 10) {                                  /* leaf_max_size (neighbors.cpp:245) */
        kd_node nd = {1, left, right, 0, 0.f, 0.f, -1, -1};
        for (int d = 0; d < 3; ++d) bbox[d].low = bbox[d].high = kd_get(t, t->vind[left], d);
        for (int k = left + 1; k < right; ++k)
            for (int d = 0; d < 3; ++d) {
                float v = kd_get(t, t->vind[k], d);
                if (bbox[d].low > v) bbox[d].low = v;
                if (bbox[d].high < v) bbox[d].high = v;
            }
        t->nodes[me] = nd;
        return me;
    }
    int* ind = t->vind + left;
    const int count = right - left;
    /* middleSplit_ */
    const float EPS = 0.00001f;
    float max_span = bbox[0].high - bbox[0].low;
    for (int d = 1; d < 3; ++d) {
        float span = bbox[d].high - bbox[d].low;
        if (span > max_span) max_span = span;
    }
    float max_spread = -1.f;
    int cutfeat = 0;
    for (int d = 0; d < 3; ++d) {
        float span = bbox[d].high - bbox[d].low;
        if (span > (1 - EPS) * max_span) {
            float mn, mx;
            kd_minmax(t, ind, count, d, &mn, &mx);
            float spread = mx - mn;
            if (spread > max_spread) { cutfeat = d; max_spread = spread; }
        }
    }
    float split_val = (bbox[cutfeat].low + bbox[cutfeat].high) / 2;
    float mn, mx, cutval;
    kd_minmax(t, ind, count, cutfeat, &mn, &mx);
    if (split_val < mn) cutval = mn;
    else if (split_val > mx) cutval = mx;
    else cutval = split_val;
    int lim1, lim2, idx;
    kd_plane_split(t, ind, count, cutfeat, cutval, &lim1, &lim2);
    if (lim1 > count / 2) idx = lim1;
    else if (lim2 < count / 2) idx = lim2;
    else idx = count / 2;

    kd_iv lb[3], rb[3];
    memcpy(lb, bbox, sizeof(lb));
    memcpy(rb, bbox, sizeof(rb));
    lb[cutfeat].high = cutval;
    const int c1 = kd_divide(t, left, left + idx, lb);
    rb[cutfeat].low = cutval;
    const int c2 = kd_divide(t, left + idx, right, rb);
    kd_node nd = {0, 0, 0, cutfeat, lb[cutfeat].high, rb[cutfeat].low, c1, c2};
    t->nodes[me] = nd;
    for (int d = 0; d < 3; ++d) {
        bbox[d].low = lb[d].low < rb[d].low ? lb[d].low : rb[d].low;
        bbox[d].high = lb[d].high > rb[d].high ? lb[d].high : rb[d].high;
    }
    return me;
}

typedef struct { size_t first; float second; } kd_pair;
typedef struct { kd_pair* v; size_t n, cap; } kd_vec;

static void kd_push(kd_vec* r, size_t idx, float d) {
    if (r->n == r->cap) {
        r->cap = r->cap ? 2 * r->cap : 64;
        r->v = (kd_pair*)realloc(r->v, sizeof(kd_pair) * r->cap);
    }
    r->v[r->n].first = idx;
    r->v[r->n].second = d;
    r->n++;
}

static void kd_search(const kd_tree* t, int node, const float* vec, float mindistsq, float* dists, float radius,
                      kd_vec* res) {
    const kd_node* nd = &t->nodes[node];
    if (nd->leaf) {
        for (int i = nd->left; i < nd->right; ++i) {
            const int index = t->vind[i];
            float result = 0.0f;                                   /* L2_Simple_Adaptor::evalMetric */
            for (int d = 0; d < 3; ++d) {
                const float diff = vec[d] - kd_get(t, index, d);
                result += diff * diff;
            }
            if (result < radius) {                                  /* worstDist() == radius; addPoint tests again */
                if (result < radius) kd_push(res, (size_t)index, result);
            }
        }
        return;
    }
    const int idx = nd->divfeat;
    const float val = vec[idx];
    const float diff1 = val - nd->divlow, diff2 = val - nd->divhigh;
    int best, other;
    float cut_dist;
    if ((diff1 + diff2) < 0) { best = nd->child1; other = nd->child2; cut_dist = (val - nd->divhigh) * (val - nd->divhigh); }
    else { best = nd->child2; other = nd->child1; cut_dist = (val - nd->divlow) * (val - nd->divlow); }
    kd_search(t, best, vec, mindistsq, dists, radius, res);
    const float dst = dists[idx];
    mindistsq = mindistsq + cut_dist - dst;
    dists[idx] = cut_dist;
    if (mindistsq * 1.0f <= radius) kd_search(t, other, vec, mindistsq, dists, radius, res);   /* epsError = 1 + 0 */
    dists[idx] = dst;
}

/* ---- libstdc++ std::sort with comp(a, b) = a.second < b.second ---- */
#define KD_LT(a, b) ((a).second < (b).second)
static void kd_swap(kd_pair* a, kd_pair* b) { kd_pair t = *a; *a = *b; *b = t; }

static void kd_adjust_heap(kd_pair* first, long hole, long len, kd_pair value) {
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (KD_LT(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    long parent = (hole - 1) / 2;                                   /* __push_heap */
    while (hole > top && KD_LT(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

static void kd_heapsort(kd_pair* first, kd_pair* last) {           /* __partial_sort(first, last, last) */
    const long len = last - first;
    if (len >= 2) {                                                 /* __make_heap */
        long parent = (len - 2) / 2;
        for (;;) {
            kd_pair v = first[parent];
            kd_adjust_heap(first, parent, len, v);
            if (parent == 0) break;
            parent--;
        }
    }
    /* __heap_select has no elements beyond `middle == last`; __sort_heap: */
    while (last - first > 1) {
        --last;
        kd_pair v = *last;                                          /* __pop_heap(first, last, last) */
        *last = *first;
        kd_adjust_heap(first, 0, last - first, v);
    }
}

static void kd_unguarded_linear_insert(kd_pair* last) {
    kd_pair val = *last;
    kd_pair* next = last - 1;
    while (KD_LT(val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}

static void kd_insertion_sort(kd_pair* first, kd_pair* last) {
    if (first == last) return;
    for (kd_pair* i = first + 1; i != last; ++i) {
        if (KD_LT(*i, *first)) {
            kd_pair val = *i;
            memmove(first + 1, first, sizeof(kd_pair) * (size_t)(i - first));
            *first = val;
        } else {
            kd_unguarded_linear_insert(i);
        }
    }
}

static void kd_introsort_loop(kd_pair* first, kd_pair* last, long depth_limit) {
    while (last - first > 16) {
        if (depth_limit == 0) { kd_heapsort(first, last); return; }
        --depth_limit;
        kd_pair* mid = first + (last - first) / 2;                  /* __unguarded_partition_pivot */
        kd_pair *a = first + 1, *b = mid, *c = last - 1;            /* __move_median_to_first(first, a, b, c) */
        if (KD_LT(*a, *b)) {
            if (KD_LT(*b, *c)) kd_swap(first, b);
            else if (KD_LT(*a, *c)) kd_swap(first, c);
            else kd_swap(first, a);
        } else if (KD_LT(*a, *c)) kd_swap(first, a);
        else if (KD_LT(*b, *c)) kd_swap(first, c);
        else kd_swap(first, b);
        kd_pair *lo = first + 1, *hi = last;                         /* __unguarded_partition(first+1, last, first) */
        for (;;) {
            while (KD_LT(*lo, *first)) ++lo;
            --hi;
            while (KD_LT(*first, *hi)) --hi;
            if (!(lo < hi)) break;
            kd_swap(lo, hi);
            ++lo;
        }
        kd_introsort_loop(lo, last, depth_limit);
        last = lo;
    }
}

static void kd_std_sort(kd_pair* first, kd_pair* last) {
    if (first == last) return;
    long n = last - first, lg = 0;
    while ((n >> (lg + 1)) > 0) ++lg;                               /* std::__lg */
    kd_introsort_loop(first, last, 2 * lg);
    if (last - first > 16) {                                         /* __final_insertion_sort */
        kd_insertion_sort(first, first + 16);
        for (kd_pair* i = first + 16; i != last; ++i) kd_unguarded_linear_insert(i);
    } else {
        kd_insertion_sort(first, last);
    }
}

/* Same contract as oracle_radius_neighbors_batch, rows in the REFERENCE's order (ties included). */
int* oracle_radius_neighbors_batch_reforder(const float* q, int nq, const float* s, int ns, const int* qlen,
                                            const int* slen, int nb, float radius, int* cols) {
    *cols = 0;
    if (nq <= 0) return NULL;
    const float r2 = radius * radius;
    kd_vec* rows = (kd_vec*)calloc((size_t)nq, sizeof(kd_vec));
    int qoff = 0, soff = 0;
    size_t max_count = 0;
    for (int b = 0; b < nb; ++b) {
        kd_tree t;
        memset(&t, 0, sizeof(t));
        t.pts = s + 3 * (size_t)soff;
        t.n = slen[b];
        if (t.n > 0) {
            t.vind = (int*)malloc(sizeof(int) * (size_t)t.n);
            for (int i = 0; i < t.n; ++i) t.vind[i] = i;
            for (int d = 0; d < 3; ++d) t.root[d].low = t.root[d].high = kd_get(&t, 0, d);
            for (int k = 1; k < t.n; ++k)
                for (int d = 0; d < 3; ++d) {
                    float v = kd_get(&t, k, d);
                    if (v < t.root[d].low) t.root[d].low = v;
                    if (v > t.root[d].high) t.root[d].high = v;
                }
            kd_divide(&t, 0, t.n, t.root);                           /* updates root bbox in place, as the reference */
        }
        for (int i = 0; i < qlen[b]; ++i) {
            kd_vec* r = &rows[qoff + i];
            if (t.n > 0) {
                const float* vec = q + 3 * (size_t)(qoff + i);
                float dists[3] = {0.f, 0.f, 0.f}, distsq = 0.f;     /* computeInitialDistances */
                for (int d = 0; d < 3; ++d) {
                    if (vec[d] < t.root[d].low) { dists[d] = (vec[d] - t.root[d].low) * (vec[d] - t.root[d].low); distsq += dists[d]; }
                    if (vec[d] > t.root[d].high) { dists[d] = (vec[d] - t.root[d].high) * (vec[d] - t.root[d].high); distsq += dists[d]; }
                }
                kd_search(&t, 0, vec, distsq, dists, r2, r);
                kd_std_sort(r->v, r->v + r->n);
                for (size_t j = 0; j < r->n; ++j) r->v[j].first += (size_t)soff;
            }
            if (r->n > max_count) max_count = r->n;
        }
        free(t.vind);
        free(t.nodes);
        qoff += qlen[b];
        soff += slen[b];
    }
    int* out = NULL;
    if (max_count > 0) {
        out = (int*)malloc(sizeof(int) * (size_t)nq * max_count);
        for (int i = 0; i < nq; ++i)
            for (size_t j = 0; j < max_count; ++j)
                out[(size_t)i * max_count + j] = j < rows[i].n ? (int)rows[i].v[j].first : ns;
    }
    for (int i = 0; i < nq; ++i) free(rows[i].v);
    free(rows);
    *cols = (int)max_count;
    return out;
}
