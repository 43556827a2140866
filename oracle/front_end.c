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
 *                                8a-2); this oracle and the HIP path both use ascending index there.
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
