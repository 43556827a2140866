/* asan_driver.c -- sanitizer job of the CPU checkers (test infrastructure; `make -C oracle asan`).
 *
 * Built together with oracle/front_end.c under -fsanitize=address,undefined (and, when /root/reference is present, a
 * second time against the unmodified reference C++ behind oracle/ref_shim.cpp): drives every entry point of the CPU
 * restatement over the cases the GPU parity tests use -- ragged and EMPTY clouds, duplicate points, a lattice full of
 * exactly equal distances, rows of hundreds of hits -- and compares restatement and reference where both are linked.
 * Exit code 0 = no sanitizer report and (with the reference) identical results.  Replaces nothing in the reference:
 * SURVEY.md section 5 lists "sanitizers" among the auxiliary subsystems the reference lacks. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int oracle_umap_order(const uint64_t* k, int m, int* order);
int oracle_grid_subsample_batch(const float* pts, int n, const int* len, int nb, float dl, int max_p, float* out_pts, int* out_len);
int* oracle_radius_neighbors_batch(const float* q, int nq, const float* s, int ns, const int* qlen, const int* slen, int nb,
                                   float radius, int* cols);
int* oracle_radius_neighbors_batch_reforder(const float* q, int nq, const float* s, int ns, const int* qlen, const int* slen,
                                            int nb, float radius, int* cols);
void oracle_free(void* p);
#ifdef WITH_REF
int* ref_batch_query(const float* q, int nq, const float* s, int ns, const int* qb, const int* sb, int nb, float radius, int* cols);
float* ref_subsample_batch(const float* p, int n, const int* b, int nb, float dl, int max_p, int* m, int* out_b);
void ref_umap_order(const unsigned long long* keys, int n, int* order);
void ref_free(void* p);
#endif

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static float frand(void) {
    rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
    return (float)((rng >> 40) & 0xFFFFFF) / 16777216.0f;
}

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { fprintf(stderr, "asan_driver: " __VA_ARGS__); fprintf(stderr, "\n"); ++fails; } } while (0)

static void one_case(int n0, int n1, float side, float lattice, float dl, float radius) {
    const int nb = 2, n = n0 + n1;
    int len[2] = {n0, n1};
    float* p = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < 3 * n; ++i) {
        float v = frand() * side - 0.25f * side;               /* negative coordinates included */
        if (lattice > 0.f) v = (float)(int)(v * lattice) / lattice;   /* snapped: duplicates and equal distances */
        p[i] = v;
    }
    /* grid subsampling */
    float* sub = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    int sl[2] = {0, 0};
    const int m = oracle_grid_subsample_batch(p, n, len, nb, dl, 0, sub, sl);
    CHECK(m >= 0 && m <= n && sl[0] + sl[1] == m, "subsample: m = %d of %d (%d + %d)", m, n, sl[0], sl[1]);
#ifdef WITH_REF
    /* (the reference itself dies on an empty cloud -- an integer division by zero in its progress arithmetic,
     * grid_subsampling.cpp:21 -- so the empty-cloud case runs through the restatement only) */
    const int ref_ok = n0 > 0 && n1 > 0;
    if (ref_ok) {
        int rm = 0, rb[2] = {0, 0};
        float* rs = ref_subsample_batch(p, n, len, nb, dl, 0, &rm, rb);
        CHECK(rm == m && rb[0] == sl[0] && rb[1] == sl[1], "subsample count differs from the reference: %d vs %d", m, rm);
        if (rm == m) CHECK(memcmp(rs, sub, sizeof(float) * 3 * (size_t)m) == 0, "subsampled points differ from the reference");
        ref_free(rs);
    }
#endif
    /* radius search: self, pool (coarse queries over fine supports), up (fine queries over coarse supports) */
    const float* Q[3] = {p, sub, p};
    const float* S[3] = {p, p, sub};
    const int NQ[3] = {n, m, n}, NS[3] = {n, n, m};
    const int* QL[3] = {len, sl, len};
    const int* SL[3] = {len, len, sl};
    for (int t = 0; t < 3; ++t) {
        int c1 = 0, c2 = 0;
        int* a = oracle_radius_neighbors_batch(Q[t], NQ[t], S[t], NS[t], QL[t], SL[t], nb, radius * (t == 2 ? 2.f : 1.f), &c1);
        int* b = oracle_radius_neighbors_batch_reforder(Q[t], NQ[t], S[t], NS[t], QL[t], SL[t], nb, radius * (t == 2 ? 2.f : 1.f), &c2);
        CHECK(c1 == c2, "table %d: column counts %d vs %d", t, c1, c2);
        if (a && b && c1 == c2) {
            long diff = 0;                               /* same SET per row (the orders differ inside tie groups only) */
            for (int r = 0; r < NQ[t]; ++r) {
                long sa = 0, sb = 0;
                for (int c = 0; c < c1; ++c) { sa += a[(long)r * c1 + c]; sb += b[(long)r * c1 + c]; }
                diff += sa != sb;
            }
            CHECK(diff == 0, "table %d: %ld rows hold different sets in the two orders", t, diff);
        }
#ifdef WITH_REF
        if (ref_ok) {
            int c3 = 0;
            int* r = ref_batch_query(Q[t], NQ[t], S[t], NS[t], QL[t], SL[t], nb, radius * (t == 2 ? 2.f : 1.f), &c3);
            CHECK(c3 == c2, "table %d: reference has %d columns, restatement %d", t, c3, c2);
            if (r && b && c3 == c2)
                CHECK(memcmp(r, b, sizeof(int) * (size_t)NQ[t] * (size_t)c2) == 0, "table %d differs from the reference entry for entry", t);
            if (r) ref_free(r);
        }
#endif
        if (a) oracle_free(a);
        if (b) oracle_free(b);
    }
    free(sub);
    free(p);
}

int main(void) {
    /* unordered_map order on adversarial keys: multiples of the first bucket counts collide in one bucket */
    enum { M = 3000 };
    uint64_t* k = (uint64_t*)malloc(sizeof(uint64_t) * M);
    int* o1 = (int*)malloc(sizeof(int) * M);
    for (int i = 0; i < M; ++i) k[i] = (uint64_t)i * ((i & 1) ? 13u : 541u) + ((uint64_t)(i % 7) << 40);
    /* keys must be distinct: make them so */
    for (int i = 0; i < M; ++i) k[i] = k[i] * 4096u + (uint64_t)i;
    CHECK(oracle_umap_order(k, M, o1) == 0, "oracle_umap_order failed");
#ifdef WITH_REF
    {
        int* o2 = (int*)malloc(sizeof(int) * M);
        ref_umap_order((const unsigned long long*)k, M, o2);
        CHECK(memcmp(o1, o2, sizeof(int) * M) == 0, "unordered_map order differs from this toolchain's libstdc++");
        free(o2);
    }
#endif
    free(k);
    free(o1);
    one_case(1500, 1300, 0.75f, 0.f, 0.05f, 0.0625f);      /* the mini pair's shape */
    one_case(900, 0, 0.6f, 0.f, 0.05f, 0.0625f);           /* an EMPTY second cloud */
    one_case(1, 1, 0.1f, 0.f, 0.05f, 0.0625f);             /* one point per cloud */
    one_case(1200, 1100, 0.5f, 64.f, 0.05f, 0.0625f);      /* lattice: duplicates, rows full of equal distances */
    one_case(2500, 2500, 0.2f, 0.f, 0.02f, 0.05f);         /* dense: rows of hundreds of hits */
    if (fails) { fprintf(stderr, "asan_driver: %d check(s) failed\n", fails); return 1; }
    printf("asan_driver ok%s\n",
#ifdef WITH_REF
           " (restatement == unmodified reference under ASan/UBSan)"
#else
           " (restatement only: reference not present)"
#endif
    );
    return 0;
}
