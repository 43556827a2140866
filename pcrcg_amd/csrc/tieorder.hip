// tieorder.hip -- the reference's neighbour ORDER inside groups of exactly equal distance, on gfx950.
//
// pcrcg_radius_query (radius.hip) returns every row in ascending (d2, index) order.  The reference
// (ref:cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-333) sorts by d2 only, with an unstable
// std::sort over the hits in the order nanoflann's KD-tree traversal found them -- so inside a group of
// EXACTLY equal d2 its order is decided by
//   * nanoflann 1.3.0 (zip:cpp_utils/nanoflann/nanoflann.hpp): the tree build (buildIndex :1190-1203,
//     computeBoundingBox :1318-1338, divideTree :857-905, middleSplit_ :909-957, planeSplit :967-1003,
//     leaf size 10 from neighbors.cpp:245) and the traversal (findNeighbors :1221-1243,
//     computeInitialDistances :1005-1022, searchLevel :1348-1410, RadiusResultSet::addPoint :246-250), and
//   * libstdc++'s std::sort (introsort: median-of-3 pivot, unguarded Hoare partition, threshold 16, heap sort
//     after 2*lg(n) levels, final insertion sort) with IndexDist_Sorter (:208-214, compares d2 only).
// With the `[:, :limit]` cut of ref:datasets/dataloader.py:65-69 that order decides WHICH members of a tie group
// survive, and column 0 of an upsample table decides which feature closest_pool copies, so it is part of the
// reference's result.  Rows without a tie have one possible order; rows with a tie (reported by
// pcrcg_radius_query_ex) are redone here:
//
//   forest build   one tree per cloud, all clouds of all pyramid levels in ONE forest built by ONE launch.  A node's
//                  split (dimension, value, index) and the Hoare partition of its index range are reproduced
//                  exactly; the partition is the sequential part of the reference, and is parallelised through the
//                  identity  "a pass swaps the k-th misplaced element from the left with the k-th misplaced element
//                  from the right":
//                    nodes > 1024 points  one 512-thread workgroup per node (block prefix sums, four gathers in
//                                         flight per thread).  It keeps a child that holds more than 3/4 of the node
//                                         (skewed trees cannot explode the task count) and, below 4096 points, all
//                                         its big descendants (a hand-off costs more than such a split);
//                    nodes <= 1024 points the whole subtree inside one workgroup with its points in LDS, one
//                                         wavefront per node of a level (ballot / popcount prefix sums).
//                  Nodes are TASKS of a persistent kernel: workgroups pop node ids from a device-side queue and push
//                  the children they do not keep (release fence before the push, acquire fence after the pop: the
//                  eight XCDs have separate L2s), until no task is queued or running.
//   reorder        one wavefront per row, all tables of a pair in one launch: the reference's traversal with an
//                  explicit stack in LDS (leaf points tested 64 at a time, appended in order with ballot /
//                  popcount); then std::sort is replayed on the LDS list: its introsort partition phase step by step
//                  on lane 0 (nothing up to 16 hits), its final insertion sort -- a STABLE sort of what that phase
//                  left -- as a rank sort on all lanes; the wavefront writes the row.
//
// The oracle's CPU restatement (oracle/front_end.c: oracle_radius_neighbors_batch_reforder) is checked against the
// unmodified reference entry for entry; this file is checked against both (tests/test_tieorder_gpu.py).
// Compiled with -ffp-contract=off (every fp32 product and sum rounds separately, like the reference build).
#include <cfloat>
#include <cstdlib>

#include "block_scan.h"
#include "common.h"

namespace pcrcg {
namespace {

typedef unsigned long long u64;

constexpr int kLeafMax = 10;          // KDTreeSingleIndexAdaptorParams(10), neighbors.cpp:245
constexpr int kSubMax = 1024;         // nodes up to this size are finished inside one workgroup (LDS)
constexpr int kSubLevelNodes = 96;    // disjoint ranges of >= 11 points inside 1024 points
constexpr int kSubThreads = 512;       // = kBigThreads: big nodes and LDS subtrees are tasks of ONE kernel
constexpr int kBigThreads = 512;      // each scan step of a big node covers kBigThreads * kBigVec positions
constexpr int kBigVec = 4;
constexpr int kOwnMax = 4096;          // a workgroup that splits a node up to this size also splits its big descendants itself
constexpr int kForestBlocks = 48;     // workgroups of the forest kernel (they pull tasks from a device-side queue).  Each is 512
                                      // threads that mostly WAIT for tasks while the top of every tree is split by one workgroup:
                                      // beside the engine's other streams more of them only take wavefront slots and LDS from
                                      // kernels that have work (S30k two-pair builds asked for 94: 502 pairs/s, 48: 520, 32: 520,
                                      // 24: 510, 160: 457; K120k 160: 119.5, 48: 133.1, 16: 131.7), and alone the build is as fast
                                      // (S30k 2.00 vs 1.94 ms per two-pair pyramid) or faster (K120k 5.41 vs 6.16 ms)
constexpr int kSpinLimitDefault = 1 << 18;   // polls of an empty queue before a workgroup gives up (status 1)
constexpr int kTravStack = 128;       // pending far children per query (<= tree depth)
constexpr int kReorderWaves = 4;      // rows per workgroup of the reorder kernel
constexpr int kMaxRow = 8192;         // longest row the reorder kernel stages: two lists of W u64 in the 160 KB LDS of a CU
                                      // (rows beyond 3 500 hits: one wavefront per workgroup with more than 64 KB of dynamic LDS)

constexpr int kStUnfinished = 1;      // the forest kernel gave up waiting for work (never seen)
constexpr int kStStack = 2;           // traversal stack overflow
constexpr int kStCount = 3;           // tree search and cell-grid search disagree on a row's hit count
constexpr int kStWidth = 4;           // a row holds more hits than the staging width

struct alignas(64) KdNode {   // 64 bytes, one cache line (aligned: lets the compiler merge the field loads)
    int left, right;        // range in vind
    int divfeat;            // -1: leaf
    float divlow, divhigh;
    int child1, child2;
    int is_root;
    float lo[3], hi[3];     // box handed down by the parent (root: the cloud's tight box, read by the search)
    int pad[2];
};

struct KdCtl {
    int node_count, status;
    int q_head, q_tail;     // task queue: slots [q_head, q_tail) are waiting
    int pending;            // tasks queued or running; 0 = forest finished
    int pad[3];
};

struct KdView {
    KdCtl* ctl;
    int* soff;       // [nb+1]
    KdNode* nodes;   // [2*ns + nb + 2]
    int* vind;       // [ns] global support index
    int* scratch;    // [ns]
    int* taskq;      // [2*ns + nb + 2] node ids, -1 = slot not written yet
    int qcap;
};

inline size_t forest_bytes(int ns, int nb) {
    const size_t N = (size_t)(ns > 0 ? ns : 0);
    return carve_bytes(1, sizeof(KdCtl)) + carve_bytes((size_t)nb + 1, sizeof(int)) +
           carve_bytes(2 * N + nb + 2, sizeof(KdNode)) + 2 * carve_bytes(N + 1, sizeof(int)) +
           carve_bytes(2 * N + nb + 2, sizeof(int));
}

inline KdView forest_view(void* ws, size_t bytes, int ns, int nb, bool* ok) {
    const size_t N = (size_t)(ns > 0 ? ns : 0);
    Carver cv(ws, bytes);
    KdView v;
    v.ctl = reinterpret_cast<KdCtl*>(cv.take<char>(sizeof(KdCtl)));
    v.soff = cv.take<int>((size_t)nb + 1);
    v.nodes = cv.take<KdNode>(2 * N + nb + 2);
    v.vind = cv.take<int>(N + 1);
    v.scratch = cv.take<int>(N + 1);
    v.qcap = (int)(2 * N + nb + 2);
    v.taskq = cv.take<int>((size_t)v.qcap);
    *ok = cv.ok();
    return v;
}

// ------------------------------------------------------------------------------------------------
// forest build
// ------------------------------------------------------------------------------------------------
// bases.per_level > 0: the clouds are LEVELS of bases.per_level clouds each, level l's rows starting at row bases.base[l] of
// `sup` (the pyramid builder keeps every level at a place sized by its row BOUND, so the levels are not contiguous); a
// cloud's end is its root node's `right`, never the next cloud's start.
struct KdBases {
    int per_level;
    int base[PCRCG_MAX_LEVELS];
};
__global__ void __launch_bounds__(256) k_kd_init(const int* __restrict__ slen, int ns, int nb, KdView v, KdBases bases) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        KdCtl* c = v.ctl;
        c->node_count = nb;
        c->status = 0;
        c->q_head = 0;
        int s = 0, tasks = 0;
        for (int b = 0; b < nb; ++b) {
            const int n = slen[b];
            if (bases.per_level > 0 && b % bases.per_level == 0) s = bases.base[b / bases.per_level];
            KdNode nd;
            nd.left = s;
            nd.right = s + n;
            nd.divfeat = -1;
            nd.divlow = nd.divhigh = 0.f;
            nd.child1 = nd.child2 = -1;
            nd.is_root = 1;
            for (int d = 0; d < 3; ++d) { nd.lo[d] = 0.f; nd.hi[d] = 0.f; }
            nd.pad[0] = nd.pad[1] = 0;
            v.nodes[b] = nd;
            if (n > 0) v.taskq[tasks++] = b;          // every non-empty cloud is a task (small roots need their box)
            v.soff[b] = s;
            s += n;
        }
        v.soff[nb] = s;
        for (int i = tasks; i < nb; ++i) v.taskq[i] = -1;
        c->q_tail = tasks;
        c->pending = tasks;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.qcap; i += gridDim.x * blockDim.x) {
        if (i < ns) v.vind[i] = i;
        if (i >= nb) v.taskq[i] = -1;                  // slots 0..nb-1 belong to the roots (thread 0 above)
    }
}

// ---- device-side task queue of the forest kernel (one thread of a workgroup calls these) ----
__device__ __forceinline__ int ald(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void kd_push_task(const KdView& v, int id) {
    // the node record and its vind range were written before the caller's release fence
    atomicAdd(&v.ctl->pending, 1);
    const int slot = atomicAdd(&v.ctl->q_tail, 1);
    __hip_atomic_store(&v.taskq[slot], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // caller fenced already
}

// -> node id, or -1 when the forest is finished (or the wait was abandoned)
__device__ __noinline__ int kd_pop_task(KdCtl* ctl, const int* taskq, int spin_limit) {
    for (int spins = 0;;) {
        const int h = ald(&ctl->q_head), t = ald(&ctl->q_tail);
        if (h < t) {
            if (atomicCAS(&ctl->q_head, h, h + 1) != h) continue;
            int id;
            while ((id = __hip_atomic_load(&taskq[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0) {   // acquire fence: caller
                __builtin_amdgcn_s_sleep(2);
                if (++spins > spin_limit) { ctl->status = kStUnfinished; return -1; }
            }
            return id;
        }
        if (ald(&ctl->pending) == 0) return -1;
        __builtin_amdgcn_s_sleep(32);
        if (++spins > spin_limit) { ctl->status = kStUnfinished; return -1; }
    }
}

// children of a split node: allocate and write their records (queueing is the caller's business)
__device__ __forceinline__ void kd_emit_children(const KdView& v, int id, int left, int right, int idx, int cutfeat,
                                                 float cutval, float divlow, float divhigh, const float* lo,
                                                 const float* hi, int* c1_out, int* c2_out, int prealloc = -1) {
    const int c1 = prealloc >= 0 ? prealloc : atomicAdd(&v.ctl->node_count, 2), c2 = c1 + 1;
    for (int side = 0; side < 2; ++side) {
        KdNode* ch = &v.nodes[c1 + side];
        ch->left = side == 0 ? left : left + idx;
        ch->right = side == 0 ? left + idx : right;
        ch->divfeat = -1;
        ch->divlow = ch->divhigh = 0.f;
        ch->child1 = ch->child2 = -1;
        ch->is_root = 0;
        for (int d = 0; d < 3; ++d) {                                            // divideTree :891-898
            ch->lo[d] = (side == 1 && d == cutfeat) ? cutval : lo[d];
            ch->hi[d] = (side == 0 && d == cutfeat) ? cutval : hi[d];
        }
    }
    KdNode* nd = &v.nodes[id];
    nd->divfeat = cutfeat;
    nd->divlow = divlow;
    nd->divhigh = divhigh;
    nd->child1 = c1;
    nd->child2 = c2;
    *c1_out = c1;
    *c2_out = c2;
}

__device__ __forceinline__ float sel3(const float* a, int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : a[2]); }

// middleSplit_ :909-944: split dimension and value from the handed-down box and the node's actual min / max
__device__ __forceinline__ void kd_choose_split(const float* lo, const float* hi, const float* mn, const float* mx,
                                                int* cutfeat, float* cutval) {
    const float EPS = 0.00001f;
    float max_span = hi[0] - lo[0];
    for (int d = 1; d < 3; ++d) {
        const float span = hi[d] - lo[d];
        if (span > max_span) max_span = span;
    }
    float max_spread = -1.f;
    int cf = 0;
    for (int d = 0; d < 3; ++d) {
        const float span = hi[d] - lo[d];
        if (span > (1 - EPS) * max_span) {
            const float spread = mx[d] - mn[d];
            if (spread > max_spread) { cf = d; max_spread = spread; }
        }
    }
    const float split_val = (sel3(lo, cf) + sel3(hi, cf)) / 2;
    const float mnc = sel3(mn, cf), mxc = sel3(mx, cf);
    *cutfeat = cf;
    *cutval = split_val < mnc ? mnc : (split_val > mxc ? mxc : split_val);
}

// One pass of planeSplit :967-1003 over positions [lo, hi) of the node, by a whole workgroup.  The sequential loop
// swaps the k-th element from the left that belongs right with the k-th element from the right that belongs
// left, until the two scans cross -- i.e. exactly the misplaced elements on either side of the final boundary
// B = lo + #small, paired in that order.
template <bool STRICT>
__device__ void kd_hoare_block(const float* __restrict__ sup, int* vind, int* scratch, int lo, int hi, int nsmall,
                               int feat, float cut, int* smem) {
    const int B = lo + nsmall;
    int m = 0;
    for (int start = lo; start < B; start += kBigThreads * kBigVec) {
        const int p0 = start + (int)threadIdx.x * kBigVec;
        bool flag[kBigVec];
        int c = 0;
#pragma unroll
        for (int u = 0; u < kBigVec; ++u) {
            flag[u] = false;
            if (p0 + u < B) {
                const float val = sup[3 * (long)vind[p0 + u] + feat];
                flag[u] = !(STRICT ? val < cut : val <= cut);
            }
            c += flag[u] ? 1 : 0;
        }
        int tot;
        int r = block_excl_scan_i32<kBigThreads>(c, &tot, smem);
#pragma unroll
        for (int u = 0; u < kBigVec; ++u)
            if (flag[u]) scratch[lo + m + r++] = p0 + u;
        m += tot;
    }
    int m2 = 0;
    for (int start = B; start < hi; start += kBigThreads * kBigVec) {
        const int p0 = start + (int)threadIdx.x * kBigVec;
        bool flag[kBigVec];
        int c = 0;
#pragma unroll
        for (int u = 0; u < kBigVec; ++u) {
            flag[u] = false;
            if (p0 + u < hi) {
                const float val = sup[3 * (long)vind[p0 + u] + feat];
                flag[u] = STRICT ? val < cut : val <= cut;
            }
            c += flag[u] ? 1 : 0;
        }
        int tot;
        int r = block_excl_scan_i32<kBigThreads>(c, &tot, smem);
#pragma unroll
        for (int u = 0; u < kBigVec; ++u)
            if (flag[u]) scratch[B + m2 + r++] = p0 + u;
        m2 += tot;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < m; k += kBigThreads) {
        const int a = scratch[lo + k], b = scratch[B + (m - 1 - k)];
        const int t = vind[a];
        vind[a] = vind[b];
        vind[b] = t;
    }
    __syncthreads();
}

__device__ __forceinline__ float wave_min_f(float x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x = fminf(x, __shfl_xor(x, d, 64));
    return x;
}
__device__ __forceinline__ float wave_max_f(float x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x = fmaxf(x, __shfl_xor(x, d, 64));
    return x;
}
__device__ __forceinline__ int wave_sum_i(int x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}

// nodes > kSubMax points: one workgroup per node of this level
// a node > kSubMax points: the whole workgroup partitions it, queues its children, and -- while one child keeps more
// than 3/4 of the points -- goes on with that child itself
__device__ void kd_big_task(const float* __restrict__ sup, const KdView& v, int id) {
    __shared__ float s_red[kBigThreads / 64][8];
    __shared__ int s_redi[kBigThreads / 64][2];
    __shared__ int s_scan[kBigThreads / 64];
    __shared__ int s_next[2];
    __shared__ int s_stack[64], s_sp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_sp = 0;
    __syncthreads();
    for (;;) {      // this node; then the big descendants this workgroup kept for itself (s_stack)
        int left, right, is_root;
        float lo[3], hi[3];
        {
            const KdNode nd = v.nodes[id];
            left = nd.left;
            right = nd.right;
            is_root = nd.is_root;
            for (int d = 0; d < 3; ++d) { lo[d] = nd.lo[d]; hi[d] = nd.hi[d]; }
        }
        for (;;) {   // this node, then -- while one child keeps more than 3/4 of the points -- that child
            const int n = right - left;
            int* ind = v.vind + left;
            int* scr = v.scratch + left;
            // actual min / max of the node (computeMinMax :836-848; for a root also the tree's box :1318-1338)
            float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            for (int p = threadIdx.x; p < n; p += 4 * kBigThreads) {      // 4 independent gathers in flight per thread
                int g[4];
                float c[4][3];
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] = p + u * kBigThreads < n ? ind[p + u * kBigThreads] : ind[p];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    for (int d = 0; d < 3; ++d) c[u][d] = sup[3 * (long)g[u] + d];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], c[u][d]); mx[d] = fmaxf(mx[d], c[u][d]); }
            }
            for (int d = 0; d < 3; ++d) { mn[d] = wave_min_f(mn[d]); mx[d] = wave_max_f(mx[d]); }
            if (lane == 0) for (int d = 0; d < 3; ++d) { s_red[wave][d] = mn[d]; s_red[wave][3 + d] = mx[d]; }
            __syncthreads();
            for (int d = 0; d < 3; ++d) {
                mn[d] = s_red[0][d];
                mx[d] = s_red[0][3 + d];
                for (int ww = 1; ww < kBigThreads / 64; ++ww) { mn[d] = fminf(mn[d], s_red[ww][d]); mx[d] = fmaxf(mx[d], s_red[ww][3 + d]); }
            }
            __syncthreads();
            if (is_root) for (int d = 0; d < 3; ++d) { lo[d] = mn[d]; hi[d] = mx[d]; }
            int cutfeat;
            float cutval;
            kd_choose_split(lo, hi, mn, mx, &cutfeat, &cutval);
            // class counts; largest value below / smallest value above the cut (the children's tight boxes on cutfeat)
            int nless = 0, neq = 0;
            float max_less = -FLT_MAX, min_greater = FLT_MAX;
            for (int p = threadIdx.x; p < n; p += 4 * kBigThreads) {
                int g[4];
                float val[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] = p + u * kBigThreads < n ? ind[p + u * kBigThreads] : -1;
#pragma unroll
                for (int u = 0; u < 4; ++u) val[u] = g[u] >= 0 ? sup[3 * (long)g[u] + cutfeat] : 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (g[u] < 0) continue;
                    if (val[u] < cutval) { ++nless; max_less = fmaxf(max_less, val[u]); }
                    else if (val[u] == cutval) ++neq;
                    else min_greater = fminf(min_greater, val[u]);
                }
            }
            nless = wave_sum_i(nless);
            neq = wave_sum_i(neq);
            max_less = wave_max_f(max_less);
            min_greater = wave_min_f(min_greater);
            if (lane == 0) { s_redi[wave][0] = nless; s_redi[wave][1] = neq; s_red[wave][6] = max_less; s_red[wave][7] = min_greater; }
            __syncthreads();
            nless = neq = 0;
            max_less = -FLT_MAX;
            min_greater = FLT_MAX;
            for (int ww = 0; ww < kBigThreads / 64; ++ww) {
                nless += s_redi[ww][0];
                neq += s_redi[ww][1];
                max_less = fmaxf(max_less, s_red[ww][6]);
                min_greater = fminf(min_greater, s_red[ww][7]);
            }
            __syncthreads();
            kd_hoare_block<true>(sup, ind, scr, 0, n, nless, cutfeat, cutval, s_scan);          // -> lim1 = nless
            if (neq > 0) kd_hoare_block<false>(sup, ind, scr, nless, n, neq, cutfeat, cutval, s_scan);   // -> lim2 = nless + neq
            const int lim1 = nless, lim2 = nless + neq, half = n / 2;
            const int idx = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);                    // middleSplit_ :951-956
            const float divlow = idx > lim1 ? cutval : max_less;
            const float divhigh = idx < lim2 ? cutval : min_greater;
            const int big_side = idx >= n - idx ? 0 : 1, big_n = big_side == 0 ? idx : n - idx;
            const int keep = (big_n > kSubMax && 4l * big_n > 3l * n) ? big_side : -1;
            // every wavefront's stores have left the CU at the barrier (workgroup release); ONE agent-scope release
            // by thread 0 below then writes this XCD's L2 back before a child is queued (other XCDs have their own L2)
            __syncthreads();
            if (threadIdx.x == 0) {
                if (is_root) for (int d = 0; d < 3; ++d) { v.nodes[id].lo[d] = lo[d]; v.nodes[id].hi[d] = hi[d]; }
                int c1, c2;
                kd_emit_children(v, id, left, right, idx, cutfeat, cutval, divlow, divhigh, lo, hi, &c1, &c2);
                // hand-offs cost more than the split of a few thousand points: below kOwnMax only the LDS subtrees
                // go to other workgroups, big children stay here (s_stack)
                bool fenced = false;
                for (int side = 0; side < 2; ++side) {
                    const int cn = side == 0 ? idx : n - idx, cid = side == 0 ? c1 : c2;
                    if (side == keep || cn <= kLeafMax) continue;
                    if (cn > kSubMax && n <= kOwnMax && s_sp < 64) { s_stack[s_sp++] = cid; continue; }
                    if (!fenced) { __threadfence(); fenced = true; }     // the agent-scope release of this hand-off
                    kd_push_task(v, cid);
                }
                s_next[0] = keep == 0 ? c1 : c2;
            }
            __syncthreads();
            if (keep < 0) break;
            id = s_next[0];
            if (keep == 0) right = left + idx; else left = left + idx;
            for (int d = 0; d < 3; ++d) {
                if (keep == 0 && d == cutfeat) hi[d] = cutval;
                if (keep == 1 && d == cutfeat) lo[d] = cutval;
            }
            is_root = 0;
            __syncthreads();
        }
        if (s_sp == 0) break;                      // uniform: written before the barrier that ended the loop above
        __syncthreads();
        if (threadIdx.x == 0) s_next[1] = s_stack[--s_sp];
        __syncthreads();
        id = s_next[1];
    }
}

// One pass of planeSplit over positions [lo, hi) of a node whose points sit in LDS, by one wavefront: the same
// pairing as kd_hoare_block with ballot / popcount prefix sums.  cv = the node's coordinate on the cut dimension
// by LDS slot, ord = slot by position, list = scratch by position.
template <bool STRICT>
__device__ __forceinline__ void kd_hoare_wave(const float* cv, int* ord, int* list, int lo, int hi, int nsmall,
                                              float cut, int lane) {
    const int B = lo + nsmall;
    const u64 below = (1ull << lane) - 1ull;
    int m = 0;
    for (int start = lo; start < B; start += 64) {
        const int p = start + lane;
        bool flag = false;
        if (p < B) {
            const float val = cv[ord[p]];
            flag = !(STRICT ? val < cut : val <= cut);
        }
        const u64 mask = __ballot(flag);
        if (flag) list[lo + m + __popcll(mask & below)] = p;
        m += __popcll(mask);
    }
    int m2 = 0;
    for (int start = B; start < hi; start += 64) {
        const int p = start + lane;
        bool flag = false;
        if (p < hi) {
            const float val = cv[ord[p]];
            flag = STRICT ? val < cut : val <= cut;
        }
        const u64 mask = __ballot(flag);
        if (flag) list[B + m2 + __popcll(mask & below)] = p;
        m2 += __popcll(mask);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < m; k += 64) {
        const int a = list[lo + k], b = list[B + (m - 1 - k)];
        const int t = ord[a];
        ord[a] = ord[b];
        ord[b] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// record a split node of an LDS subtree and queue its children that are not leaves
__device__ __forceinline__ void kd_sub_emit(const KdView& v, int nid, int left, int l, int r, int idx, int cutfeat,
                                            float cutval, float divlow, float divhigh, const float* lo, const float* hi,
                                            int child_ids, int* nq_id, int* nq_l, int* nq_r, float (*nq_box)[6], int* nq_cnt) {
    int c1, c2;
    kd_emit_children(v, nid, left + l, left + r, idx, cutfeat, cutval, divlow, divhigh, lo, hi, &c1, &c2, child_ids);
    for (int side = 0; side < 2; ++side) {
        const int cl = side == 0 ? l : l + idx, cr = side == 0 ? l + idx : r;
        if (cr - cl > kLeafMax) {
            const int slot = atomicAdd(nq_cnt, 1);
            nq_id[slot] = side == 0 ? c1 : c2;
            nq_l[slot] = cl;
            nq_r[slot] = cr;
            for (int d = 0; d < 3; ++d) { nq_box[slot][d] = lo[d]; nq_box[slot][3 + d] = hi[d]; }
            if (side == 0) nq_box[slot][3 + cutfeat] = cutval; else nq_box[slot][cutfeat] = cutval;
        }
    }
}

// nodes <= kSubMax points: one workgroup builds the whole subtree with its points in LDS; the nodes of a level are
// dealt to the workgroup's wavefronts (ballot / popcount partition).  (One LANE per small node running the
// reference's sequential loops literally was measured too: slower, 200 vs 122 us per pyramid.)
__device__ void kd_sub_task(const float* __restrict__ sup, const KdView& v, int id) {
    __shared__ int s_gi[kSubMax];
    __shared__ float s_c[3][kSubMax];
    __shared__ int s_ord[kSubMax];
    __shared__ int s_list[kSubMax];
    __shared__ int q_id[2][kSubLevelNodes], q_l[2][kSubLevelNodes], q_r[2][kSubLevelNodes];
    __shared__ float q_box[2][kSubLevelNodes][6];
    __shared__ int q_cnt[2];
    __shared__ int s_base;
    constexpr int kWaves = kSubThreads / 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    {
        const KdNode nd = v.nodes[id];
        const int left = nd.left, n = nd.right - nd.left;
        for (int p = threadIdx.x; p < n; p += kSubThreads) {
            const int g = v.vind[left + p];
            s_gi[p] = g;
            s_ord[p] = p;
            for (int d = 0; d < 3; ++d) s_c[d][p] = sup[3 * (long)g + d];
        }
        if (threadIdx.x == 0) {
            q_cnt[0] = 0;
            q_cnt[1] = 0;
        }
        __syncthreads();
        if (wave == 0) {
            float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            if (nd.is_root) {      // a small cloud: its tight box is the tree's box (:1318-1338)
                for (int p = lane; p < n; p += 64)
                    for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], s_c[d][p]); mx[d] = fmaxf(mx[d], s_c[d][p]); }
                for (int d = 0; d < 3; ++d) { mn[d] = wave_min_f(mn[d]); mx[d] = wave_max_f(mx[d]); }
                if (lane == 0) for (int d = 0; d < 3; ++d) { v.nodes[id].lo[d] = mn[d]; v.nodes[id].hi[d] = mx[d]; }
            }
            if (n > kLeafMax && lane == 0) {
                q_id[0][0] = id;
                q_l[0][0] = 0;
                q_r[0][0] = n;
                for (int d = 0; d < 3; ++d) {
                    q_box[0][0][d] = nd.is_root ? mn[d] : nd.lo[d];
                    q_box[0][0][3 + d] = nd.is_root ? mx[d] : nd.hi[d];
                }
                q_cnt[0] = 1;
            }
        }
        __syncthreads();
        int cur = 0;
        while (q_cnt[cur] > 0) {
            const int cnt = q_cnt[cur];
            if (threadIdx.x == 0) s_base = atomicAdd(&v.ctl->node_count, 2 * cnt);   // ids of this level's children
            __syncthreads();
            const int base = s_base;
            for (int i = wave; i < cnt; i += kWaves) {
                const int nid = q_id[cur][i], l = q_l[cur][i], r = q_r[cur][i], count = r - l;
                float lo[3], hi[3], nmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, nmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
                for (int d = 0; d < 3; ++d) { lo[d] = q_box[cur][i][d]; hi[d] = q_box[cur][i][3 + d]; }
                for (int p = l + lane; p < r; p += 64) {
                    const int o = s_ord[p];
                    for (int d = 0; d < 3; ++d) { nmn[d] = fminf(nmn[d], s_c[d][o]); nmx[d] = fmaxf(nmx[d], s_c[d][o]); }
                }
                for (int d = 0; d < 3; ++d) { nmn[d] = wave_min_f(nmn[d]); nmx[d] = wave_max_f(nmx[d]); }
                int cutfeat;
                float cutval;
                kd_choose_split(lo, hi, nmn, nmx, &cutfeat, &cutval);
                const float* cv = s_c[cutfeat];
                int nless = 0, neq = 0;
                float max_less = -FLT_MAX, min_greater = FLT_MAX;
                for (int p = l + lane; p < r; p += 64) {
                    const float val = cv[s_ord[p]];
                    if (val < cutval) { ++nless; max_less = fmaxf(max_less, val); }
                    else if (val == cutval) ++neq;
                    else min_greater = fminf(min_greater, val);
                }
                nless = wave_sum_i(nless);
                neq = wave_sum_i(neq);
                max_less = wave_max_f(max_less);
                min_greater = wave_min_f(min_greater);
                kd_hoare_wave<true>(cv, s_ord, s_list, l, r, nless, cutval, lane);
                if (neq > 0) kd_hoare_wave<false>(cv, s_ord, s_list, l + nless, r, neq, cutval, lane);
                const int lim1 = nless, lim2 = nless + neq, half = count / 2;
                const int idx = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);
                const float divlow = idx > lim1 ? cutval : max_less;
                const float divhigh = idx < lim2 ? cutval : min_greater;
                if (lane == 0)
                    kd_sub_emit(v, nid, left, l, r, idx, cutfeat, cutval, divlow, divhigh, lo, hi, base + 2 * i, q_id[cur ^ 1],
                                q_l[cur ^ 1], q_r[cur ^ 1], q_box[cur ^ 1], &q_cnt[cur ^ 1]);
            }
            __syncthreads();
            if (threadIdx.x == 0) q_cnt[cur] = 0;
            cur ^= 1;
            __syncthreads();
        }
        for (int p = threadIdx.x; p < n; p += kSubThreads) v.vind[left + p] = s_gi[s_ord[p]];
        __syncthreads();
    }
}

// The forest in one launch: workgroups pull nodes from the device-side queue until no task is queued or running.
// A node's record and index range are written by the workgroup that split its parent, possibly on another CU:
// release fence before the push, acquire fence (L1 invalidate) after the pop.
__global__ void __launch_bounds__(kBigThreads) k_kd_forest(const float* __restrict__ sup, KdView v, int spin_limit) {
    __shared__ int s_task;
    static_assert(kBigThreads == kSubThreads, "one workgroup shape for both task kinds");
    bool more = true;
    while (more) {
        if (threadIdx.x == 0) s_task = kd_pop_task(v.ctl, v.taskq, spin_limit);
        __syncthreads();
        const int id = s_task;
        more = id >= 0;
        if (more) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int n = v.nodes[id].right - v.nodes[id].left;
            if (n > kSubMax) kd_big_task(sup, v, id);
            else kd_sub_task(sup, v, id);      // a subtree's results are read by the next kernel only
        }
        __syncthreads();
        if (more && threadIdx.x == 0) atomicSub(&v.ctl->pending, 1);
    }
}

// ------------------------------------------------------------------------------------------------
// libstdc++ std::sort(first, last, comp) with comp(a, b) = a.d2 < b.d2, replayed on packed (d2 bits << 32 | index)
// entries.  d2 >= +0, so the unsigned order of the upper word is the float order.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool lt(u64 a, u64 b) { return (unsigned)(a >> 32) < (unsigned)(b >> 32); }

__device__ __forceinline__ void adjust_heap(u64* first, int hole, int len, u64 value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (lt(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;                       // __push_heap
    while (hole > top && lt(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

__device__ __forceinline__ void heap_sort(u64* first, int len) {      // __partial_sort(first, last, last)
    if (len >= 2) {
        for (int parent = (len - 2) / 2;; --parent) {  // __make_heap
            adjust_heap(first, parent, len, first[parent]);
            if (parent == 0) break;
        }
    }
    for (int last = len; last > 1;) {                  // __sort_heap
        --last;
        const u64 value = first[last];
        first[last] = first[0];
        adjust_heap(first, 0, last, value);
    }
}

__device__ __forceinline__ void std_sort_partition_phase(u64* a, int n) {
    if (n == 0) return;
    int lg = 0;
    while ((n >> (lg + 1)) > 0) ++lg;
    // __introsort_loop: the recursion on [cut, last) and the loop on [first, cut) touch disjoint ranges, so an
    // explicit stack in any order gives the same array
    int st_first[64], st_last[64], st_depth[64], sp = 0;
    st_first[0] = 0; st_last[0] = n; st_depth[0] = 2 * lg; sp = 1;
    while (sp > 0) {
        --sp;
        int first = st_first[sp], last = st_last[sp], depth = st_depth[sp];
        while (last - first > 16) {
            if (depth == 0) { heap_sort(a + first, last - first); break; }
            --depth;
            const int mid = first + (last - first) / 2;
            const int ia = first + 1, ib = mid, ic = last - 1;          // __move_median_to_first(first, a, b, c)
            int med;
            if (lt(a[ia], a[ib])) med = lt(a[ib], a[ic]) ? ib : (lt(a[ia], a[ic]) ? ic : ia);
            else med = lt(a[ia], a[ic]) ? ia : (lt(a[ib], a[ic]) ? ic : ib);
            { const u64 t = a[first]; a[first] = a[med]; a[med] = t; }
            int lo = first + 1, hi = last;                                 // __unguarded_partition(first + 1, last, first)
            for (;;) {
                while (lt(a[lo], a[first])) ++lo;
                --hi;
                while (lt(a[first], a[hi])) --hi;
                if (!(lo < hi)) break;
                const u64 t = a[lo]; a[lo] = a[hi]; a[hi] = t;
                ++lo;
            }
            if (sp < 64) { st_first[sp] = lo; st_last[sp] = last; st_depth[sp] = depth; ++sp; }
            last = lo;
        }
    }
    // __final_insertion_sort (insertion sort of the first 16, unguarded inserts of the rest) moves an element only
    // past STRICTLY greater ones: it is the stable sort of what the loop above left, and the caller does that with
    // all lanes (stable_sort_wave) instead of one.
}

// the stable sort by d2 of a[0, n) into out[0, n), by the whole wavefront: rank = #smaller + #equal before
__device__ __forceinline__ void stable_sort_wave(const u64* a, u64* out, int n, int lane) {
    for (int i = lane; i < n; i += 64) {
        const u64 mine = a[i];
        const unsigned key = (unsigned)(mine >> 32);
        int r = 0;
        for (int j = 0; j < n; ++j) {
            const unsigned kj = (unsigned)(a[j] >> 32);
            r += (kj < key || (kj == key && j < i)) ? 1 : 0;
        }
        out[r] = mine;
    }
}

// ------------------------------------------------------------------------------------------------
// reorder: one wavefront per row, the tables of a pair in one launch
// ------------------------------------------------------------------------------------------------
struct ReorderJobs {
    pcrcg_reorder_job job[PCRCG_MAX_REORDER_JOBS];
    KdView view[PCRCG_MAX_REORDER_JOBS];          // the forest a job searches (its own, or the call's) ...
    const float* sup[PCRCG_MAX_REORDER_JOBS];     // ... and that forest's points
    int row_begin[PCRCG_MAX_REORDER_JOBS + 1];
    int njobs;
};

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
// the first 32 bytes of a node (left, right, divfeat, divlow, divhigh, child1, child2, is_root) through the SCALAR cache: the
// walk below is one dependent node fetch after the other, the node is the same for all lanes, and nothing in this kernel
// writes nodes (the forest kernel did, before this launch)
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x8 node_head(const KdNode* p) {
    i32x8 r;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    return r;
}
__device__ __forceinline__ float unif(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

__global__ void __launch_bounds__(kReorderWaves * 64) k_reorder(const ReorderJobs jobs, int W, int* __restrict__ status) {
    extern __shared__ u64 s_dyn[];
    // per wavefront: W staged hits, then a region that holds the traversal stack (node, mindistsq, dists[3] per
    // entry) first and the sorted row afterwards
    const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6)), waves = (int)(blockDim.x >> 6);
    const size_t second = (size_t)W > (size_t)kTravStack * 5 / 2 ? (size_t)W : (size_t)kTravStack * 5 / 2;
    const size_t per_wave = (size_t)W + second + 1;
    u64* mine = s_dyn + per_wave * wave;
    u64* sorted = mine + W;
    int* st_node = reinterpret_cast<int*>(mine + W);
    float* st_min = reinterpret_cast<float*>(st_node + kTravStack);
    float* st_d = st_min + kTravStack;      // [kTravStack][3]
    const int t = blockIdx.x * waves + wave;
    if (t >= jobs.row_begin[jobs.njobs]) return;
    int ji = 0;
    while (ji + 1 < jobs.njobs && t >= jobs.row_begin[ji + 1]) ++ji;
    const pcrcg_reorder_job& jb = jobs.job[ji];
    const KdView& v = jobs.view[ji];
    const float* __restrict__ sup = jobs.sup[ji];
    const int local = t - jobs.row_begin[ji];
    if (local == 0 && lane == 0 && v.ctl->status && status) *status = v.ctl->status;
    const int qi = jb.rows ? uni(jb.rows[local]) : local;
    const float* q = jb.q;
    int b = 0, qacc = 0;
    while (b < jb.nbq - 1 && qi >= qacc + jb.qlen[b]) { qacc += jb.qlen[b]; ++b; }
    const int root = jb.cloud0 + b;
    // (a cloud ends at its root's `right`: the next cloud may start elsewhere -- levels at places of their own, KdBases)
    int seg = v.soff[jb.cloud0], pad = v.nodes[jb.cloud0 + jb.nbq - 1].right - seg;
    if (jb.group > 0) {   // independent groups of clouds: relative to the query's own group
        const int g0 = jb.cloud0 + (b / jb.group) * jb.group, g1 = min(g0 + jb.group, jb.cloud0 + jb.nbq);
        seg = v.soff[g0];
        pad = v.nodes[g1 - 1].right - seg;
    }
    const float r2 = jb.radius * jb.radius;      // neighbors.cpp:226
    const float vx = q[3 * (long)qi], vy = q[3 * (long)qi + 1], vz = q[3 * (long)qi + 2];
    const u64 below = (1ull << lane) - 1ull;
    int n = 0, err = 0;
    const KdNode* rt = &v.nodes[root];
    if (rt->right > rt->left) {
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, distsq = 0.f;                  // computeInitialDistances :1005-1022
        if (vx < rt->lo[0]) { d0 = (vx - rt->lo[0]) * (vx - rt->lo[0]); distsq += d0; }
        if (vx > rt->hi[0]) { d0 = (vx - rt->hi[0]) * (vx - rt->hi[0]); distsq += d0; }
        if (vy < rt->lo[1]) { d1 = (vy - rt->lo[1]) * (vy - rt->lo[1]); distsq += d1; }
        if (vy > rt->hi[1]) { d1 = (vy - rt->hi[1]) * (vy - rt->hi[1]); distsq += d1; }
        if (vz < rt->lo[2]) { d2 = (vz - rt->lo[2]) * (vz - rt->lo[2]); distsq += d2; }
        if (vz > rt->hi[2]) { d2 = (vz - rt->hi[2]) * (vz - rt->hi[2]); distsq += d2; }
        int sp = 1;
        st_node[0] = root;
        st_min[0] = distsq;
        st_d[0] = d0; st_d[1] = d1; st_d[2] = d2;
        while (sp > 0) {
            --sp;
            __builtin_amdgcn_wave_barrier();
            int node = uni(st_node[sp]);
            const float mind = unif(st_min[sp]);
            d0 = unif(st_d[3 * sp]); d1 = unif(st_d[3 * sp + 1]); d2 = unif(st_d[3 * sp + 2]);
            for (;;) {                                                     // searchLevel :1348-1410
                const i32x8 nh = node_head(&v.nodes[node]);
                const int feat = nh[2];
                if (feat < 0) {
                    const int l = nh[0], r = nh[1];
                    for (int base = l; base < r; base += 64) {             // the leaf's points in vind order
                        const int i = base + lane;
                        bool hit = false;
                        float result = 0.0f;
                        int g = 0;
                        if (i < r) {
                            g = v.vind[i];
                            const float e0 = vx - sup[3 * (long)g], e1 = vy - sup[3 * (long)g + 1], e2 = vz - sup[3 * (long)g + 2];
                            result += e0 * e0;
                            result += e1 * e1;
                            result += e2 * e2;
                            hit = result < r2;
                        }
                        const u64 mask = __ballot(hit);
                        const int pos = n + __popcll(mask & below);
                        if (hit && pos < W) mine[pos] = ((u64)__float_as_uint(result) << 32) | (unsigned)g;
                        n += __popcll(mask);
                    }
                    break;
                }
                const float val = feat == 0 ? vx : (feat == 1 ? vy : vz);
                const float divlow = __int_as_float(nh[3]), divhigh = __int_as_float(nh[4]);
                const int child1 = nh[5], child2 = nh[6];
                const float diff1 = val - divlow, diff2 = val - divhigh;
                int best, other;
                float cut_dist;
                if ((diff1 + diff2) < 0) { best = child1; other = child2; cut_dist = (val - divhigh) * (val - divhigh); }
                else { best = child2; other = child1; cut_dist = (val - divlow) * (val - divlow); }
                // the far child is visited after the near subtree, with the same mindistsq / dists as here
                const float dcur = feat == 0 ? d0 : (feat == 1 ? d1 : d2);
                const float mind2 = mind + cut_dist - dcur;
                if (mind2 * 1.0f <= r2) {
                    if (sp < kTravStack) {
                        st_node[sp] = other;        // every lane stores the same value
                        st_min[sp] = mind2;
                        st_d[3 * sp] = feat == 0 ? cut_dist : d0;
                        st_d[3 * sp + 1] = feat == 1 ? cut_dist : d1;
                        st_d[3 * sp + 2] = feat == 2 ? cut_dist : d2;
                        ++sp;
                    } else {
                        err = kStStack;
                    }
                }
                node = best;
            }
        }
    }
    if (n > W && !err) err = kStWidth;
    if (jb.count && n != jb.count[qi] && !err) err = kStCount;
    if (err) {
        if (status && lane == 0) *status = err;
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // std::sort: the introsort partition phase is sequential (lane 0; nothing to do up to 16 hits), the final
    // insertion sort is a stable sort and is done by all lanes
    if (lane == 0 && n > 16) std_sort_partition_phase(mine, n);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    stable_sort_wave(mine, sorted, n, lane);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    long long* row = reinterpret_cast<long long*>(jb.idx) + (long)qi * jb.cols;
    for (int j = lane; j < jb.cols; j += 64)
        row[j] = j < n ? (long long)((int)(unsigned)(sorted[j] & 0xFFFFFFFFull) - seg) : (long long)pad;
}

}  // namespace
}  // namespace pcrcg

using namespace pcrcg;

extern "C" {

size_t pcrcg_kdforest_ws_bytes(int ns, int nb) { return forest_bytes(ns, nb < 1 ? 1 : nb); }

int pcrcg_kdforest_build(const float* sup, int ns, const int* slen, int nb, void* forest, size_t forest_bytes_,
                         void* stream) {
    return pcrcg::kdforest_build_levels(sup, ns, slen, nb, 0, nullptr, forest, forest_bytes_, as_stream(stream));
}
}

// per_level > 0: nb / per_level levels of per_level clouds, level l's rows from row level_base[l] of `sup`; ns = a bound on
// the rows `sup` spans (what the workspace was sized for)
int pcrcg::kdforest_build_levels(const float* sup, int ns, const int* slen, int nb, int per_level, const int* level_base,
                                 void* forest, size_t forest_bytes_, hipStream_t stream) {
    PCRCG_CHECK_ARG(ns >= 0 && nb >= 1 && slen && forest);
    PCRCG_CHECK_ARG(per_level == 0 || (per_level > 0 && nb % per_level == 0 && nb / per_level <= PCRCG_MAX_LEVELS && level_base));
    KdBases bases;
    bases.per_level = per_level;
    for (int l = 0; l < PCRCG_MAX_LEVELS; ++l) bases.base[l] = per_level > 0 && l < nb / per_level ? level_base[l] : 0;
    PCRCG_CHECK_ARG(ns == 0 || sup);
    hipStream_t st = stream;
    bool ok;
    KdView v = forest_view(forest, forest_bytes_, ns, nb, &ok);
    if (!ok) {
        set_error("pcrcg_kdforest_build: workspace too small (%zu needed, %zu given)", forest_bytes(ns, nb), forest_bytes_);
        return PCRCG_EWORKSPACE;
    }
    int init_blocks = (v.qcap + 255) / 256;
    init_blocks = init_blocks < 1 ? 1 : (init_blocks > 1024 ? 1024 : init_blocks);
    hipLaunchKernelGGL(k_kd_init, dim3(init_blocks), dim3(256), 0, st, slen, ns, nb, v, bases);
    if (ns > 0) {
        // one workgroup per ~1024 points can be busy at the deepest level of big nodes / the LDS subtrees
        int blocks = ns / (2 * kSubMax) + nb;
        if (blocks > kForestBlocks) blocks = kForestBlocks;
        if (debug_opts().kd_blocks > 0) blocks = debug_opts().kd_blocks;      // tuning aid
        const int spin = debug_opts().kd_spin_limit;          // debugging aid
        hipLaunchKernelGGL(k_kd_forest, dim3(blocks), dim3(kBigThreads), 0, st, sup, v, spin > 0 ? spin : kSpinLimitDefault);
    }
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

extern "C" {

int pcrcg_radius_reorder_jobs(const pcrcg_reorder_job* jobs, int njobs, const float* sup, int ns, int nb,
                              const void* forest, int* status, void* stream) {
    PCRCG_CHECK_ARG(njobs >= 0 && njobs <= PCRCG_MAX_REORDER_JOBS && (njobs == 0 || jobs) && nb >= 1 && ns >= 0);
    ReorderJobs pack;
    int total = 0, width = 1;
    pack.njobs = 0;
    for (int i = 0; i < njobs; ++i) {
        const pcrcg_reorder_job& j = jobs[i];
        // a job searches its own forest (j.forest, over j.sup: the pyramid builder keeps one forest per level) or the call's
        const bool own = j.forest != nullptr;
        const int fnb = own ? j.forest_nb : nb, fns = own ? j.forest_ns : ns;
        PCRCG_CHECK_ARG(own || forest);
        PCRCG_CHECK_ARG(j.nq >= 0 && j.nbq >= 1 && j.cloud0 >= 0 && j.cloud0 + j.nbq <= fnb && fns >= 0 && j.qlen && j.idx);
        PCRCG_CHECK_ARG(j.cols >= 1 && j.max_count >= 0 && j.max_count <= kMaxRow && j.radius > 0.0f);
        const int nrows = j.rows ? j.nrows : j.nq;
        PCRCG_CHECK_ARG(nrows >= 0 && nrows <= j.nq && (j.nq == 0 || (j.q && (own ? j.sup : sup))));
        if (nrows == 0) continue;
        bool vok;
        pack.job[pack.njobs] = j;
        pack.view[pack.njobs] = forest_view(const_cast<void*>(own ? j.forest : forest), forest_bytes(fns, fnb), fns, fnb, &vok);
        pack.sup[pack.njobs] = own ? j.sup : sup;
        pack.row_begin[pack.njobs] = total;
        ++pack.njobs;
        total += nrows;
        if (j.max_count > width) width = j.max_count;
    }
    pack.row_begin[pack.njobs] = total;
    if (total == 0) return PCRCG_OK;
    hipStream_t st = as_stream(stream);
    const size_t second = (size_t)width > (size_t)kTravStack * 5 / 2 ? (size_t)width : (size_t)kTravStack * 5 / 2;
    const size_t per_wave = (size_t)width + second + 1;
    // 64 KB of dynamic LDS per workgroup while that holds the widest row; beyond, one wavefront per workgroup with up to
    // 2 * 8192 * 8 B + 8 B = 128 KB (gfx950: 160 KB per CU), which the kernel has to be granted explicitly
    const int waves = width > 2000 ? 1 : (width > 512 ? 2 : kReorderWaves);
    const size_t lds = per_wave * waves * sizeof(u64);
    if (lds > 64 * 1024) PCRCG_GRANT_LDS(k_reorder);
    hipLaunchKernelGGL(k_reorder, dim3((total + waves - 1) / waves), dim3(waves * 64), lds, st, pack, width, status);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

int pcrcg_radius_reorder(const float* q, int nq, const int* qlen, int nbq, const float* sup, int ns, int nb,
                         const void* forest, int cloud0, float radius, const int* rows, int nrows, const int* count,
                         int max_count, int cols, int64_t* idx, int* status, void* stream) {
    pcrcg_reorder_job j;
    j.q = q;
    j.qlen = qlen;
    j.rows = rows;
    j.count = count;
    j.idx = idx;
    j.nq = nq;
    j.nbq = nbq;
    j.cloud0 = cloud0;
    j.nrows = nrows;
    j.max_count = max_count;
    j.cols = cols;
    j.radius = radius;
    j.group = 0;
    j.sup = nullptr;
    j.forest = nullptr;
    j.forest_ns = j.forest_nb = 0;
    return pcrcg_radius_reorder_jobs(&j, 1, sup, ns, nb, forest, status, stream);
}
}
