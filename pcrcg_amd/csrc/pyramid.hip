// pyramid.hip -- host-side pyramid builder: the whole front end of one fragment pair (3 grid subsamplings, 4 cell
// grids, 10 radius searches, the tie-order restore step) enqueued by ONE C-ABI call from plain C arguments.  The
// counterpart of collate_fn_descriptor's pyramid loop (ref:datasets/dataloader.py:230-361) and of
// batch_grid_subsampling_kpconv / batch_neighbors_kpconv (:14-69); same sequence as pcrcg_amd/pyramid.py's
// pyramid_steps (the Python mirror, kept for calibration and for tie_order="reference"), which this replaces on the
// pipeline's hot path: ~85 launches cost one FFI crossing and ~0.3 ms of host time instead of ~1.7 ms of Python.
//
// No device code of its own: it sequences pcrcg_grid_subsample_batch, pcrcg_cellgrid_build, pcrcg_radius_query_ex,
// pcrcg_kdforest_build and pcrcg_radius_reorder_jobs over a caller-provided arena.  The row count of a subsampled
// level sizes the next level's tables, so the call WAITS for the stream once per pooled level (12 bytes come back
// through the caller's pinned scratch) and once for the ten tables' column counts; callers that want the GPU busy
// meanwhile run several pairs on several streams from several host threads (pcrcg_amd/pairstream.py) -- the call
// holds no lock and no global state.
//
// Arena layout: persistent results first (points of all levels contiguous -- which is also the KD-forest's input,
// so nothing is concatenated later --, lengths [levels][nb], features, tables, per-table counts / tie rows, cell
// grids), transient subsampling scratch with stack discipline.  Everything the returned pcrcg_batch points to lives
// in the arena (except level 0's points when they are used in place).
#include <sched.h>
#include <time.h>

#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"

namespace pcrcg {
namespace {

// Host round trip, three ways (DebugOpts::pyr_wait): 0 = async copy + hipStreamSynchronize, 1 = async copy + event (default:
// waits for the caller's own work only, so several host threads can share one stream),
// 2 = a one-wavefront kernel stores the words straight into the caller's pinned scratch (system-scope release of a
// sequence tag last) and the host polls the tag -- no runtime call, no runtime lock held while waiting.
__global__ void k_post(int* __restrict__ h_dst, const int* __restrict__ src, int n, const int* __restrict__ src2, int n2,
                       int tag) {
    for (int i = threadIdx.x; i < n; i += 64) __hip_atomic_store(h_dst + 1 + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int i = threadIdx.x; i < n2; i += 64)
        __hip_atomic_store(h_dst + 1 + n + i, src2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(h_dst, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// DebugOpts::pyr_trace = 1: host microseconds spent enqueueing vs waiting, per call, printed at exit (tuning aid)
struct Trace {
    bool on = debug_opts().pyr_trace != 0;
    double enq = 0, wait = 0;
    long calls = 0, waits = 0;
    ~Trace() {
        if (on && calls)
            fprintf(stderr, "pcrcg_pyramid_build: %ld calls, per call: enqueue %.1f us, waiting %.1f us in %.1f round trips\n",
                    calls, enq / calls, wait / calls, (double)waits / calls);
    }
};
Trace g_trace;
inline double now_us() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

// ---- which streams share a hardware dispatcher (pcrcg_stream_pipe_classes) ----
// The command processor of gfx950 runs FOUR compute dispatchers ("pipes"); every HIP stream lives on a hardware queue of
// one of them, and a dispatcher hands out the workgroups of ONE kernel at a time: while a kernel of many workgroups is
// being dispatched -- for a GEMM beside other work that is most of its run time -- every other stream on the same dispatcher
// stands still (profiles/r06_queue_pipes.txt: a one-workgroup kernel completes in 12 us beside such a dispatch on another
// dispatcher and in ~470 us behind it on the same one; streams i and i + 4 share one, whatever GPU_MAX_HW_QUEUES says).
// That is the pair engine's "concurrency wall" of rounds 3-5: a fifth busy stream does not add concurrency, it halves two.
__global__ void k_probe_many(int* sink, int spin) {
    int x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1664525 + 1013904223;
    if (x == 0x7fffffff) sink[0] = x;
}
__global__ void k_probe_tiny(int* sink) {
    if (threadIdx.x == 1234567) sink[1] = 1;
}

int wait_mode() {
    return debug_opts().pyr_wait;
}

// words src[0..n) (+ src2[0..n2)) -> h_scratch[1..]; returns when they are there
// (`enqueue`: the stream's enqueue lock held by the caller, released once the transfer is in the stream, before the wait)
int fetch_(int* h_scratch, const int* src, int n, const int* src2, int n2, hipStream_t st, std::unique_lock<std::mutex>* enqueue);
int fetch(int* h_scratch, const int* src, int n, const int* src2, int n2, hipStream_t st,
          std::unique_lock<std::mutex>* enqueue = nullptr) {
    if (!g_trace.on) return fetch_(h_scratch, src, n, src2, n2, st, enqueue);
    const double t0 = now_us();
    const int rc = fetch_(h_scratch, src, n, src2, n2, st, enqueue);
    g_trace.wait += now_us() - t0;      // (racy across threads: a tuning aid)
    g_trace.waits += 1;
    return rc;
}
int fetch_(int* h_scratch, const int* src, int n, const int* src2, int n2, hipStream_t st, std::unique_lock<std::mutex>* enqueue) {
    const int mode = wait_mode();
    auto let_go = [&] { if (enqueue && enqueue->owns_lock()) enqueue->unlock(); };
    if (mode == 2) {
        volatile int* flag = h_scratch;
        const int tag = (*flag & 0x7fffffff) + 1;
        hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, st, h_scratch, src, n, src2, n2, tag);
        PCRCG_CHECK_LAUNCH();
        let_go();
        long spins = 0;
        while (*flag != tag) {
            if (++spins > 64) sched_yield();
            if (spins > 200000000L) { set_error("pcrcg_pyramid_build: device never posted its counts"); return PCRCG_ELAUNCH; }
        }
        __sync_synchronize();
        return PCRCG_OK;
    }
    PCRCG_CHECK_HIP(hipMemcpyAsync(h_scratch + 1, src, sizeof(int) * n, hipMemcpyDeviceToHost, st));
    if (n2 > 0) PCRCG_CHECK_HIP(hipMemcpyAsync(h_scratch + 1 + n, src2, sizeof(int) * n2, hipMemcpyDeviceToHost, st));
    if (mode == 1) {
        hipEvent_t ev;
        PCRCG_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, st);
        let_go();
        if (e == hipSuccess) e = hipEventSynchronize(ev);
        (void)hipEventDestroy(ev);
        PCRCG_CHECK_HIP(e);
    } else {
        let_go();
        PCRCG_CHECK_HIP(hipStreamSynchronize(st));      // (waits for everything in the stream: a later call's chain too)
    }
    return PCRCG_OK;
}

struct Arena {
    char* base;
    size_t cap, off = 0, peak = 0;
    bool dry;
    Arena(void* p, size_t bytes, bool dry_) : base(static_cast<char*>(p)), cap(bytes), dry(dry_) {}
    void* raw(size_t bytes) {
        bytes = (bytes + 255) & ~size_t(255);
        void* p = base ? base + off : nullptr;
        off += bytes;
        if (off > peak) peak = off;
        return p;
    }
    template <typename T>
    T* take(size_t count) { return static_cast<T*>(raw((count ? count : 1) * sizeof(T))); }
    bool ok() const { return dry || off <= cap; }
};

struct TableRec {
    int kind;              // 0 conv (neighbors), 1 pool, 2 upsample
    int level;             // level of the table in the batch
    int q_level;           // level whose points are the queries
    int64_t* idx;
    int* counts;           // [nq] untruncated list lengths (tie restore cross-check), or null
    int* ties;             // [nq] rows holding a tie, or null
    int* meta;             // device [P + 2]: max_count per group, status, tie_rows
    const float* q;
    const int* qlen;
    int nq, limit, sup_level;
    float radius;
    const void* grid;      // cell grid of the supports, their count and cloud lengths (for the redo pass)
    const void* qgrid;     // a cell grid over the QUERIES (any cell size), or null: the cell-cooperative search walks it
    int ns;
    const int* slen;
};

}  // namespace
}  // namespace pcrcg

using namespace pcrcg;

// events of one call (stream fork / join); destroyed when the call returns -- a recorded wait keeps what it needs
struct EventBox {
    std::vector<hipEvent_t> ev;
    int make(hipEvent_t* e) {
        PCRCG_CHECK_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        ev.push_back(*e);
        return PCRCG_OK;
    }
    ~EventBox() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
};

// The input clouds may come in several PARTS (the pairs of a grouped build, each where its caller left it): part i holds
// n_parts[i] rows and nb_parts[i] cloud lengths; they are copied behind each other into the arena (the builder copies its
// input anyway), so stacking pairs costs no concatenation kernel on the caller's side.
struct Parts {
    const float* const* pts;
    const int* n;
    const int* const* len;
    const int* nb;
    int count;
};

// Row bound of every level: level 0 holds n0 rows, a subsampled level at most `shrink` times the bound of the level it
// comes from (the caller's promise, pcrcg_pyramid_cfg::shrink; a cloud that keeps more is reported as PCRCG_EWORKSPACE).
// Everything on the device is SIZED by these bounds -- buffers, tables, launch grids -- and every kernel reads the
// actual row counts from the cloud lengths, which never leave the device until the end of the chain.
static void level_caps(int n0, const pcrcg_pyramid_cfg* cfg, int* cap) {
    double shrink = cfg->shrink;
    if (!(shrink > 0.0) || shrink > 1.0) shrink = 1.0;
    cap[0] = n0;
    for (int l = 1; l < cfg->n_levels; ++l) {
        const double c = ceil((double)cap[l - 1] * shrink);
        cap[l] = c < 1.0 ? 1 : (c > (double)cap[l - 1] ? cap[l - 1] : (int)c);
    }
}

// Builders on several host threads may share ONE stream (the pair engine's pipelined front end): a call holds the stream's
// enqueue lock while it enqueues its chain and lets go of it before it waits for its round trip, so the chains stay whole
// (one after the other in the stream, never interleaved) while one call's wait overlaps the next call's enqueue.
struct EnqueueLocks {
    std::mutex mu;
    hipStream_t key[16];
    std::mutex lock[16];
    int n = 0;
    std::mutex* of(hipStream_t st) {
        std::lock_guard<std::mutex> g(mu);
        for (int i = 0; i < n; ++i)
            if (key[i] == st) return &lock[i];
        if (n == 16) return &lock[15];
        key[n] = st;
        return &lock[n++];
    }
};
static EnqueueLocks g_enqueue;

static int pyramid_run(const Parts& in, int n0, int nb, const pcrcg_pyramid_cfg* cfg, Arena& A,
                       int* h_scratch, pcrcg_batch* out, int* h_lengths, int* h_status, pcrcg_pyramid_restore* deferred,
                       hipStream_t st, std::unique_lock<std::mutex>* enqueue = nullptr) {
    const int L = cfg->n_levels;
    const bool dry = A.dry;
    const bool want_ties = cfg->tie_order != 0;
    const int group = cfg->group > 0 ? cfg->group : 0;
    const int P = group > 0 ? nb / group : 1;          // output batches
    const int MS = P + 2;                              // ints of table metadata
    int cap[PCRCG_MAX_LEVELS];
    level_caps(n0, cfg, cap);
    // ---- persistent block 1: points of all levels (level l at its own bound-sized place), lengths, features, metadata ----
    // (one block, every level a whole number of ROWS behind the first: the upper levels' forest indexes them as one array)
    float* level_pts[PCRCG_MAX_LEVELS];
    {
        size_t rows = 0;
        for (int l = 0; l < L; ++l) rows += (size_t)cap[l] + 1;
        float* pts_all = A.take<float>(3 * rows);
        rows = 0;
        for (int l = 0; l < L; ++l) { level_pts[l] = pts_all + 3 * rows; rows += (size_t)cap[l] + 1; }
    }
    int* lens_all = A.take<int>((size_t)L * nb);
    float* feats = A.take<float>((size_t)n0);
    const int max_tables = 3 * L;
    // [MS per table] + subsampled row counts [L] + "a level outgrew its bound" [1] + the restore step's status word [1]
    const size_t meta_ints = (size_t)MS * max_tables + L + 2;
    int* metas = A.take<int>(meta_ints);
    int* m_dev = metas + MS * max_tables;
    int* overflow = m_dev + L;
    int* tie_status = overflow + 1;
    if (!A.ok()) return PCRCG_EWORKSPACE;

    const bool eager = debug_opts().radius_eager_redo != 0;   // A/B aid
    std::vector<TableRec> tables;
    tables.reserve(max_tables);
    if (!dry) {
        PCRCG_CHECK_HIP(hipMemsetAsync(metas, 0, sizeof(int) * meta_ints, st));
        size_t row = 0, cloud = 0;
        for (int i = 0; i < in.count; ++i) {
            if (in.n[i] > 0)
                PCRCG_CHECK_HIP(hipMemcpyAsync(level_pts[0] + 3 * row, in.pts[i], sizeof(float) * 3 * (size_t)in.n[i], hipMemcpyDeviceToDevice, st));
            PCRCG_CHECK_HIP(hipMemcpyAsync(lens_all + cloud, in.len[i], sizeof(int) * in.nb[i], hipMemcpyDeviceToDevice, st));
            row += (size_t)in.n[i];
            cloud += (size_t)in.nb[i];
        }
        PCRCG_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(feats), 0x3f800000, (size_t)(n0 > 0 ? n0 : 1), st));
    }

    // ---- streams.  The chain is a DAG, not a line: a level's SUBSAMPLING needs only the level's points; the level's cell
    // grid and its conv search need them too and nothing from the subsampling; only the pool / upsample searches and the
    // next level need both.  And a KD-forest (restore step) needs only its levels' points.  With side streams from the
    // caller (pcrcg_pyramid_cfg::side_stream / side_stream2) the three subsamplings run back to back on the first from the
    // moment the input is in place -- beside grid 0 and the conv search of level 0, the longest kernel of the chain -- and
    // the forests on the second (or behind the subsamplings on the first); `st` keeps the grids and the ten searches and
    // waits, level by level, for the subsampled points.  Without side streams everything is enqueued on `st` in an order
    // that is valid for one stream (what rounds 2-5 did).
    // Two forests: level 0's, built as soon as the input is in place, and one over the subsampled levels, built when the
    // last of them exists.  Level 0 holds three quarters of the rows and the deepest trees.  (One forest PER level, one
    // launch each, was measured first: four persistent launches one after the other cost 3.9 ms of forest time per
    // four-pair chain against 1.1 for one launch -- the levels' critical paths add up instead of overlapping.)
    // Tie rows are the rule (every two-point cell of a subsampling puts one into the pool table), so the builds do not
    // wait to be told that there are some.
    hipStream_t sub_st = cfg->side_stream ? as_stream(cfg->side_stream) : st;
    hipStream_t f_st = cfg->side_stream2 ? as_stream(cfg->side_stream2) : sub_st;
    EventBox events;
    // b waits for everything enqueued on a so far
    auto order = [&](hipStream_t a, hipStream_t b) -> int {
        if (a == b || dry) return PCRCG_OK;
        hipEvent_t ev;
        PCRCG_PROPAGATE(events.make(&ev));
        PCRCG_CHECK_HIP(hipEventRecord(ev, a));
        PCRCG_CHECK_HIP(hipStreamWaitEvent(b, ev, 0));
        return PCRCG_OK;
    };
    // In line (no stream of their own) ONE forest over all levels, built when the last level exists: the levels' trees then
    // grow side by side in one persistent launch (1.07 ms per four-pair chain in the engine against 2 x 0.6 for two launches).
    const bool one_forest = !want_ties || f_st == st || L < 2;
    void* forest[2] = {nullptr, nullptr};              // [0] level 0 (one_forest: all levels), [1] levels 1 .. L-1
    size_t forest_b[2] = {0, 0};
    int forest_ns[2] = {0, 0}, forest_nb[2] = {0, 0};
    int level_base[PCRCG_MAX_LEVELS] = {};               // row of level l relative to the first level of its forest
    auto build_forest = [&](int which) -> int {        // 0: level 0, 1: levels 1 .. L-1, 2: all levels (-> forest[0])
        if (!want_ties || (which == 1 && L < 2)) return PCRCG_OK;
        const int first = which == 1 ? 1 : 0, last = which == 0 ? 0 : L - 1, slot = which == 1 ? 1 : 0;
        for (int l = first; l <= last; ++l) level_base[l - first] = (int)((level_pts[l] - level_pts[first]) / 3);
        forest_ns[slot] = level_base[last - first] + cap[last];
        forest_nb[slot] = (last - first + 1) * nb;
        forest_b[slot] = pcrcg_kdforest_ws_bytes(forest_ns[slot], forest_nb[slot]);
        forest[slot] = A.raw(forest_b[slot]);
        if (!A.ok()) return PCRCG_EWORKSPACE;
        if (dry) return PCRCG_OK;
        PCRCG_PROPAGATE(order(which == 0 ? st : sub_st, f_st));       // the input copies / the last subsampling
        return kdforest_build_levels(level_pts[first], forest_ns[slot], lens_all + (size_t)first * nb, forest_nb[slot],
                                     last > first ? nb : 0, level_base, forest[slot], forest_b[slot], f_st);
    };

    const bool use_cells = debug_opts().radius_cells != 0;
    // nq / ns are the BOUNDS of the query / support level (grids and tables are carved for them)
    auto add_table = [&](int kind, int level, int q_level, const void* grid, const void* qgrid, float radius, const float* q,
                         const int* qlen, int nq, int ns, const int* slen, int limit, int sup_level) -> int {
        TableRec t;
        t.kind = kind; t.level = level; t.q_level = q_level;
        t.idx = A.take<int64_t>((size_t)nq * limit);
        t.counts = want_ties ? A.take<int>((size_t)nq) : nullptr;
        t.ties = want_ties ? A.take<int>((size_t)nq) : nullptr;
        t.meta = metas + MS * tables.size();
        t.q = q; t.qlen = qlen; t.nq = nq; t.limit = limit; t.sup_level = sup_level; t.radius = radius;
        t.grid = grid; t.ns = ns; t.slen = slen; t.qgrid = use_cells ? qgrid : nullptr;
        if (!A.ok()) return PCRCG_EWORKSPACE;
        // first pass only: rows with more than 128 hits (kCellListCap; 256 = kListCapFast for the per-query kernel) (and, from the cell-cooperative search, rows of a cell whose
        // neighbourhood does not fit LDS) are marked and announced in the metadata; whether any table has one is known
        // with the metadata round trip below, and only then (normally never) the redo pass runs
        if (!dry && t.qgrid)
            PCRCG_PROPAGATE(radius_cells_pass(t.qgrid, q, nq, qlen, grid, ns, slen, nb, group, radius, limit, t.idx, t.counts,
                                              t.meta, t.meta + P, t.ties, want_ties ? t.meta + P + 1 : nullptr, st, eager ? 0 : 1));
        else if (!dry)
            PCRCG_PROPAGATE(radius_query_pass(q, nq, qlen, ns, slen, nb, group, radius, grid, limit, t.idx, t.counts, t.meta,
                                              t.meta + P, t.ties, want_ties ? t.meta + P + 1 : nullptr, st, eager ? 0 : 1));
        tables.push_back(t);
        return PCRCG_OK;
    };
    auto build_grid = [&](const float* sup, int ns, const int* slen, float radius, void** grid) -> int {
        const size_t gb = pcrcg_cellgrid_ws_bytes(ns, nb);
        *grid = A.raw(gb);
        if (!A.ok()) return PCRCG_EWORKSPACE;
        if (!dry) PCRCG_PROPAGATE(pcrcg_cellgrid_build(sup, ns, slen, nb, radius, *grid, gb, st));
        return PCRCG_OK;
    };

    // the subsamplings' scratch: ONE block for all levels (they run one after the other on one stream; level 0 needs the
    // most).  It is NOT handed back to the arena: beside the searches on another stream nothing may share its memory.
    const size_t sub_wsb = L > 1 ? pcrcg_grid_subsample_ws_bytes(cap[0], nb) : 0;
    void* sub_ws = L > 1 ? A.raw(sub_wsb) : nullptr;
    const bool knock_morton = debug_opts().pyr_morton != 0 && L > 1;       // measurement aid (morton_knock.hip)
    const size_t mk_wsb = knock_morton ? morton_knock_ws_bytes(cap[1]) : 0;
    void* mk_ws = knock_morton ? A.raw(mk_wsb) : nullptr;
    if (!A.ok()) return PCRCG_EWORKSPACE;

    void* carried = nullptr;
    float carried_r = 0.f;
    PCRCG_PROPAGATE(order(st, sub_st));                  // the input is in place: the subsamplings may start
    if (!one_forest) PCRCG_PROPAGATE(build_forest(0));
    else if (L == 1) PCRCG_PROPAGATE(build_forest(0));
    for (int l = 0; l < L; ++l) {
        const int limit = cfg->limit[l];
        const float r_conv = cfg->r_conv[l], r_pool = cfg->r_pool[l];
        float* pts = level_pts[l];
        int* lens = lens_all + (size_t)l * nb;
        const int n = cap[l];
        void* grid = nullptr;
        float grid_r = 0.f;
        const bool pooled = cfg->pooled[l] && l + 1 < L;
        if (!pooled && l + 1 < L) {
            set_error("pcrcg_pyramid_build: level %d of %d is not pooled", l, L);
            return PCRCG_EBADARG;
        }
        // the level's subsampling first: on its own stream it runs beside the grid and the conv search below
        if (pooled && !dry)
            PCRCG_PROPAGATE(grid_subsample_bound(pts, n, lens, nb, cfg->dl[l], 0, level_pts[l + 1], lens + nb, m_dev + l, cap[l + 1],
                                                 overflow, sub_ws, sub_wsb, sub_st));
        if (pooled && !dry && knock_morton)
            PCRCG_PROPAGATE(morton_knock_level(level_pts[l + 1], cap[l + 1], lens + nb, nb, mk_ws, mk_wsb, sub_st));
        if (pooled && l + 2 == L) PCRCG_PROPAGATE(build_forest(one_forest ? 2 : 1));      // the last subsampled level exists
        if (cfg->has_conv[l]) {
            if (carried && carried_r == r_conv) { grid = carried; grid_r = carried_r; }
            else { PCRCG_PROPAGATE(build_grid(pts, n, lens, r_conv, &grid)); grid_r = r_conv; }
            PCRCG_PROPAGATE(add_table(0, l, l, grid, grid, r_conv, pts, lens, n, n, lens, limit, l));
        }
        carried = nullptr;
        if (pooled) {
            float* sub = level_pts[l + 1];
            int* sub_len = lens + nb;
            const int m = cap[l + 1];
            if (grid == nullptr || grid_r != r_pool) {
                PCRCG_PROPAGATE(build_grid(pts, n, lens, r_pool, &grid));
                grid_r = r_pool;
            }
            PCRCG_PROPAGATE(order(sub_st, st));          // from here on `st` reads the subsampled level
            // the coarse level's grid first: it is the upsample table's support grid, the next level's conv grid AND the
            // query grid of the pool table (every query set of the pyramid walks a grid of its own, cell by cell)
            void* up_grid = nullptr;
            PCRCG_PROPAGATE(build_grid(sub, m, sub_len, 2 * r_pool, &up_grid));
            PCRCG_PROPAGATE(add_table(1, l, l + 1, grid, up_grid, r_pool, sub, sub_len, m, n, lens, limit, l));
            PCRCG_PROPAGATE(add_table(2, l, l, up_grid, grid, 2 * r_pool, pts, lens, n, m, sub_len, cfg->up_nearest ? 1 : limit, l + 1));
            carried = up_grid;
            carried_r = 2 * r_pool;
        }
    }
    if (dry) return PCRCG_OK;
    hipEvent_t forests_done = nullptr;
    if (want_ties && f_st != st) {
        PCRCG_PROPAGATE(events.make(&forests_done));
        PCRCG_CHECK_HIP(hipEventRecord(forests_done, f_st));
    }

    // ---- the ONE round trip of the call: column counts, capacity status, rows holding ties, row counts, cloud lengths ----
    const int nt = (int)tables.size();
    const int meta_words = MS * max_tables + L + 1;
    PCRCG_PROPAGATE(fetch(h_scratch, metas, meta_words, lens_all, L * nb, st, enqueue));
    // whatever follows on `st` (the reorder step, the caller's readers, a second attempt in the same arena) comes after the forests
    if (forests_done) PCRCG_CHECK_HIP(hipStreamWaitEvent(st, forests_done, 0));
    const int* hm = h_scratch + 1;
    if (hm[MS * max_tables + L] != 0) {
        set_error("pcrcg_pyramid_build: a level keeps more rows than pcrcg_pyramid_cfg::shrink = %.3f allows -- call again with a "
                  "larger bound (1.0 always fits)", cfg->shrink);
        return PCRCG_EWORKSPACE;
    }
    for (int i = 0; i < L * nb; ++i) h_lengths[i] = hm[meta_words + i];
    int level_n[PCRCG_MAX_LEVELS];
    for (int l = 0; l < L; ++l) {
        level_n[l] = 0;
        for (int c = 0; c < nb; ++c) level_n[l] += h_lengths[(size_t)l * nb + c];
    }
    {   // tables with a row of more hits than the first pass stages (128 in the cell search, 256 in the per-query kernel): redo pass now, then the metadata once more (it appends tie rows)
        int redone = 0;
        for (int i = 0; i < nt; ++i) {
            int widest = 0;
            for (int p = 0; p < P; ++p) widest = hm[MS * i + p] > widest ? hm[MS * i + p] : widest;
            const bool handed_over = (hm[MS * i + P] & kRadiusRedoStatus) != 0;     // (the redo pass clears the bit)
            if ((widest <= radius_fast_cap() && !handed_over) || eager) continue;
            const TableRec& t = tables[i];
            PCRCG_PROPAGATE(radius_query_pass(t.q, t.nq, t.qlen, t.ns, t.slen, nb, group, t.radius, t.grid, t.limit, t.idx,
                                              t.counts, t.meta, t.meta + P, t.ties, want_ties ? t.meta + P + 1 : nullptr, st,
                                              2));
            ++redone;
        }
        if (redone) PCRCG_PROPAGATE(fetch(h_scratch, metas, meta_words, lens_all, L * nb, st));
    }
    // rows of group p at level l: [row0, row0 + rows)
    auto group_rows = [&](int l, int p, int* row0, int* rows) {
        const int c0 = group > 0 ? p * group : 0, c1 = group > 0 ? c0 + group : nb;
        int a = 0, r = 0;
        for (int c = 0; c < c1; ++c) (c < c0 ? a : r) += h_lengths[(size_t)l * nb + c];
        *row0 = a;
        *rows = r;
    };
    for (int p = 0; p < P; ++p) {
        pcrcg_batch& o = out[p];
        o.n_levels = L;
        int row0, rows;
        group_rows(0, p, &row0, &rows);
        o.features = feats + row0;
        o.feat_dim = 1;
        o.len_src_c = h_lengths[(size_t)(L - 1) * nb + (group > 0 ? p * group : 0)];
        for (int l = 0; l < L; ++l) {
            group_rows(l, p, &row0, &rows);
            o.points[l] = level_pts[l] + 3 * (size_t)row0;
            o.n_points[l] = rows;
            o.stack_lengths[l] = lens_all + (size_t)l * nb + (group > 0 ? p * group : 0);
            o.neighbors[l] = pcrcg_table{nullptr, rows, 0, 1};
            o.pools[l] = pcrcg_table{nullptr, 0, 0, 1};
            o.upsamples[l] = pcrcg_table{nullptr, 0, 0, 1};
        }
    }
    pcrcg_pyramid_restore local;
    pcrcg_pyramid_restore& R = deferred ? *deferred : local;
    R.njobs = 0;
    for (int i = 0; i < nt; ++i) {
        TableRec& t = tables[i];
        const int status = hm[MS * i + P], tie_rows = hm[MS * i + P + 1];
        if (status != 0) {
            set_error("pcrcg_pyramid_build: radius search capacity exceeded (table %d)", i);
            return PCRCG_ECAPACITY;
        }
        int widest = 0;
        for (int p = 0; p < P; ++p) {
            const int max_count = hm[MS * i + p];
            widest = max_count > widest ? max_count : widest;
            // neighbors[:, :limit] keeps FEWER columns when the longest list is shorter (ref:datasets/dataloader.py:65-67)
            const int cols = max_count < t.limit ? (max_count > 0 ? max_count : 0) : t.limit;
            int row0, rows;
            group_rows(t.q_level, p, &row0, &rows);
            pcrcg_table tab{t.idx + (size_t)row0 * t.limit, rows, cols, t.limit};
            if (t.kind == 0) out[p].neighbors[t.level] = tab;
            else if (t.kind == 1) out[p].pools[t.level] = tab;
            else out[p].upsamples[t.level] = tab;
        }
        if (want_ties && widest > 0 && tie_rows > 0 && R.njobs < PCRCG_MAX_REORDER_JOBS) {
            pcrcg_reorder_job& j = R.jobs[R.njobs++];
            const int fi = (one_forest || t.sup_level == 0) ? 0 : 1;
            j.q = t.q; j.qlen = t.qlen; j.rows = t.ties; j.count = t.counts; j.idx = t.idx;
            j.nq = level_n[t.q_level]; j.nbq = nb; j.cloud0 = one_forest ? t.sup_level * nb : (fi == 0 ? 0 : (t.sup_level - 1) * nb);
            j.nrows = tie_rows;
            j.max_count = widest < 8192 ? widest : 8192;   // (tieorder.hip stages at most 8192 hits per row)
            j.cols = t.limit; j.radius = t.radius; j.group = group;
            j.sup = level_pts[fi]; j.forest = forest[fi]; j.forest_ns = forest_ns[fi]; j.forest_nb = forest_nb[fi];
        }
    }
    // ---- the reference's order inside groups of exactly equal distance (tieorder.hip) ----------------------
    R.tie_status = tie_status;
    if (!deferred) return pcrcg_pyramid_restore_run(&R, h_status, st);
    return PCRCG_OK;
}

extern "C" {

size_t pcrcg_pyramid_ws_bytes(int n0, int nb, const pcrcg_pyramid_cfg* cfg) {
    if (!cfg || n0 < 0 || nb < 1 || cfg->n_levels < 1 || cfg->n_levels > PCRCG_MAX_LEVELS) return 0;
    Arena A(nullptr, 0, true);
    const Parts none{nullptr, nullptr, nullptr, nullptr, 0};
    if (pyramid_run(none, n0, nb, cfg, A, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != PCRCG_OK) return 0;
    return A.peak + 4096;
}

// A non-blocking HIP stream (hipStreamNonBlocking) with an optional priority (0 = default, -1 = high).
int pcrcg_stream_create(void** stream, int priority) {
    PCRCG_CHECK_ARG(stream != nullptr);
    hipStream_t st = nullptr;
    PCRCG_CHECK_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, priority));
    *stream = st;
    return PCRCG_OK;
}

// cls[i] = dispatcher class of streams[i]: 0 for streams[0]'s, then 1, 2, ... in order of first appearance.  Measured, not
// looked up (the runtime does not say): a dispatch-bound kernel (2 M one-wavefront workgroups, ~0.9 ms of dispatching) on a
// class representative, a one-workgroup kernel on the candidate 150 us later, host-timed; "same dispatcher" = the tiny
// kernel took more than a third of what is left of the big dispatch.  Every candidate is tested against one representative
// per known class (at most four exist), each test twice; ~1.5 ms per test.  Call it on an otherwise idle GPU.
// scratch: >= 16 bytes of device memory.
int pcrcg_stream_pipe_classes(void* const* streams, int n, int* cls, void* scratch) {
    PCRCG_CHECK_ARG(streams && cls && scratch && n >= 1 && n <= 64);
    int* sink = static_cast<int*>(scratch);
    PCRCG_CHECK_HIP(hipDeviceSynchronize());
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_probe_tiny, dim3(1), dim3(64), 0, as_stream(streams[i]), sink);   // warm every queue
    PCRCG_CHECK_HIP(hipDeviceSynchronize());
    const int blocks = 2000000;
    double t0 = now_us();
    hipLaunchKernelGGL(k_probe_many, dim3(blocks), dim3(64), 0, as_stream(streams[0]), sink, 64);
    PCRCG_CHECK_HIP(hipStreamSynchronize(as_stream(streams[0])));
    const double big_us = now_us() - t0;
    if (big_us < 400.0) {      // the probe needs a dispatch that outlasts the launch of the tiny kernel
        set_error("pcrcg_stream_pipe_classes: the dispatch-bound probe kernel took only %.0f us", big_us);
        return PCRCG_ELAUNCH;
    }
    int reps[8], nrep = 0;
    auto shares = [&](int a, int b, bool* out) -> int {      // does streams[b] wait for a dispatch on streams[a]?
        // best of three, and a trial counts only if the host got from the big launch to the tiny one in time (a preempted
        // host thread finds the dispatch over and would report "another dispatcher")
        int yes = 0, no = 0;
        for (int trial = 0; trial < 10 && yes < 2 && no < 2; ++trial) {
            const double launch = now_us();
            hipLaunchKernelGGL(k_probe_many, dim3(blocks), dim3(64), 0, as_stream(streams[a]), sink, 64);
            const double start = now_us();
            while (now_us() - start < 150.0) {}
            const double a0 = now_us();
            hipLaunchKernelGGL(k_probe_tiny, dim3(1), dim3(64), 0, as_stream(streams[b]), sink);
            const double a1 = now_us();
            PCRCG_CHECK_HIP(hipStreamSynchronize(as_stream(streams[b])));
            const double d = now_us() - a0;
            PCRCG_CHECK_HIP(hipDeviceSynchronize());
            if (a1 - launch > 0.6 * big_us) continue;
            (d > (big_us - 150.0) / 3.0 ? yes : no) += 1;
        }
        if (yes < 2 && no < 2) { set_error("pcrcg_stream_pipe_classes: the host is too busy for the probe's timing"); return PCRCG_ELAUNCH; }
        *out = yes == 2;
        return PCRCG_OK;
    };
    for (int i = 0; i < n; ++i) {
        cls[i] = -1;
        for (int j = 0; j < i && cls[i] < 0; ++j)
            if (streams[j] == streams[i]) cls[i] = cls[j];
        for (int r = 0; r < nrep && cls[i] < 0; ++r) {
            bool same = false;
            PCRCG_PROPAGATE(shares(reps[r], i, &same));
            if (same) cls[i] = r;
        }
        if (cls[i] < 0) {
            if (nrep == 8) { set_error("pcrcg_stream_pipe_classes: more than 8 dispatcher classes -- the probe is not measuring what it should"); return PCRCG_ELAUNCH; }
            reps[nrep] = i;
            cls[i] = nrep++;
        }
    }
    return PCRCG_OK;
}

int pcrcg_stream_destroy(void* stream) {
    PCRCG_CHECK_HIP(hipStreamDestroy(as_stream(stream)));
    return PCRCG_OK;
}

int pcrcg_pyramid_restore_run(const pcrcg_pyramid_restore* r, int* h_status, void* stream) {
    PCRCG_CHECK_ARG(r && r->tie_status && r->njobs >= 0 && r->njobs <= PCRCG_MAX_REORDER_JOBS);
    hipStream_t st = as_stream(stream);
    if (r->njobs > 0) PCRCG_PROPAGATE(pcrcg_radius_reorder_jobs(r->jobs, r->njobs, nullptr, 0, 1, nullptr, r->tie_status, st));
    if (h_status) PCRCG_CHECK_HIP(hipMemcpyAsync(h_status, r->tie_status, sizeof(int), hipMemcpyDeviceToHost, st));
    return PCRCG_OK;
}

static int pyramid_build_checked(const Parts& in, int n0, int nb, const pcrcg_pyramid_cfg* cfg, void* ws, size_t ws_bytes,
                                 int* h_scratch, pcrcg_batch* out, int* h_lengths, int* h_status,
                                 pcrcg_pyramid_restore* deferred, void* stream) {
    PCRCG_CHECK_ARG(cfg && ws && h_scratch && out && h_lengths);
    PCRCG_CHECK_ARG(n0 >= 1 && nb >= 1 && nb <= 16);
    PCRCG_CHECK_ARG(cfg->group >= 0 && (cfg->group == 0 || nb % cfg->group == 0));
    PCRCG_CHECK_ARG((cfg->group > 0 ? nb / cfg->group : 1) + 2 <= 16 && cfg->n_levels * nb <= 64);
    PCRCG_CHECK_ARG(cfg->n_levels >= 1 && cfg->n_levels <= PCRCG_MAX_LEVELS && 3 * cfg->n_levels <= PCRCG_MAX_REORDER_JOBS);
    for (int l = 0; l < cfg->n_levels; ++l)
        PCRCG_CHECK_ARG(cfg->limit[l] >= 1 && cfg->r_conv[l] > 0.f && (l + 1 == cfg->n_levels || (cfg->dl[l] > 0.f && cfg->r_pool[l] > 0.f)));
    Arena A(ws, ws_bytes, false);
    hipStream_t st = as_stream(stream);
    const double t0 = g_trace.on ? now_us() : 0.0, w0 = g_trace.wait;
    std::unique_lock<std::mutex> enqueue(*g_enqueue.of(st));
    const int rc = pyramid_run(in, n0, nb, cfg, A, h_scratch, out, h_lengths, h_status, deferred, st, &enqueue);
    if (g_trace.on) { g_trace.enq += now_us() - t0 - (g_trace.wait - w0); g_trace.calls += 1; }
    if (rc == PCRCG_EWORKSPACE)
        if (A.off > ws_bytes) set_error("pcrcg_pyramid_build: arena too small (%zu needed so far, %zu given): size it with pcrcg_pyramid_ws_bytes for this cfg", A.off, ws_bytes);
    return rc;
}

int pcrcg_pyramid_build(const float* pts, int n0, const int* len, int nb, const pcrcg_pyramid_cfg* cfg, void* ws,
                        size_t ws_bytes, int* h_scratch, pcrcg_batch* out, int* h_lengths, int* h_status,
                        pcrcg_pyramid_restore* deferred, void* stream) {
    PCRCG_CHECK_ARG(pts && len);
    const Parts in{&pts, &n0, &len, &nb, 1};
    return pyramid_build_checked(in, n0, nb, cfg, ws, ws_bytes, h_scratch, out, h_lengths, h_status, deferred, stream);
}

int pcrcg_pyramid_build_parts(const float* const* pts_parts, const int* n_parts, const int* const* len_parts,
                              const int* nb_parts, int parts, const pcrcg_pyramid_cfg* cfg, void* ws, size_t ws_bytes,
                              int* h_scratch, pcrcg_batch* out, int* h_lengths, int* h_status,
                              pcrcg_pyramid_restore* deferred, void* stream) {
    PCRCG_CHECK_ARG(pts_parts && n_parts && len_parts && nb_parts && parts >= 1 && parts <= 8);
    long n0 = 0, nb = 0;
    for (int i = 0; i < parts; ++i) {
        PCRCG_CHECK_ARG(n_parts[i] >= 0 && nb_parts[i] >= 1 && len_parts[i] && (n_parts[i] == 0 || pts_parts[i]));
        n0 += n_parts[i];
        nb += nb_parts[i];
    }
    PCRCG_CHECK_ARG(n0 <= 0x7fffffffL);
    const Parts in{pts_parts, n_parts, len_parts, nb_parts, parts};
    return pyramid_build_checked(in, (int)n0, (int)nb, cfg, ws, ws_bytes, h_scratch, out, h_lengths, h_status, deferred, stream);
}
}
