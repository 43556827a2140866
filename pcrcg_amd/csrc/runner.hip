// runner.hip -- host-side network runner: enqueues the whole KPFCNN forward
// (ref:models/architectures.py:181-191, 516-610; blocks ref:models/blocks.py:536-709; GNN
// ref:models/gcn.py:96-217) from ONE C-ABI call.  It contains no device code of its own: it sequences
// the kernels of this library (pcrcg_* entry points) over a caller-provided arena, exactly like the
// Python mirror in pcrcg_amd/{blocks,gcn,architectures}.py does op by op -- the two are compared in
// tests/test_model_gpu.py.  The point is host cost: several hundred kernel launches per fragment pair
// cost one FFI crossing instead of several hundred.
//
// Memory: block outputs live until the end of the forward (skip connections need some of them and the
// total is ~0.25 GB for a 2 x 30k-point pair); temporaries of a block are released when it returns
// (stack discipline).  pcrcg_kpfcnn_ws_bytes runs the same code with launches disabled to size it.
#include <vector>

#include <cstdlib>

#include "common.h"

namespace pcrcg {
// gemm.hip
bool gemm_bt_accumulates(int m, int n, int k, long m_total = 0);
int gemm_bt_colstats(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k,
                     const float* row_scale, const float* bias, void* colstats, size_t colstats_bytes, int* h_chunks,
                     hipStream_t st, bool c_zeroed, bool colstats_sums = false);
int gemm_bf16a_bt_colstats(const void* a_bf16, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k,
                           const float* row_scale, const float* bias, void* colstats, size_t colstats_bytes,
                           int* h_chunks, hipStream_t st, bool c_zeroed, bool colstats_sums);
bool gemm_colstats_sums_ok();
bool gemm_extra_ok();
int gemm_bt_extra(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k, hipStream_t st,
                  bool c_zeroed, const GemmExtra& ex, const float* bias = nullptr, void* colstats = nullptr,
                  size_t colstats_bytes = 0, int* h_chunks = nullptr, bool colstats_sums = false);
bool gemm_pair_ok();
int gemm_bt_colstats_pair(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k,
                          const float* row_scale, const float* bias, void* colstats, size_t colstats_bytes, int* h_chunks,
                          hipStream_t st, bool c_zeroed, bool colstats_sums, const GemmGroup* grp);
int gemm_bt_extra_pair(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k, hipStream_t st,
                       bool c_zeroed, const GemmExtra& ex, const float* bias, void* colstats, size_t colstats_bytes,
                       int* h_chunks, bool colstats_sums, const GemmGroup* grp);
namespace {

constexpr int GMAX = 4;   // fragment pairs one call can carry (pcrcg_kpfcnn_forward_group)

// Row-major fp32 matrix view -- one per fragment pair of the call (same width and leading dimension, own rows).  With
// several pairs every operator below runs its per-pair kernels once per pair and its weight products ONCE for all pairs (GemmGroup):
// the pairs never mix (InstanceNorm statistics, neighbour tables, kNN and attention stay per pair -- SURVEY.md 8e), but a
// coarse-level product that fills a fifth of the chip for one pair fills twice that for two at the same duration, and
// every product costs one launch per two pairs.
struct Mat {
    float* p[GMAX] = {};
    int rows[GMAX] = {};
    int cols = 0, ld = 0;
    bool zeroed = false;   // lives in the zero arena and has not been written yet
};

struct Ctx {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    hipStream_t st = nullptr;
    bool dry = false;
    bool bf16 = false;      // pcrcg_model.feature_bf16
    int G = 1;              // pairs of this call
    int rc = PCRCG_OK;

    void* raw(size_t bytes) {
        bytes = (bytes + 255) & ~size_t(255);
        void* p = base ? base + off : nullptr;
        off += bytes;
        if (off > peak) peak = off;
        if (!dry && off > cap && rc == PCRCG_OK) {
            set_error("pcrcg_kpfcnn_forward: workspace too small (%zu needed so far, %zu given)", off, cap);
            rc = PCRCG_EWORKSPACE;
        }
        return p;
    }
    Mat mat(const int* rows, int cols, int ld = 0) {
        Mat m;
        m.cols = cols; m.ld = ld ? ld : cols;
        for (int g = 0; g < G; ++g) {
            m.rows[g] = rows[g];
            m.p[g] = static_cast<float*>(raw((size_t)(rows[g] > 0 ? rows[g] : 1) * m.ld * sizeof(float)));
        }
        return m;
    }
    // Zero arena: the front part of the workspace, cleared by ONE memset at the start of the forward and handed out
    // without reuse.  Outputs of split-K products come from it (the product accumulates into C with atomics and would
    // otherwise clear C itself: 39 memset launches per S30k forward).
    char* zbase = nullptr;
    size_t zcap = 0, zoff = 0;
    void* zraw(size_t bytes) {
        bytes = (bytes + 255) & ~size_t(255);
        void* p = zbase ? zbase + zoff : nullptr;
        zoff += bytes;
        if (!dry && zoff > zcap && rc == PCRCG_OK) {
            set_error("pcrcg_kpfcnn_forward: zero arena too small (%zu needed so far, %zu reserved)", zoff, zcap);
            rc = PCRCG_EWORKSPACE;
        }
        return p;
    }
    int max_rows(const int* rows) const { int m = rows[0]; for (int g = 1; g < G; ++g) m = rows[g] > m ? rows[g] : m; return m; }
    // the output of a [rows, k] x [cols, k]^T product (the plan of the pair with most rows decides for all)
    Mat gemm_out(const int* rows, int cols, int k) {
        long total = 0;
        for (int g = 0; g < G; ++g) total += rows[g];
        if (!debug_opts().zero_arena || !gemm_bt_accumulates(max_rows(rows), cols, k, total)) return mat(rows, cols);
        Mat m;
        m.cols = cols; m.ld = cols;
        for (int g = 0; g < G; ++g) {
            m.rows[g] = rows[g];
            m.p[g] = static_cast<float*>(zraw((size_t)(rows[g] > 0 ? rows[g] : 1) * cols * sizeof(float)));
        }
        m.zeroed = true;
        return m;
    }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
    bool live() const { return !dry && rc == PCRCG_OK; }
    void check(int r) { if (r != PCRCG_OK && rc == PCRCG_OK) rc = r; }
    // several pairs in one launch: only with the split-bf16 arithmetic (PCRCG_GEMM_MODE=0 runs the products pair by pair)
    bool paired() const { return G >= 2 && gemm_pair_ok(); }
    bool paired_ok() const { return gemm_pair_ok(); }
};

inline int pad4(int v) { return (v + 3) & ~3; }
inline Mat cols(const Mat& m, int c0, int n) {
    Mat r = m;
    for (int g = 0; g < GMAX; ++g) r.p[g] = m.p[g] ? m.p[g] + c0 : nullptr;
    r.cols = n;
    return r;
}
// rows [r0[g], r0[g] + n[g]) of every pair's matrix
inline Mat rows(const Mat& m, const int* r0, const int* n) {
    Mat r = m;
    for (int g = 0; g < GMAX; ++g) {
        r.p[g] = m.p[g] ? m.p[g] + (long)r0[g] * m.ld : nullptr;
        r.rows[g] = n[g];
    }
    return r;
}

// A GEMM output together with the InstanceNorm column partials its epilogue may have produced (per pair).
struct Stat {
    void* partials[GMAX] = {};                // [2][cols][chunks] fp64 partials, valid when chunks > 0; or, when `sums`,
                                                 // zeroed [2][cols] fp64 accumulators that hold the column sums when chunks == -1
    size_t bytes = 0;
    int chunks[GMAX] = {};
    bool sums = false;
};

// The GEMM epilogue adds its output's column sums into accumulators from the zero arena with fp64 atomics -- one pair per
// column and 64-row tile, the workgroup's two row halves meet in LDS first -- and the normalisation derives mean / rstd
// from them itself: no partial buffers, no finishing launch (51 per S30k forward before).  With the per-wavefront
// atomics of the first version the 60 000-row outputs (1 876 per address) measured slower than partials + finishing
// kernel and kept those; with one per tile (938) the sums win everywhere: 470-473 vs 457-466 pairs/s.
// (DebugOpts::stat_sums_rows: outputs above that many rows keep the deterministic partials.)
Stat stat_buffer(Ctx& c, const int* rows, int cols) {
    Stat s;
    const int mr = c.max_rows(rows);
    if (debug_opts().stat_sums && mr <= debug_opts().stat_sums_rows && gemm_colstats_sums_ok()) {
        s.sums = true;
        s.bytes = 2 * sizeof(double) * (size_t)cols;
        for (int g = 0; g < c.G; ++g) s.partials[g] = c.zraw(s.bytes);
        return s;
    }
    s.bytes = pcrcg_gemm_colstats_bytes(mr, cols);
    for (int g = 0; g < c.G; ++g) s.partials[g] = c.raw(s.bytes);
    return s;
}

// widths the sums form of the normalisation kernel serves (its thread -> channel-group map)
inline bool sums_apply_ok(const Mat& x, const Mat& y, const Mat* res) {
    const int c4 = x.cols / 4;
    return x.cols % 4 == 0 && x.cols >= 4 && (c4 <= 256 ? 256 % c4 == 0 : c4 % 256 == 0) && x.ld % 4 == 0 && y.ld % 4 == 0 &&
           (!res || res->ld % 4 == 0);
}

// (mean, rstd) of pair g's x: from the producing GEMM's partials when it left some, else by a pass over x
void col_stats(Ctx& c, const Mat& x, int g, const Stat* s, float* stats, void* ws, size_t wsb) {
    if (s && s->chunks[g] == -1)      // column sums: the finishing kernel reads them as one chunk per column
        c.check(pcrcg_instnorm_stats_from_partials(s->partials[g], 1, x.cols, (double)x.rows[g], 1e-5f, stats, c.st));
    else if (s && s->chunks[g] > 0)
        c.check(pcrcg_instnorm_stats_from_partials(s->partials[g], s->chunks[g], x.cols, (double)x.rows[g], 1e-5f, stats, c.st));
    else
        c.check(pcrcg_instnorm_stats(x.p[g], x.rows[g], x.cols, x.ld, 1e-5f, stats, ws, wsb, c.st));
}

// sums-mode statistics the producing product could not leave (a split-K product): one pass into the same accumulators
void fill_sums(Ctx& c, const Mat& x, Stat* xs) {
    if (!c.live() || !xs || !xs->sums) return;
    {   // the pairs whose product left nothing: one launch for all of them
        const float* xp[GMAX]; double* sp[GMAX]; int np[GMAX]; int idx[GMAX]; int cnt = 0;
        for (int g = 0; g < c.G && cnt < 4; ++g)
            if (xs->chunks[g] == 0 && x.rows[g] >= 1) { xp[cnt] = x.p[g]; sp[cnt] = static_cast<double*>(xs->partials[g]); np[cnt] = x.rows[g]; idx[cnt++] = g; }
        if (cnt >= 2) {
            c.check(instnorm_colsums_multi(xp, sp, np, cnt, x.cols, x.ld, c.st));
            for (int i = 0; i < cnt; ++i) xs->chunks[idx[i]] = -1;
        }
    }
    for (int g = 0; g < c.G; ++g)
        if (xs->chunks[g] == 0) {
            c.check(pcrcg_instnorm_colsums(x.p[g], x.rows[g], x.cols, x.ld, xs->partials[g], c.st));
            xs->chunks[g] = -1;
        }
}
inline bool all_sums(const Ctx& c, const Stat* s) {
    if (!s) return false;
    for (int g = 0; g < c.G; ++g)
        if (s->chunks[g] != -1) return false;
    return true;
}

// y = lrelu(IN(x) [+ IN(res) | + res], slope)
void norm_act(Ctx& c, const Mat& x, float slope, const Mat& y, Stat* xs = nullptr, const Mat* res = nullptr,
              bool norm_res = false, Stat* rs = nullptr) {
    fill_sums(c, x, xs);
    if (res && norm_res) fill_sums(c, *res, rs);
    if (c.live() && all_sums(c, xs) && (!res || !norm_res || all_sums(c, rs)) && sums_apply_ok(x, y, res)) {
        // every pair of the call in ONE launch (round 5: a launch per pair cost the stream ~5 us of queue time each)
        NormJob jobs[GMAX];
        bool al = true;
        for (int g = 0; g < c.G; ++g) {
            jobs[g] = NormJob{x.p[g], static_cast<const double*>(xs->partials[g]), res ? res->p[g] : nullptr,
                              (res && norm_res) ? static_cast<const double*>(rs->partials[g]) : nullptr, y.p[g], nullptr, nullptr,
                              x.rows[g], (double)(x.rows[g] > 0 ? x.rows[g] : 1)};
            al = al && ((reinterpret_cast<uintptr_t>(x.p[g]) | reinterpret_cast<uintptr_t>(y.p[g]) |
                         reinterpret_cast<uintptr_t>(res ? res->p[g] : nullptr)) & 15) == 0;
        }
        if (al && c.G <= 4) {
            c.check(instnorm_apply_sums_multi(jobs, c.G, x.cols, x.ld, 1e-5f, res ? res->ld : 0, slope, y.ld, false, c.st));
            return;
        }
        for (int g = 0; g < c.G; ++g)
            c.check(pcrcg_instnorm_apply_sums(x.p[g], x.rows[g], x.cols, x.ld, xs->partials[g], (double)x.rows[g], 1e-5f,
                                              res ? res->p[g] : nullptr, res ? res->ld : 0,
                                              (res && norm_res) ? rs->partials[g] : nullptr, slope, y.p[g], y.ld, c.st));
        return;
    }
    const size_t m = c.mark();
    float* stats = static_cast<float*>(c.raw(sizeof(float) * 2 * x.cols));
    float* rstats = (res && norm_res) ? static_cast<float*>(c.raw(sizeof(float) * 2 * x.cols)) : nullptr;
    const size_t wsb = pcrcg_instnorm_ws_bytes(x.cols);
    void* ws = c.raw(wsb);
    if (c.live())
        for (int g = 0; g < c.G; ++g) {
            col_stats(c, x, g, xs, stats, ws, wsb);
            if (rstats) col_stats(c, *res, g, rs, rstats, ws, wsb);
            c.check(pcrcg_instnorm_apply(x.p[g], x.rows[g], x.cols, x.ld, stats, res ? res->p[g] : nullptr, res ? res->ld : 0,
                                         rstats, slope, y.p[g], y.ld, c.st));
        }
    c.release(m);
}

// the further pairs' sides of a product
GemmGroup group_of(const Ctx& c, const Mat& x, const Mat& y, Stat* st, const float* const* row_scale = nullptr) {
    GemmGroup grp;
    grp.n = c.G - 1;
    for (int g = 1; g < c.G; ++g) {
        GemmPair& p = grp.p[g - 1];
        p.a = x.p[g];
        p.c = y.p[g];
        p.m = x.rows[g];
        p.row_scale = row_scale ? row_scale[g] : nullptr;
        p.colstats = st ? st->partials[g] : nullptr;
        p.h_chunks = st ? &st->chunks[g] : nullptr;
        p.c_zeroed = y.zeroed;
    }
    return grp;
}

// y = x @ w^T (+ bias); w is [out, in] with leading dimension ldw; optionally leaves the column statistics of y for the
// InstanceNorm that follows.  row_scale: per-pair row factors (KPConv's 1 / neighbour count)
void linear(Ctx& c, const Mat& x, const float* w, int ldw, const float* bias, const Mat& y, Stat* st = nullptr,
            const float* const* row_scale = nullptr) {
    if (!c.live()) return;
    if (c.paired()) {
        GemmGroup p = group_of(c, x, y, st, row_scale);
        c.check(gemm_bt_colstats_pair(x.p[0], x.ld, w, ldw, y.p[0], y.ld, x.rows[0], y.cols, x.cols, row_scale ? row_scale[0] : nullptr,
                                      bias, st ? st->partials[0] : nullptr, st ? st->bytes : 0, st ? &st->chunks[0] : nullptr, c.st,
                                      y.zeroed, st && st->sums, &p));
        return;
    }
    for (int g = 0; g < c.G; ++g)
        c.check(gemm_bt_colstats(x.p[g], x.ld, w, ldw, y.p[g], y.ld, x.rows[g], y.cols, x.cols, row_scale ? row_scale[g] : nullptr,
                                 bias, st ? st->partials[g] : nullptr, st ? st->bytes : 0, st ? &st->chunks[g] : nullptr, c.st,
                                 y.zeroed, st && st->sums));
}

// C (+)= f(A)[rows] @ w^T: the GemmExtra forms, for all pairs.  idx / ns: per-pair gather table (first column) and source
// row count, sums: per-pair column sums of the raw A (normalise-on-load), or NULL
void linear_extra(Ctx& c, const Mat& a, const float* w, int ldw, const float* bias, const Mat& y, int out_rows_of_table,
                  const pcrcg_table* const* tabs, const float* zero_row, Stat* a_sums, float a_slope, bool accumulate,
                  bool c_zeroed, Stat* st = nullptr) {
    if (!c.live()) return;
    (void)out_rows_of_table;
    auto extra = [&](int g) {
        GemmExtra ex;
        if (a_sums) {
            ex.a_sums = static_cast<const double*>(a_sums->partials[g]);
            ex.a_count = (double)a.rows[g];
            ex.a_slope = a_slope;
        }
        if (tabs) {
            ex.a_idx = reinterpret_cast<const long long*>(tabs[g]->idx);
            ex.a_idx_ld = tabs[g]->ld;
            ex.a_ns = a.rows[g];
            ex.a_zero = zero_row;
        }
        ex.accumulate = accumulate;
        return ex;
    };
    auto m_of = [&](int g) { return tabs ? tabs[g]->rows : a.rows[g]; };
    bool same_ld = true;
    for (int g = 1; g < c.G && tabs; ++g) same_ld = same_ld && tabs[g]->ld == tabs[0]->ld;
    if (c.paired() && same_ld) {
        GemmExtra ex = extra(0);
        GemmGroup p;
        p.n = c.G - 1;
        for (int g = 1; g < c.G; ++g) {
            GemmPair& q = p.p[g - 1];
            q.a = a.p[g];
            q.c = y.p[g];
            q.m = m_of(g);
            q.colstats = st ? st->partials[g] : nullptr;
            q.h_chunks = st ? &st->chunks[g] : nullptr;
            q.c_zeroed = c_zeroed;
            if (tabs) { q.a_idx = reinterpret_cast<const long long*>(tabs[g]->idx); q.a_ns = a.rows[g]; }
            if (a_sums) { q.a_sums = static_cast<const double*>(a_sums->partials[g]); q.a_count = (double)a.rows[g]; }
        }
        c.check(gemm_bt_extra_pair(a.p[0], a.ld, w, ldw, y.p[0], y.ld, m_of(0), y.cols, a.cols, c.st, c_zeroed, ex, bias,
                                   st ? st->partials[0] : nullptr, st ? st->bytes : 0, st ? &st->chunks[0] : nullptr,
                                   st && st->sums, &p));
        return;
    }
    for (int g = 0; g < c.G; ++g) {
        GemmExtra ex = extra(g);
        c.check(gemm_bt_extra(a.p[g], a.ld, w, ldw, y.p[g], y.ld, m_of(g), y.cols, a.cols, c.st, c_zeroed, ex, bias,
                              st ? st->partials[g] : nullptr, st ? st->bytes : 0, st ? &st->chunks[g] : nullptr, st && st->sums));
    }
}

// y = lrelu(IN(x), slope) @ w^T (+ bias) with the normalisation done inside the product's A loads (GemmExtra::a_sums):
// x is the RAW output of the producing product and xs its column sums.  Returns false when that form does not apply
// (the caller then normalises into a matrix of its own and calls linear()).
bool norm_fuse_on() { return debug_opts().fuse_norm && gemm_extra_ok(); }
bool lazy_stats_ready(Ctx& c, const Mat& x, Stat* xs) {
    if (!xs || !xs->sums || x.ld % 4 != 0 || x.cols > 4096) return false;
    fill_sums(c, x, xs);              // nobody left the sums yet (split-K or accumulated output): one pass
    return true;
}
bool linear_norm(Ctx& c, const Mat& x, Stat* xs, float slope, const float* w, int ldw, const float* bias, const Mat& y,
                 Stat* st = nullptr) {
    if (!norm_fuse_on() || !lazy_stats_ready(c, x, xs)) return false;
    linear_extra(c, x, w, ldw, bias, y, 0, nullptr, nullptr, xs, slope, false, y.zeroed, st);
    return true;
}

struct Batches { const pcrcg_batch* b[GMAX]; };

void kpconv(Ctx& c, const Batches& B, const pcrcg_block& blk, const Mat& x, const Mat& y, Stat* st = nullptr,
            void* const* packed_ws = nullptr) {
    const int l = blk.layer;
    const size_t m = c.mark();
    // channel counts that are not a multiple of 4 (the 129-channel PCR-CG input): zero-padded copy of the
    // features + zero-padded weights, so that the MFMA gather kernel applies (zeros change neither the sums nor
    // the neighbour count of the normaliser)
    Mat xin = x;
    const float* w = blk.kp_w;
    int cin = x.cols;
    if (blk.kp_w_pad && blk.cin_pad > x.cols) {
        cin = blk.cin_pad;
        xin = c.mat(x.rows, cin);
        if (c.live())
            for (int g = 0; g < c.G; ++g) {
                c.check(hipMemsetAsync(xin.p[g], 0, sizeof(float) * (size_t)x.rows[g] * cin, c.st) == hipSuccess ? PCRCG_OK
                                                                                                                   : PCRCG_ELAUNCH);
                c.check(pcrcg_copy2d(x.p[g], x.ld, xin.p[g], cin, x.rows[g], x.cols, c.st));
            }
        w = blk.kp_w_pad;
    }
    int nq[GMAX], ns[GMAX];
    const float* q[GMAX];
    const pcrcg_table* tab[GMAX];
    float* inv_n[GMAX];
    void* ws[GMAX];
    size_t wsb[GMAX];
    for (int g = 0; g < c.G; ++g) {
        const pcrcg_batch& b = *B.b[g];
        tab[g] = blk.strided ? &b.pools[l] : &b.neighbors[l];
        q[g] = blk.strided ? b.points[l + 1] : b.points[l];
        nq[g] = blk.strided ? b.n_points[l + 1] : b.n_points[l];
        ns[g] = b.n_points[l];
        inv_n[g] = static_cast<float*>(c.raw(sizeof(float) * (nq[g] > 0 ? nq[g] : 1)));
        wsb[g] = pcrcg_kpconv_ws_bytes(ns[g]);
        ws[g] = (packed_ws && packed_ws[g]) ? packed_ws[g] : c.raw(wsb[g]);
    }
    // cin = 1 (the first layer of the geometry-only configurations) with a K-contiguous weight copy: that copy has rows of 16
    // floats (the 16th zero, pcrcg_amd/runner.py) and the gather kernel writes wf in rows of 16 -- whole k-steps, so the
    // contraction is the grouped fp16 A B^T product with the statistics in its epilogue like every other layer's (round 6;
    // before: one k-major six-product launch per pair and a column-sum pass over its output)
    const bool c1_16 = cin == 1 && blk.kp_wt != nullptr && debug_opts().c1_rows16 != 0;
    const float* const kp_wt = (cin == 1 && !c1_16) ? nullptr : blk.kp_wt;     // (cin = 1: the copy is the 16-float form or nothing)
    const int kk = c1_16 ? 16 : PCRCG_KPOINTS * cin;
    // bf16 feature storage (pcrcg_model.feature_bf16): the gathers read a bf16 copy of x and wf is bf16 in HBM -- half
    // the bytes of the two streams that bound the encoder; the contraction takes wf as the (single-term) bf16 operand
    // against the exact three-term split of the fp32 weights, fp32 accumulate and fp32 output.
    if (c.bf16 && kp_wt && cin % 32 == 0) {
        for (int g = 0; g < c.G; ++g) {
            const pcrcg_batch& b = *B.b[g];
            void* xb = c.raw(sizeof(unsigned short) * (size_t)(ns[g] > 0 ? ns[g] : 1) * cin);
            void* wfb = c.raw(sizeof(unsigned short) * (size_t)(nq[g] > 0 ? nq[g] : 1) * kk);
            if (c.live()) {
                c.check(pcrcg_kpconv_aggregate_bf16(q[g], nq[g], b.points[l], ns[g], tab[g]->idx, tab[g]->cols, tab[g]->ld, xin.p[g],
                                                    cin, blk.kp, blk.extent, xb, wfb, inv_n[g], ws[g], wsb[g], c.st));
                c.check(gemm_bf16a_bt_colstats(wfb, kk, kp_wt, kk, y.p[g], y.ld, nq[g], y.cols, kk, inv_n[g], nullptr,
                                               st ? st->partials[g] : nullptr, st ? st->bytes : 0, st ? &st->chunks[g] : nullptr,
                                               c.st, y.zeroed, st && st->sums));
            }
        }
        c.release(m);
        return;
    }
    Mat wf = c.mat(nq, kk);
    if (c.live()) {
        for (int g = 0; g < c.G; ++g) {
            const pcrcg_batch& b = *B.b[g];
            if (c1_16)
                c.check(kpconv_aggregate_rows(q[g], nq[g], b.points[l], ns[g], tab[g]->idx, tab[g]->cols, tab[g]->ld, xin.p[g], cin,
                                              blk.kp, blk.extent, wf.p[g], inv_n[g], ws[g], wsb[g], c.st, /*pack=*/true,
                                              /*stream_out=*/true, 16));
            else if (packed_ws && packed_ws[g] && xin.p[g] == x.p[g])
                c.check(kpconv_aggregate_rows(q[g], nq[g], b.points[l], ns[g], tab[g]->idx, tab[g]->cols, tab[g]->ld, xin.p[g], cin,
                                              blk.kp, blk.extent, wf.p[g], inv_n[g], ws[g], wsb[g], c.st, /*pack=*/false,
                                              /*stream_out=*/true));
            else
                c.check(pcrcg_kpconv_aggregate(q[g], nq[g], b.points[l], ns[g], tab[g]->idx, tab[g]->cols, tab[g]->ld, xin.p[g], cin,
                                               blk.kp, blk.extent, wf.p[g], inv_n[g], ws[g], wsb[g], c.st));
        }
        // contraction wf @ W: against the K-contiguous copy wt [cout, 15*cin] when the descriptor carries one
        // (C = A * B^T form: both operands k-contiguous, the form the split-bf16 GEMM is built for).
        // (Measured and not adopted, rounds 2 and 3: aggregating + contracting row chunks so that wf stays in L2 /
        // Infinity Cache between the two kernels -- isolated 3.50 vs 3.31 ms per forward with 48 MB chunks; inside the
        // engine 416 / 446 / 459 / 462 pairs/s with 16 / 32 / 64 / 120 MB chunks against 460-464 unchunked: the extra
        // launches cost more than the on-chip re-read saves.)
        if (kp_wt)
            linear(c, wf, kp_wt, kk, nullptr, y, st, inv_n);
        else   // (descriptor without the K-contiguous weight copy: the plain entry point knows only the partials layout)
            for (int g = 0; g < c.G; ++g)
                c.check(pcrcg_gemm_f32_colstats(wf.p[g], wf.ld, w, y.cols, 0, y.p[g], y.ld, nq[g], y.cols, kk, inv_n[g], nullptr,
                                                (st && !st->sums) ? st->partials[g] : nullptr, (st && !st->sums) ? st->bytes : 0,
                                                (st && !st->sums) ? &st->chunks[g] : nullptr, c.st));
    }
    c.release(m);
}

// input channels of the block's KPConv as the gather kernel sees them (padded to a multiple of 4)
int kp_cin(const pcrcg_block& blk, const Mat& x) { return (blk.kp_w_pad && blk.cin_pad > x.cols) ? blk.cin_pad : x.cols; }
// length of a wf row (the contraction's K)
int kp_k(const pcrcg_block& blk, const Mat& x) {
    return (x.cols == 1 && blk.kp_wt && debug_opts().c1_rows16) ? 16 : PCRCG_KPOINTS * kp_cin(blk, x);
}

void out_rows(const Ctx& c, const Batches& B, const pcrcg_block& blk, int* rows) {
    for (int g = 0; g < GMAX; ++g)
        rows[g] = g < c.G ? (blk.strided ? B.b[g]->n_points[blk.layer + 1] : B.b[g]->n_points[blk.layer]) : 0;
}

// SimpleBlock.forward (ref:models/blocks.py:578-590)
Mat simple_block(Ctx& c, const Batches& B, const pcrcg_block& blk, const Mat& x) {
    int nq[GMAX];
    out_rows(c, B, blk, nq);
    Mat y = c.mat(nq, blk.mid_dim);
    const size_t m = c.mark();
    Mat t = c.gemm_out(nq, y.cols, kp_k(blk, x));
    Stat ts = stat_buffer(c, nq, t.cols);
    kpconv(c, B, blk, x, t, &ts);
    norm_act(c, t, 0.1f, y, &ts);
    c.release(m);
    return y;
}

// u = lrelu(IN(t), slope) for a u that is the input of the block's KPConv: the normalisation kernel leaves the KPConv's
// support records (coordinates + "feature row sums to a positive value" flag) in kp_ws as it goes, so the KPConv needs no
// pass over u of its own (10 launches per S30k forward: +4 % pairs/s with them knocked out).  False: not applicable.
bool norm_act_pack(Ctx& c, const Batches& B, int layer, const Mat& t, float slope, const Mat& u, Stat* ts, void* const* kp_ws,
                   const size_t* kp_ws_bytes) {
    if (!debug_opts().fuse_pack || c.bf16 || !instnorm_pack_ok(t.cols, t.ld, u.ld) || u.ld != u.cols) return false;
    const size_t m = c.mark();
    float* stats = static_cast<float*>(c.raw(sizeof(float) * 2 * t.cols));
    const size_t wsb = pcrcg_instnorm_ws_bytes(t.cols);
    void* ws = c.raw(wsb);
    fill_sums(c, t, ts);              // a split-K product left nothing: one pass into the accumulators
    if (c.live() && all_sums(c, ts) && c.G <= 4) {          // every pair of the call in one launch
        NormJob jobs[GMAX];
        bool ok = true;
        for (int g = 0; g < c.G; ++g) {
            float4* pk = kpconv_pk_ptr(kp_ws[g], kp_ws_bytes[g], t.rows[g]);
            ok = ok && pk != nullptr && ((reinterpret_cast<uintptr_t>(t.p[g]) | reinterpret_cast<uintptr_t>(u.p[g])) & 15) == 0;
            jobs[g] = NormJob{t.p[g], static_cast<const double*>(ts->partials[g]), nullptr, nullptr, u.p[g], B.b[g]->points[layer], pk,
                              t.rows[g], (double)(t.rows[g] > 0 ? t.rows[g] : 1)};
        }
        if (ok) {
            c.check(instnorm_apply_sums_multi(jobs, c.G, t.cols, t.ld, 1e-5f, 0, slope, u.ld, true, c.st));
            c.release(m);
            return true;
        }
    }
    if (c.live())
        for (int g = 0; g < c.G; ++g) {
            const float* s_pts = B.b[g]->points[layer];
            float4* pk = kpconv_pk_ptr(kp_ws[g], kp_ws_bytes[g], t.rows[g]);
            if (!pk) c.check(PCRCG_EWORKSPACE);
            else if (ts && ts->chunks[g] == -1)
                c.check(instnorm_apply_pack(t.p[g], t.rows[g], t.cols, t.ld, nullptr, static_cast<const double*>(ts->partials[g]),
                                            (double)t.rows[g], 1e-5f, slope, u.p[g], u.ld, s_pts, pk, c.st));
            else {
                col_stats(c, t, g, ts, stats, ws, wsb);
                c.check(instnorm_apply_pack(t.p[g], t.rows[g], t.cols, t.ld, stats, nullptr, 0.0, 1e-5f, slope, u.p[g], u.ld, s_pts,
                                            pk, c.st));
            }
        }
    c.release(m);
    return true;
}

// ResnetBottleneckBlock.forward (ref:models/blocks.py:650-678)
Mat resnet_block(Ctx& c, const Batches& B, const pcrcg_block& blk, const Mat& feats) {
    int nq[GMAX];
    out_rows(c, B, blk, nq);
    Mat y = c.mat(nq, blk.out_dim);
    const size_t m = c.mark();
    Mat x = feats;
    void* kp_ws[GMAX] = {};
    size_t kp_wsb[GMAX] = {};
    bool packed = false;
    if (blk.unary1) {
        Mat t = c.gemm_out(feats.rows, blk.mid_dim, feats.cols), u = c.mat(feats.rows, blk.mid_dim);
        Stat ts = stat_buffer(c, t.rows, t.cols);
        linear(c, feats, blk.unary1, feats.cols, nullptr, t, &ts);
        // u feeds the KPConv whose supports are this block's input rows: pack its support records on the way
        if (blk.mid_dim % 4 == 0 && !(blk.kp_w_pad && blk.cin_pad > blk.mid_dim)) {
            for (int g = 0; g < c.G; ++g) {
                kp_wsb[g] = pcrcg_kpconv_ws_bytes(feats.rows[g]);
                kp_ws[g] = c.raw(kp_wsb[g]);
            }
            packed = norm_act_pack(c, B, blk.layer, t, 0.1f, u, &ts, kp_ws, kp_wsb);
        }
        if (!packed) norm_act(c, t, 0.1f, u, &ts);
        x = u;
    }
    Mat k = c.gemm_out(nq, blk.mid_dim, PCRCG_KPOINTS * kp_cin(blk, x)), kn = c.mat(nq, blk.mid_dim);
    Stat ks = stat_buffer(c, nq, blk.mid_dim);
    kpconv(c, B, blk, x, k, &ks, packed ? kp_ws : nullptr);
    Mat u2 = c.gemm_out(nq, blk.out_dim, blk.mid_dim);
    Stat u2s = stat_buffer(c, nq, blk.out_dim);
    // lrelu(IN(k)) is read by unary2 alone: its product normalises k on load when k's statistics are column sums
    if (!linear_norm(c, k, &ks, 0.1f, blk.unary2, blk.mid_dim, nullptr, u2, &u2s)) {
        norm_act(c, k, 0.1f, kn, &ks);
        linear(c, kn, blk.unary2, blk.mid_dim, nullptr, u2, &u2s);
    }
    Mat sc = feats;
    if (blk.strided) {   // max_pool shortcut (:672-673)
        sc = c.mat(nq, feats.cols);
        if (c.live() && c.G <= 4) {          // every pair's pool in one launch
            GatherJob jobs[GMAX];
            for (int g = 0; g < c.G; ++g) {
                const pcrcg_table& t = B.b[g]->pools[blk.layer];
                jobs[g] = GatherJob{feats.p[g], t.idx, sc.p[g], feats.rows[g], nq[g], t.cols, t.ld};
            }
            c.check(gather_max_multi(jobs, c.G, feats.cols, c.st));
        } else if (c.live())
            for (int g = 0; g < c.G; ++g) {
                const pcrcg_table& t = B.b[g]->pools[blk.layer];
                c.check(pcrcg_gather_max(feats.p[g], feats.rows[g], feats.cols, t.idx, nq[g], t.cols, t.ld, sc.p[g], c.st));
            }
    }
    if (blk.shortcut && blk.layer < debug_opts().knock_tail) {
        norm_act(c, u2, 0.1f, y, &u2s);
    } else if (blk.shortcut) {
        Mat s2 = c.gemm_out(nq, blk.out_dim, sc.cols);
        Stat s2s = stat_buffer(c, nq, blk.out_dim);
        linear(c, sc, blk.shortcut, sc.cols, nullptr, s2, &s2s);
        norm_act(c, u2, 0.1f, y, &u2s, &s2, true, &s2s);      // lrelu(IN(unary2) + IN(shortcut))
    } else {
        norm_act(c, u2, 0.1f, y, &u2s, &sc, false);           // lrelu(IN(unary2) + shortcut)
    }
    c.release(m);
    return y;
}

// x_out = lrelu(IN2d(max_j e), 0.2) for e[i,j,:] = cn[i, :cw] + cn[idx[i,j], cw:2cw]: reduction + normalisation
// (max commutes with the monotone normalise + LeakyReLU); statistics as fp64 sums when the widths allow, else as pairs
void edge_norm(Ctx& c, const float* cn, int ld_cn, int cw, const int* idx, int n, int k, float* emax, int ld_emax, float* out,
               int ld_out, void* sums, bool sums_ok, float* stats, void* ws, size_t wsb) {
    if (sums && sums_ok) {
        c.check(pcrcg_edgeconv_reduce_sums(cn, ld_cn, cn + cw, ld_cn, idx, n, k, cw, emax, ld_emax, sums, c.st));
        c.check(pcrcg_instnorm_apply_sums(emax, n, cw, ld_emax, sums, (double)n * (double)k, 1e-5f, nullptr, 0, nullptr, 0.2f, out,
                                          ld_out, c.st));
        return;
    }
    c.check(pcrcg_edgeconv_reduce(cn, ld_cn, cn + cw, ld_cn, idx, n, k, cw, 1e-5f, emax, ld_emax, stats, ws, wsb, c.st));
    c.check(pcrcg_instnorm_apply(emax, n, cw, ld_emax, stats, nullptr, 0, nullptr, 0.2f, out, ld_out, c.st));
}

// edge_norm for every cloud of the call: the reductions of all of them in ONE launch where the row-parallel kernel applies
void edge_norm_all(Ctx& c, const Mat& cn, int cw, int* const* idx, const int* n, const int* k, const Mat& e, const Mat& out,
                   void* const* sums, bool sums_ok, float* stats, void* ws, size_t wsb) {
    bool multi = sums_ok;
    EdgeCloud cl[GMAX];
    for (int g = 0; g < c.G; ++g) {
        multi = multi && sums[g] != nullptr;
        cl[g] = EdgeCloud{cn.p[g], cn.p[g] + cw, idx[g], e.p[g], static_cast<double*>(sums[g]), n[g], k[g]};
    }
    if (multi && edgeconv_rows_ok(cl, c.G, cn.ld, cn.ld, e.ld, cw)) {
        c.check(edgeconv_rows_multi(cl, c.G, cn.ld, cn.ld, e.ld, cw, true, nullptr, c.st));
        NormJob jobs[GMAX];
        bool al = c.G <= 4;
        for (int g = 0; g < c.G; ++g) {
            jobs[g] = NormJob{e.p[g], static_cast<const double*>(sums[g]), nullptr, nullptr, out.p[g], nullptr, nullptr, n[g],
                              (double)n[g] * (double)k[g]};
            al = al && ((reinterpret_cast<uintptr_t>(e.p[g]) | reinterpret_cast<uintptr_t>(out.p[g])) & 15) == 0 && n[g] > 0 && k[g] > 0;
        }
        if (al) {
            c.check(instnorm_apply_sums_multi(jobs, c.G, cw, e.ld, 1e-5f, 0, 0.2f, out.ld, false, c.st));
            return;
        }
        for (int g = 0; g < c.G; ++g)
            c.check(pcrcg_instnorm_apply_sums(e.p[g], n[g], cw, e.ld, sums[g], (double)n[g] * (double)k[g], 1e-5f, nullptr, 0, nullptr,
                                              0.2f, out.p[g], out.ld, c.st));
        return;
    }
    for (int g = 0; g < c.G; ++g)
        edge_norm(c, cn.p[g], cn.ld, cw, idx[g], n[g], k[g], e.p[g], e.ld, out.p[g], out.ld, sums[g], sums_ok, stats, ws, wsb);
}

// SelfAttention.forward (ref:models/gcn.py:110-134) on row-major [n, ch]: one cloud of every pair
// knn_given: the clouds' kNN tables when the caller has them already (they depend on the coordinates only, which no layer
// changes: the forward computes them once for all self-attention layers), or NULL
Mat self_attention(Ctx& c, const pcrcg_model& mdl, const pcrcg_gnn_layer& gl, const float* const* coords, const Mat& f,
                   int* const* knn_given = nullptr) {
    const int ch = f.cols;
    Mat y = c.mat(f.rows, ch);
    const size_t m = c.mark();
    int kq[GMAX];
    int* idx[GMAX];
    for (int g = 0; g < c.G; ++g) {
        const int n = f.rows[g];
        kq[g] = mdl.knn_k < n - 1 ? mdl.knn_k : n - 1;
        idx[g] = knn_given ? knn_given[g] : static_cast<int*>(c.raw(sizeof(int) * (size_t)n * (kq[g] > 0 ? kq[g] : 1)));
    }
    Mat cat = c.mat(f.rows, 4 * ch);
    const size_t wsb = pcrcg_edgeconv_ws_bytes(2 * ch);
    void* ws = c.raw(wsb);
    float* stats = static_cast<float*>(c.raw(sizeof(float) * 4 * ch));
    Mat cn1 = c.gemm_out(f.rows, 2 * ch, ch), e1 = c.mat(f.rows, ch), cn2 = c.gemm_out(f.rows, 4 * ch, ch), e2 = c.mat(f.rows, 2 * ch);
    Mat x3 = c.gemm_out(f.rows, ch, 4 * ch);
    Stat x3s = stat_buffer(c, f.rows, ch);
    // InstanceNorm2d statistics of the two edge convolutions as fp64 sums (zero arena) when that form applies
    void *sums1[GMAX], *sums2[GMAX];
    for (int g = 0; g < c.G; ++g) {
        sums1[g] = gemm_colstats_sums_ok() ? c.zraw(2 * sizeof(double) * (size_t)ch) : nullptr;
        sums2[g] = gemm_colstats_sums_ok() ? c.zraw(2 * sizeof(double) * (size_t)(2 * ch)) : nullptr;
    }
    Mat cat1 = cols(cat, ch, ch), cat2 = cols(cat, 2 * ch, 2 * ch);
    const bool ok1 = sums_apply_ok(e1, cat1, nullptr), ok2 = sums_apply_ok(e2, cat2, nullptr);
    if (c.live()) {
        if (!knn_given)
            for (int g = 0; g < c.G; ++g) c.check(pcrcg_knn(coords[g], f.rows[g], kq[g], idx[g], c.st));
        if (c.G <= 4) c.check(copy2d_multi(f.p, cat.p, f.rows, c.G, f.ld, cat.ld, ch, c.st));                // x0, every cloud
        else
            for (int g = 0; g < c.G; ++g) c.check(pcrcg_copy2d(f.p[g], f.ld, cat.p[g], cat.ld, f.rows[g], ch, c.st));
        // x1 = max_k lrelu(IN2d(conv1(cat(f_i, f_j - f_i))))  (:121-125)
        linear(c, f, gl.edge1, ch, nullptr, cn1);
        edge_norm_all(c, cn1, ch, idx, f.rows, kq, e1, cat1, sums1, ok1, stats, ws, wsb);
        // x2 from x1 with conv2 (:127-129)
        linear(c, cat1, gl.edge2, ch, nullptr, cn2);
        edge_norm_all(c, cn2, 2 * ch, idx, f.rows, kq, e2, cat2, sums2, ok2, stats, ws, wsb);
        // x3 = lrelu(IN(conv3(cat(x0,x1,x2))))  (:131-132)
        linear(c, cat, gl.conv3, 4 * ch, nullptr, x3, &x3s);
    }
    norm_act(c, x3, 0.2f, y, &x3s);
    c.release(m);
    return y;
}

// x + AttentionalPropagation(x, src)  (ref:models/gcn.py:151-185, 213-214)
// q_given: the query projection of x, computed by the caller (both directions of a layer in one launch), or NULL
Mat cross_attention(Ctx& c, const pcrcg_model& mdl, const pcrcg_gnn_layer& gl, const Mat& x, const Mat& src, const Mat* q_given = nullptr) {
    const int ch = x.cols, h = mdl.heads, d = ch / h;
    Mat y = c.mat(x.rows, ch);
    const size_t m = c.mark();
    // the key and value projections read the same rows: ONE product of width 2 ch when the caller packed the two weight
    // matrices (and biases) behind each other (pcrcg_amd/runner.py does), k and v are then column blocks of its output
    const bool kv_fused = gl.wv == gl.wk + (size_t)ch * ch && gl.bv == gl.bk + ch && debug_opts().gnn_merge;
    Mat q = q_given ? *q_given : c.gemm_out(x.rows, ch, ch), kk, v, msg = c.mat(x.rows, ch);
    if (kv_fused) {
        Mat kv = c.gemm_out(src.rows, 2 * ch, ch);
        linear(c, src, gl.wk, ch, gl.bk, kv);
        kk = cols(kv, 0, ch);
        v = cols(kv, ch, ch);
        kk.zeroed = v.zeroed = false;
    } else {
        kk = c.gemm_out(src.rows, ch, ch);
        v = c.gemm_out(src.rows, ch, ch);
    }
    int sc_rows[GMAX] = {};
    for (int g = 0; g < c.G; ++g) sc_rows[g] = x.rows[g];
    int ms_max = 0;
    for (int g = 0; g < c.G; ++g) ms_max = src.rows[g] > ms_max ? src.rows[g] : ms_max;
    Mat sc = c.mat(sc_rows, ms_max);
    int r2[GMAX];
    for (int g = 0; g < GMAX; ++g) r2[g] = x.rows[g];
    Mat cat = c.mat(r2, 2 * ch), h0 = c.gemm_out(r2, 2 * ch, 2 * ch), h1 = c.mat(r2, 2 * ch), delta = c.gemm_out(r2, ch, 2 * ch);
    if (!q_given) linear(c, x, gl.wq, ch, gl.bq, q);
    if (!kv_fused) {
        linear(c, src, gl.wk, ch, gl.bk, kk);
        linear(c, src, gl.wv, ch, gl.bv, v);
    }
    bool att_done = false;
    if (c.live()) {       // every pair's attention in ONE launch where the matrix-core kernel applies (round 5)
        AttnCloud cl[GMAX];
        for (int g = 0; g < c.G; ++g) cl[g] = AttnCloud{q.p[g], kk.p[g], v.p[g], msg.p[g], x.rows[g], src.rows[g]};
        if (attention_mfma_ok(cl, c.G, q.ld, kk.ld, v.ld, d)) {
            c.check(attention_mfma_multi(cl, c.G, q.ld, kk.ld, v.ld, msg.ld, h, d, 1.0f / sqrtf((float)d), c.st));
            att_done = true;
        }
    }
    if (c.live())
        for (int g = 0; g < c.G; ++g) {
            const int n = x.rows[g], ms = src.rows[g];
            if (att_done) {
            } else if (pcrcg_attention_supported(d) && (long)n * ms <= 1000000) {
                // all heads in one launch (heads are contiguous column blocks after the weight permutation); a latency win on
                // the few hundred coarse points of an indoor pair (49 vs 138 us of kernels + 11 launches less per call,
                // scripts/attention_bench.py) -- beyond ~1000 x 1000 the GEMM path below is faster (1900 x 1900: 228 vs 640 us)
                c.check(pcrcg_attention(q.p[g], q.ld, kk.p[g], kk.ld, v.p[g], v.ld, msg.p[g], msg.ld, n, ms, h, d,
                                        1.0f / sqrtf((float)d), c.st));
            } else {
                for (int i = 0; i < h; ++i) {
                    c.check(pcrcg_gemm_f32(q.p[g] + i * d, q.ld, kk.p[g] + i * d, kk.ld, 1, sc.p[g], ms, n, ms, d, nullptr, nullptr,
                                           c.st));
                    c.check(pcrcg_softmax_rows(sc.p[g], n, ms, ms, 1.0f / sqrtf((float)d), c.st));
                    c.check(pcrcg_gemm_f32(sc.p[g], ms, v.p[g] + i * d, v.ld, 0, msg.p[g] + i * d, msg.ld, n, d, ms, nullptr, nullptr,
                                           c.st));
                }
            }
        }
    if (c.live() && c.G <= 4) c.check(copy2d_multi(x.p, cat.p, x.rows, c.G, x.ld, cat.ld, ch, c.st));
    else if (c.live())
        for (int g = 0; g < c.G; ++g) c.check(pcrcg_copy2d(x.p[g], x.ld, cat.p[g], cat.ld, x.rows[g], ch, c.st));
    linear(c, msg, gl.wm, ch, gl.bm, cols(cat, ch, ch));       // merge, written next to x: cat([x, message])
    Stat h0s = stat_buffer(c, r2, 2 * ch);
    linear(c, cat, gl.w0, 2 * ch, gl.b0, h0, &h0s);
    if (!linear_norm(c, h0, &h0s, 0.0f, gl.w3, 2 * ch, gl.b3, delta)) {      // InstanceNorm1d + ReLU, inside w3's A loads
        norm_act(c, h0, 0.0f, h1, &h0s);
        linear(c, h1, gl.w3, 2 * ch, gl.b3, delta);
    }
    if (c.live())
        for (int g = 0; g < c.G; ++g) c.check(pcrcg_add(x.p[g], delta.p[g], y.p[g], (long)x.rows[g] * ch, c.st));
    c.release(m);
    return y;
}

void forward(Ctx& c, const pcrcg_model& mdl, const Batches& B, const pcrcg_outputs* out) {
    const int L = B.b[0]->n_levels;
    c.bf16 = mdl.feature_bf16 != 0;
    Mat x;
    for (int g = 0; g < c.G; ++g) {
        x.p[g] = const_cast<float*>(B.b[g]->features);
        x.rows[g] = B.b[g]->n_points[0];
    }
    x.cols = x.ld = B.b[0]->feat_dim;
    std::vector<Mat> skips;
    // 1. encoder (:519-524)
    for (int i = 0; i < mdl.n_enc; ++i) {
        if (mdl.enc_skip[i]) skips.push_back(x);
        const pcrcg_block& blk = mdl.enc[i];
        x = blk.type == PCRCG_BLK_SIMPLE ? simple_block(c, B, blk, x) : resnet_block(c, B, blk, x);
    }
    // 2. bottleneck (:527-528) and 3. GNN (:532-536)
    const int gd = mdl.gnn_dim;
    int nc[GMAX] = {}, ns[GMAX] = {}, nt[GMAX] = {}, zero[GMAX] = {};
    const float *c0[GMAX], *c1[GMAX];
    for (int g = 0; g < c.G; ++g) {
        nc[g] = B.b[g]->n_points[L - 1];
        ns[g] = B.b[g]->len_src_c;
        nt[g] = nc[g] - ns[g];
        c0[g] = B.b[g]->points[L - 1];
        c1[g] = B.b[g]->points[L - 1] + 3 * (long)ns[g];
    }
    Mat fc = c.gemm_out(nc, gd, mdl.enc_out_dim);
    linear(c, x, mdl.bottle_w, mdl.enc_out_dim, mdl.bottle_b, fc);
    Mat d0 = rows(fc, zero, ns), d1 = rows(fc, ns, nt);
    // the kNN graphs of the coarse clouds: geometry only (ref:models/gcn.py:48-51 builds them from the coordinates in every
    // self-attention layer: the same tables each time) -- once per forward (round 5), order [source clouds | target clouds]
    int* knn0[GMAX] = {};
    int* knn1[GMAX] = {};
    int n_self = 0;
    for (int i = 0; i < mdl.n_gnn; ++i) n_self += mdl.gnn[i].cross ? 0 : 1;
    const bool knn_once = n_self >= 2 && debug_opts().gnn_merge;
    if (knn_once)
        for (int g = 0; g < c.G; ++g) {
            const int k0 = mdl.knn_k < ns[g] - 1 ? mdl.knn_k : ns[g] - 1, k1 = mdl.knn_k < nt[g] - 1 ? mdl.knn_k : nt[g] - 1;
            knn0[g] = static_cast<int*>(c.raw(sizeof(int) * (size_t)ns[g] * (k0 > 0 ? k0 : 1)));
            knn1[g] = static_cast<int*>(c.raw(sizeof(int) * (size_t)nt[g] * (k1 > 0 ? k1 : 1)));
            if (c.live()) {
                c.check(pcrcg_knn(c0[g], ns[g], k0, knn0[g], c.st));
                c.check(pcrcg_knn(c1[g], nt[g], k1, knn1[g], c.st));
            }
        }
    for (int i = 0; i < mdl.n_gnn; ++i) {
        const pcrcg_gnn_layer& gl = mdl.gnn[i];
        if (gl.cross && 2 * c.G <= GMAX && c.paired_ok() && debug_opts().gnn_merge) {
            // the second direction's queries come from d1, which the first direction does not change: both query
            // projections in one launch (2 G products sharing wq)
            const int G0 = c.G, ch = d0.cols;
            const size_t mk = c.mark();
            Mat x2 = d0;
            for (int g = 0; g < G0; ++g) { x2.p[G0 + g] = d1.p[g]; x2.rows[G0 + g] = d1.rows[g]; }
            c.G = 2 * G0;
            Mat q2 = c.gemm_out(x2.rows, ch, ch);
            linear(c, x2, gl.wq, ch, gl.bq, q2);
            c.G = G0;
            Mat q0 = q2, q1 = q2;
            for (int g = 0; g < GMAX; ++g) {
                q0.p[g] = g < G0 ? q2.p[g] : nullptr; q0.rows[g] = g < G0 ? q2.rows[g] : 0;
                q1.p[g] = g < G0 ? q2.p[G0 + g] : nullptr; q1.rows[g] = g < G0 ? q2.rows[G0 + g] : 0;
            }
            // (the results are allocated above the projections: the arena is a stack, so they stay until the forward ends)
            d0 = cross_attention(c, mdl, gl, d0, d1, &q0);
            d1 = cross_attention(c, mdl, gl, d1, d0, &q1);   // sees the updated d0 (:214)
            (void)mk;
        } else if (gl.cross) {
            d0 = cross_attention(c, mdl, gl, d0, d1);
            d1 = cross_attention(c, mdl, gl, d1, d0);   // sees the updated d0 (:214)
        } else if (2 * c.G <= GMAX && c.paired_ok() && debug_opts().gnn_merge) {
            // The layer is applied to the source cloud and to the target cloud of every pair with the SAME weights and no
            // exchange between them (ref:models/gcn.py:207-211): 2 G independent clouds through ONE pass -- every weight
            // product of the layer once for all of them (round 5: three launches per layer instead of six; the
            // per-cloud kernels are unchanged)
            const int G0 = c.G;
            Mat f2;
            const float* cc[GMAX];
            int* kk[GMAX];
            f2.cols = d0.cols; f2.ld = d0.ld;
            for (int g = 0; g < G0; ++g) {
                f2.p[g] = d0.p[g]; f2.rows[g] = d0.rows[g]; cc[g] = c0[g]; kk[g] = knn0[g];
                f2.p[G0 + g] = d1.p[g]; f2.rows[G0 + g] = d1.rows[g]; cc[G0 + g] = c1[g]; kk[G0 + g] = knn1[g];
            }
            c.G = 2 * G0;
            const Mat y2 = self_attention(c, mdl, gl, cc, f2, knn_once ? kk : nullptr);
            c.G = G0;
            d0 = y2; d1 = y2;
            for (int g = 0; g < GMAX; ++g) {
                d0.p[g] = g < G0 ? y2.p[g] : nullptr; d0.rows[g] = g < G0 ? y2.rows[g] : 0;
                d1.p[g] = g < G0 ? y2.p[G0 + g] : nullptr; d1.rows[g] = g < G0 ? y2.rows[G0 + g] : 0;
            }
        } else {
            d0 = self_attention(c, mdl, gl, c0, d0, knn_once ? knn0 : nullptr);
            d1 = self_attention(c, mdl, gl, c1, d1, knn_once ? knn1 : nullptr);
        }
    }
    // coarse features [score | saliency | proj_gnn feats] (:538-565), rows 16-byte aligned
    const int wc = gd + 2;
    Mat xc = c.mat(nc, wc, pad4(wc));
    {
        const size_t m = c.mark();
        Mat gcat = c.mat(nc, gd), fn = c.mat(nc, gd);
        int st_rows[GMAX], ts_rows[GMAX];
        int nt_max = 0, ns_max = 0;
        for (int g = 0; g < GMAX; ++g) { st_rows[g] = ns[g]; ts_rows[g] = nt[g]; nt_max = nt[g] > nt_max ? nt[g] : nt_max; ns_max = ns[g] > ns_max ? ns[g] : ns_max; }
        Mat pst = c.mat(st_rows, nt_max), pts = c.mat(ts_rows, ns_max);
        if (c.live())
            for (int g = 0; g < c.G; ++g) {
                c.check(pcrcg_copy2d(d0.p[g], d0.ld, gcat.p[g], gcat.ld, ns[g], gd, c.st));
                c.check(pcrcg_copy2d(d1.p[g], d1.ld, gcat.p[g] + (long)ns[g] * gcat.ld, gcat.ld, nt[g], gd, c.st));
            }
        Mat feats = cols(xc, 2, gd), score = cols(xc, 0, 1), sal = cols(xc, 1, 1);
        linear(c, gcat, mdl.proj_gnn_w, gd, mdl.proj_gnn_b, feats);           // :538
        linear(c, feats, mdl.proj_score_w, gd, mdl.proj_score_b, score);      // :539
        if (c.live())
            for (int g = 0; g < c.G; ++g) {
                c.check(pcrcg_l2norm_rows(feats.p[g], feats.ld, fn.p[g], fn.ld, nc[g], gd, c.st));   // :541
                const float inv_t = 1.0f / mdl.temperature;
                const float* fs = fn.p[g];
                const float* ft = fn.p[g] + (long)ns[g] * fn.ld;
                // s1 = softmax(inner/T) @ tgt_scores, s2 = softmax(inner^T/T) @ src_scores  (:562-563); the softmax and
                // the product with the score column are one pass over each inner-product matrix
                c.check(pcrcg_gemm_f32(fs, fn.ld, ft, fn.ld, 1, pst.p[g], nt[g], ns[g], nt[g], gd, nullptr, nullptr, c.st));
                c.check(pcrcg_softmax_matvec(pst.p[g], ns[g], nt[g], nt[g], inv_t, score.p[g] + (long)ns[g] * xc.ld, xc.ld, sal.p[g],
                                             xc.ld, c.st));
                c.check(pcrcg_gemm_f32(ft, fn.ld, fs, fn.ld, 1, pts.p[g], ns[g], nt[g], ns[g], gd, nullptr, nullptr, c.st));
                c.check(pcrcg_softmax_matvec(pts.p[g], nt[g], ns[g], ns[g], inv_t, score.p[g], xc.ld, sal.p[g] + (long)ns[g] * xc.ld,
                                             xc.ld, c.st));
            }
        c.release(m);
    }
    x = xc;
    // 4. decoder (:567-570)
    const bool fuse_off = !debug_opts().fuse_upsample;
    // `x` may be LAZY between two decoder stages: the raw output of a unary's products whose InstanceNorm + LeakyReLU the
    // next stage's gathering product applies on load (xs = its column sums); materialise() applies it for anyone else
    Stat xs;
    bool lazy = false;
    auto materialise = [&]() {
        if (!lazy) return;
        Mat y = c.mat(x.rows, x.cols);
        norm_act(c, x, 0.1f, y, &xs);
        x = y;
        lazy = false;
    };
    for (int j = 0; j < mdl.n_dec; ++j) {
        const pcrcg_block& blk = mdl.dec[j];
        if (blk.type == PCRCG_BLK_UPSAMPLE) {
            const pcrcg_table* tabs[GMAX];
            int trows[GMAX] = {};
            for (int g = 0; g < c.G; ++g) { tabs[g] = &B.b[g]->upsamples[blk.layer - 1]; trows[g] = tabs[g]->rows; }
            const bool concat = j + 1 < mdl.n_dec && mdl.dec_concat[j + 1];
            const int cs = concat ? skips.back().cols : 0;
            const pcrcg_block* next = j + 1 < mdl.n_dec ? &mdl.dec[j + 1] : nullptr;
            // nearest_upsample -> cat(skip) -> unary as TWO products into one output: the unary's first columns on rows
            // of x gathered through the table (closest_pool, the shadow index reads a zero row), plus its skip columns on
            // the skip features.  Neither the upsampled matrix nor the concatenation is written: at level 0 that is
            // 31 + 61 MB of stores and 92 MB of loads that the product no longer waits for.
            if (!fuse_off && concat && next && next->mlp_skip && next->skip_dim == cs && gemm_extra_ok() &&
                (next->type == PCRCG_BLK_UNARY || next->type == PCRCG_BLK_LAST_UNARY) && x.ld % 4 == 0 &&
                skips.back().ld % 4 == 0) {
                const Mat& sk = skips.back();
                const bool last = next->type == PCRCG_BLK_LAST_UNARY;
                Mat tt = c.mat(trows, next->out_dim, pad4(next->out_dim));    // rows 16-byte aligned: it may be gathered next
                float* zero_row = static_cast<float*>(c.zraw(sizeof(float) * (size_t)(x.cols + 8)));
                if (lazy && !(norm_fuse_on() && lazy_stats_ready(c, x, &xs))) materialise();
                // the producer's normalisation (when lazy) applied to the gathered rows; then the skip part on top
                linear_extra(c, x, next->mlp, next->mlp_ld, nullptr, tt, 0, tabs, zero_row, lazy ? &xs : nullptr, 0.1f, false, false);
                linear_extra(c, sk, next->mlp_skip, next->mlp_skip_ld, nullptr, tt, 0, nullptr, nullptr, nullptr, 1.0f, true, true);
                skips.pop_back();
                lazy = false;
                x = tt;
                if (!last) {                                         // its normalisation is left to the next consumer
                    xs = stat_buffer(c, tt.rows, tt.cols);           // (filled by one pass over tt: two products wrote it)
                    lazy = true;
                }
                ++j;                                                 // the unary block is done
                continue;
            }
            materialise();
            Mat y = c.mat(trows, x.cols + cs, pad4(x.cols + cs));
            Mat xd = x;
            if (x.ld != x.cols) {   // gather_first reads dense rows
                xd = c.mat(x.rows, x.cols);
                if (c.live())
                    for (int g = 0; g < c.G; ++g) c.check(pcrcg_copy2d(x.p[g], x.ld, xd.p[g], xd.ld, x.rows[g], x.cols, c.st));
            }
            if (c.live())
                for (int g = 0; g < c.G; ++g) {
                    c.check(pcrcg_gather_first(xd.p[g], xd.rows[g], xd.cols, tabs[g]->idx, tabs[g]->rows, tabs[g]->ld, y.p[g], y.ld, c.st));
                    if (concat) {
                        const Mat& s = skips.back();
                        c.check(pcrcg_copy2d(s.p[g], s.ld, y.p[g] + x.cols, y.ld, s.rows[g], s.cols, c.st));
                    }
                }
            if (concat) skips.pop_back();
            x = y;
        } else if (blk.type == PCRCG_BLK_UNARY) {
            materialise();
            Mat t = c.gemm_out(x.rows, blk.out_dim, x.cols), y = c.mat(x.rows, blk.out_dim);
            Stat ts = stat_buffer(c, t.rows, t.cols);
            linear(c, x, blk.mlp, blk.mlp_ld, nullptr, t, &ts);
            norm_act(c, t, 0.1f, y, &ts);
            x = y;
        } else {   // last_unary
            materialise();
            Mat y = c.mat(x.rows, blk.out_dim, pad4(blk.out_dim));
            linear(c, x, blk.mlp, blk.mlp_ld, nullptr, y);
            x = y;
        }
    }
    materialise();
    // heads (:571-582)
    if (c.live()) {
        const int fd = mdl.final_dim;
        if (c.G <= 4) {           // the three heads of every pair in one launch
            float *ff[GMAX], *so[GMAX], *ss[GMAX];
            for (int g = 0; g < c.G; ++g) { ff[g] = out[g].feats_f; so[g] = out[g].scores_overlap; ss[g] = out[g].scores_saliency; }
            c.check(heads_multi(x.p, x.rows, c.G, x.ld, fd, ff, so, ss, c.st));
        } else
        for (int g = 0; g < c.G; ++g) {
            c.check(pcrcg_l2norm_rows(x.p[g], x.ld, out[g].feats_f, fd, x.rows[g], fd, c.st));
            c.check(pcrcg_sigmoid_scores(x.p[g] + fd, x.ld, out[g].scores_overlap, x.rows[g], c.st));
            c.check(pcrcg_sigmoid_scores(x.p[g] + fd + 1, x.ld, out[g].scores_saliency, x.rows[g], c.st));
        }
    }
}

int validate(const pcrcg_model* m, const pcrcg_batch* b) {
    PCRCG_CHECK_ARG(m && b);
    PCRCG_CHECK_ARG(m->n_enc >= 1 && m->n_enc <= PCRCG_MAX_BLOCKS && m->n_dec >= 1 && m->n_dec <= PCRCG_MAX_BLOCKS);
    PCRCG_CHECK_ARG(m->n_gnn >= 0 && m->n_gnn <= PCRCG_MAX_GNN);
    PCRCG_CHECK_ARG(b->n_levels >= 1 && b->n_levels <= PCRCG_MAX_LEVELS);
    PCRCG_CHECK_ARG(b->len_src_c >= 1 && b->len_src_c < b->n_points[b->n_levels - 1]);
    PCRCG_CHECK_ARG(m->heads >= 1 && m->gnn_dim % m->heads == 0 && m->temperature > 0.0f);
    for (int i = 0; i < m->n_enc; ++i) PCRCG_CHECK_ARG(m->enc[i].layer >= 0 && m->enc[i].layer + m->enc[i].strided < b->n_levels);
    return PCRCG_OK;
}

int validate_group(const pcrcg_model* m, const pcrcg_batch* b, int n) {
    PCRCG_CHECK_ARG(b && n >= 1 && n <= GMAX);
    for (int g = 0; g < n; ++g) {
        PCRCG_PROPAGATE(validate(m, b + g));
        PCRCG_CHECK_ARG(b[g].n_levels == b[0].n_levels && b[g].feat_dim == b[0].feat_dim);
    }
    return PCRCG_OK;
}

}  // namespace
}  // namespace pcrcg

using namespace pcrcg;

extern "C" {

size_t pcrcg_kpfcnn_group_ws_bytes(const pcrcg_model* model, const pcrcg_batch* batches, int n) {
    if (validate_group(model, batches, n) != PCRCG_OK) return 0;
    Ctx c;
    c.dry = true;
    c.G = n;
    Batches B;
    for (int g = 0; g < GMAX; ++g) B.b[g] = g < n ? batches + g : nullptr;
    forward(c, *model, B, nullptr);
    return c.peak + c.zoff + 4096;
}

size_t pcrcg_kpfcnn_ws_bytes(const pcrcg_model* model, const pcrcg_batch* batch) {
    return pcrcg_kpfcnn_group_ws_bytes(model, batch, 1);
}

int pcrcg_kpfcnn_forward_group(const pcrcg_model* model, const pcrcg_batch* batches, const pcrcg_outputs* outs, int n, void* ws,
                               size_t ws_bytes, void* stream) {
    PCRCG_PROPAGATE(validate_group(model, batches, n));
    PCRCG_CHECK_ARG(outs && ws);
    for (int g = 0; g < n; ++g) PCRCG_CHECK_ARG(outs[g].feats_f && outs[g].scores_overlap && outs[g].scores_saliency);
    Batches B;
    for (int g = 0; g < GMAX; ++g) B.b[g] = g < n ? batches + g : nullptr;
    // pass 1 (no launches): how much of the workspace the zero arena takes for these batches
    Ctx d;
    d.dry = true;
    d.G = n;
    forward(d, *model, B, nullptr);
    if (d.zoff + d.peak > ws_bytes) {
        set_error("pcrcg_kpfcnn_forward: workspace too small (%zu needed, %zu given)", d.zoff + d.peak, ws_bytes);
        return PCRCG_EWORKSPACE;
    }
    Ctx c;
    c.G = n;
    c.zbase = static_cast<char*>(ws);
    c.zcap = d.zoff;
    c.base = c.zbase + d.zoff;
    c.cap = ws_bytes - d.zoff;
    c.st = as_stream(stream);
    if (d.zoff > 0) PCRCG_CHECK_HIP(hipMemsetAsync(ws, 0, d.zoff, c.st));
    forward(c, *model, B, outs);
    return c.rc;
}

int pcrcg_kpfcnn_forward(const pcrcg_model* model, const pcrcg_batch* batch, const pcrcg_outputs* out, void* ws,
                         size_t ws_bytes, void* stream) {
    return pcrcg_kpfcnn_forward_group(model, batch, out, 1, ws, ws_bytes, stream);
}
}
