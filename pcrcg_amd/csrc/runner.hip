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
bool gemm_bt_accumulates(int m, int n, int k);
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
namespace {

struct Mat {          // row-major fp32 matrix view
    float* p = nullptr;
    int rows = 0, cols = 0, ld = 0;
    bool zeroed = false;   // lives in the zero arena and has not been written yet
};

struct Ctx {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    hipStream_t st = nullptr;
    bool dry = false;
    bool bf16 = false;      // pcrcg_model.feature_bf16
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
    Mat mat(int rows, int cols, int ld = 0) {
        Mat m;
        m.rows = rows; m.cols = cols; m.ld = ld ? ld : cols;
        m.p = static_cast<float*>(raw((size_t)(rows > 0 ? rows : 1) * m.ld * sizeof(float)));
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
    // the output of a [rows, k] x [cols, k]^T product
    Mat gemm_out(int rows, int cols, int k) {
        if (!debug_opts().zero_arena || !gemm_bt_accumulates(rows, cols, k)) return mat(rows, cols);
        Mat m;
        m.rows = rows; m.cols = cols; m.ld = cols;
        m.p = static_cast<float*>(zraw((size_t)(rows > 0 ? rows : 1) * cols * sizeof(float)));
        m.zeroed = true;
        return m;
    }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
    bool live() const { return !dry && rc == PCRCG_OK; }
    void check(int r) { if (r != PCRCG_OK && rc == PCRCG_OK) rc = r; }
};

inline int pad4(int v) { return (v + 3) & ~3; }
inline Mat cols(const Mat& m, int c0, int n) { Mat r = m; r.p = m.p ? m.p + c0 : nullptr; r.cols = n; return r; }
inline Mat rows(const Mat& m, int r0, int n) { Mat r = m; r.p = m.p ? m.p + (long)r0 * m.ld : nullptr; r.rows = n; return r; }

struct Stat;
// y = x @ w^T (+ bias); w is [out, in] with leading dimension ldw; optionally leaves the column
// partials of y for the InstanceNorm that follows
void linear(Ctx& c, const Mat& x, const float* w, int ldw, const float* bias, const Mat& y, Stat* st = nullptr);

// A GEMM output together with the InstanceNorm column partials its epilogue may have produced.
struct Stat {
    void* partials = nullptr;   // [2][cols][chunks] fp64 partials, valid when chunks > 0; or, when `sums`, zeroed
                                // [2][cols] fp64 accumulators that hold the column sums when chunks == -1
    size_t bytes = 0;
    int chunks = 0;
    bool sums = false;
};

// The GEMM epilogue adds its output's column sums into accumulators from the zero arena with fp64 atomics -- one pair per
// column and 64-row tile, the workgroup's two row halves meet in LDS first -- and the normalisation derives mean / rstd
// from them itself: no partial buffers, no finishing launch (51 per S30k forward before).  With the per-wavefront
// atomics of the first version the 60 000-row outputs (1 876 per address) measured slower than partials + finishing
// kernel and kept those; with one per tile (938) the sums win everywhere: 470-473 vs 457-466 pairs/s.
// (DebugOpts::stat_sums_rows: outputs above that many rows keep the deterministic partials.)
Stat stat_buffer(Ctx& c, int rows, int cols) {
    Stat s;
    if (debug_opts().stat_sums && rows <= debug_opts().stat_sums_rows && gemm_colstats_sums_ok()) {
        s.sums = true;
        s.bytes = 2 * sizeof(double) * (size_t)cols;
        s.partials = c.zraw(s.bytes);
        return s;
    }
    s.bytes = pcrcg_gemm_colstats_bytes(rows, cols);
    s.partials = c.raw(s.bytes);
    return s;
}

// widths the sums form of the normalisation kernel serves (its thread -> channel-group map)
inline bool sums_apply_ok(const Mat& x, const Mat& y, const Mat* res) {
    const int c4 = x.cols / 4;
    return x.cols % 4 == 0 && x.cols >= 4 && (c4 <= 256 ? 256 % c4 == 0 : c4 % 256 == 0) && x.ld % 4 == 0 && y.ld % 4 == 0 &&
           (!res || res->ld % 4 == 0);
}

// (mean, rstd) of x: from the producing GEMM's partials when it left some, else by a pass over x
void col_stats(Ctx& c, const Mat& x, const Stat* s, float* stats, void* ws, size_t wsb) {
    if (s && s->chunks == -1)      // column sums: the finishing kernel reads them as one chunk per column
        c.check(pcrcg_instnorm_stats_from_partials(s->partials, 1, x.cols, (double)x.rows, 1e-5f, stats, c.st));
    else if (s && s->chunks > 0)
        c.check(pcrcg_instnorm_stats_from_partials(s->partials, s->chunks, x.cols, (double)x.rows, 1e-5f, stats, c.st));
    else
        c.check(pcrcg_instnorm_stats(x.p, x.rows, x.cols, x.ld, 1e-5f, stats, ws, wsb, c.st));
}

// y = lrelu(IN(x) [+ IN(res) | + res], slope)
void norm_act(Ctx& c, const Mat& x, float slope, const Mat& y, Stat* xs = nullptr, const Mat* res = nullptr,
              bool norm_res = false, Stat* rs = nullptr) {
    // sums-mode statistics the producing GEMM could not leave (a split-K product): one pass into the same accumulators
    if (c.live() && xs && xs->sums && xs->chunks == 0) {
        c.check(pcrcg_instnorm_colsums(x.p, x.rows, x.cols, x.ld, xs->partials, c.st));
        xs->chunks = -1;
    }
    if (c.live() && res && norm_res && rs && rs->sums && rs->chunks == 0) {
        c.check(pcrcg_instnorm_colsums(res->p, res->rows, res->cols, res->ld, rs->partials, c.st));
        rs->chunks = -1;
    }
    if (c.live() && xs && xs->chunks == -1 && (!res || !norm_res || (rs && rs->chunks == -1)) && sums_apply_ok(x, y, res)) {
        c.check(pcrcg_instnorm_apply_sums(x.p, x.rows, x.cols, x.ld, xs->partials, (double)x.rows, 1e-5f, res ? res->p : nullptr,
                                          res ? res->ld : 0, (res && norm_res) ? rs->partials : nullptr, slope, y.p, y.ld,
                                          c.st));
        return;
    }
    const size_t m = c.mark();
    float* stats = static_cast<float*>(c.raw(sizeof(float) * 2 * x.cols));
    float* rstats = (res && norm_res) ? static_cast<float*>(c.raw(sizeof(float) * 2 * x.cols)) : nullptr;
    const size_t wsb = pcrcg_instnorm_ws_bytes(x.cols);
    void* ws = c.raw(wsb);
    if (c.live()) {
        col_stats(c, x, xs, stats, ws, wsb);
        if (rstats) col_stats(c, *res, rs, rstats, ws, wsb);
        c.check(pcrcg_instnorm_apply(x.p, x.rows, x.cols, x.ld, stats, res ? res->p : nullptr, res ? res->ld : 0,
                                     rstats, slope, y.p, y.ld, c.st));
    }
    c.release(m);
}

void linear(Ctx& c, const Mat& x, const float* w, int ldw, const float* bias, const Mat& y, Stat* st) {
    if (!c.live()) return;
    c.check(gemm_bt_colstats(x.p, x.ld, w, ldw, y.p, y.ld, x.rows, y.cols, x.cols, nullptr, bias,
                             st ? st->partials : nullptr, st ? st->bytes : 0, st ? &st->chunks : nullptr, c.st, y.zeroed,
                             st && st->sums));
}

// packed_ws != NULL: a pcrcg_kpconv_ws_bytes(ns) workspace whose support records the producer of x has already filled
// (norm_act_pack): the aggregate kernel starts without the row-positive pass
// y = lrelu(IN(x), slope) @ w^T (+ bias) with the normalisation done inside the product's A loads (GemmExtra::a_sums):
// x is the RAW output of the producing product and xs its column sums.  Returns false when that form does not apply
// (the caller then normalises into a matrix of its own and calls linear()).
bool norm_fuse_on() {
    return debug_opts().fuse_norm && gemm_extra_ok();
}
bool lazy_stats_ready(Ctx& c, const Mat& x, Stat* xs) {
    if (!xs || !xs->sums || x.ld % 4 != 0 || x.cols > 4096) return false;
    if (c.live() && xs->chunks == 0) {               // nobody left the sums yet (split-K or accumulated output): one pass
        c.check(pcrcg_instnorm_colsums(x.p, x.rows, x.cols, x.ld, xs->partials, c.st));
        xs->chunks = -1;
    }
    return true;
}
bool linear_norm(Ctx& c, const Mat& x, Stat* xs, float slope, const float* w, int ldw, const float* bias, const Mat& y,
                 Stat* st = nullptr) {
    if (!norm_fuse_on() || !lazy_stats_ready(c, x, xs)) return false;
    if (!c.live()) return true;
    GemmExtra ex;
    ex.a_sums = static_cast<const double*>(xs->partials);
    ex.a_count = (double)x.rows;
    ex.a_slope = slope;
    c.check(gemm_bt_extra(x.p, x.ld, w, ldw, y.p, y.ld, x.rows, y.cols, x.cols, c.st, y.zeroed, ex, bias,
                          st ? st->partials : nullptr, st ? st->bytes : 0, st ? &st->chunks : nullptr, st && st->sums));
    return true;
}

void kpconv(Ctx& c, const pcrcg_batch& b, const pcrcg_block& blk, const Mat& x, const Mat& y, Stat* st = nullptr,
            void* packed_ws = nullptr) {
    const int l = blk.layer;
    const pcrcg_table& t = blk.strided ? b.pools[l] : b.neighbors[l];
    const float* q = blk.strided ? b.points[l + 1] : b.points[l];
    const int nq = blk.strided ? b.n_points[l + 1] : b.n_points[l];
    const int ns = b.n_points[l];
    const size_t m = c.mark();
    // channel counts that are not a multiple of 4 (the 129-channel PCR-CG input): zero-padded copy of the
    // features + zero-padded weights, so that the MFMA gather kernel applies (zeros change neither the sums nor
    // the neighbour count of the normaliser)
    const float* xp = x.p;
    const float* w = blk.kp_w;
    int cin = x.cols;
    if (blk.kp_w_pad && blk.cin_pad > x.cols) {
        cin = blk.cin_pad;
        Mat pad = c.mat(x.rows, cin);
        if (c.live()) {
            c.check(hipMemsetAsync(pad.p, 0, sizeof(float) * (size_t)x.rows * cin, c.st) == hipSuccess ? PCRCG_OK
                                                                                                       : PCRCG_ELAUNCH);
            c.check(pcrcg_copy2d(x.p, x.ld, pad.p, cin, x.rows, x.cols, c.st));
        }
        xp = pad.p;
        w = blk.kp_w_pad;
    }
    float* inv_n = static_cast<float*>(c.raw(sizeof(float) * (nq > 0 ? nq : 1)));
    const size_t wsb = pcrcg_kpconv_ws_bytes(ns);
    void* ws = packed_ws ? packed_ws : c.raw(wsb);
    // bf16 feature storage (pcrcg_model.feature_bf16): the gathers read a bf16 copy of x and wf is bf16 in HBM -- half
    // the bytes of the two streams that bound the encoder; the contraction takes wf as the (single-term) bf16 operand
    // against the exact three-term split of the fp32 weights, fp32 accumulate and fp32 output.
    if (c.bf16 && blk.kp_wt && cin % 32 == 0) {
        const int kk = PCRCG_KPOINTS * cin;
        void* xb = c.raw(sizeof(unsigned short) * (size_t)(ns > 0 ? ns : 1) * cin);
        void* wfb = c.raw(sizeof(unsigned short) * (size_t)(nq > 0 ? nq : 1) * kk);
        if (c.live()) {
            c.check(pcrcg_kpconv_aggregate_bf16(q, nq, b.points[l], ns, t.idx, t.cols, t.ld, xp, cin, blk.kp, blk.extent,
                                                xb, wfb, inv_n, ws, wsb, c.st));
            c.check(gemm_bf16a_bt_colstats(wfb, kk, blk.kp_wt, kk, y.p, y.ld, nq, y.cols, kk, inv_n, nullptr,
                                           st ? st->partials : nullptr, st ? st->bytes : 0, st ? &st->chunks : nullptr,
                                           c.st, y.zeroed, st && st->sums));
        }
        c.release(m);
        return;
    }
    Mat wf = c.mat(nq, PCRCG_KPOINTS * cin);
    if (c.live()) {
        if (packed_ws && xp == x.p)
            c.check(kpconv_aggregate_rows(q, nq, b.points[l], ns, t.idx, t.cols, t.ld, xp, cin, blk.kp, blk.extent, wf.p, inv_n,
                                          ws, wsb, c.st, /*pack=*/false, /*stream_out=*/true));
        else
            c.check(pcrcg_kpconv_aggregate(q, nq, b.points[l], ns, t.idx, t.cols, t.ld, xp, cin, blk.kp, blk.extent,
                                           wf.p, inv_n, ws, wsb, c.st));
        // contraction wf @ W: against the K-contiguous copy wt [cout, 15*cin] when the descriptor carries one
        // (C = A * B^T form: both operands k-contiguous, the form the split-bf16 GEMM is built for).
        // (Measured and not adopted, rounds 2 and 3: aggregating + contracting row chunks so that wf stays in L2 /
        // Infinity Cache between the two kernels -- isolated 3.50 vs 3.31 ms per forward with 48 MB chunks; inside the
        // engine 416 / 446 / 459 / 462 pairs/s with 16 / 32 / 64 / 120 MB chunks against 460-464 unchunked: the extra
        // launches cost more than the on-chip re-read saves.)
        if (blk.kp_wt)
            c.check(gemm_bt_colstats(wf.p, wf.ld, blk.kp_wt, wf.cols, y.p, y.ld, nq, y.cols, wf.cols, inv_n, nullptr,
                                     st ? st->partials : nullptr, st ? st->bytes : 0, st ? &st->chunks : nullptr, c.st,
                                     y.zeroed, st && st->sums));
        else   // (descriptor without the K-contiguous weight copy: the plain entry point knows only the partials layout)
            c.check(pcrcg_gemm_f32_colstats(wf.p, wf.ld, w, y.cols, 0, y.p, y.ld, nq, y.cols, wf.cols, inv_n, nullptr,
                                            (st && !st->sums) ? st->partials : nullptr, (st && !st->sums) ? st->bytes : 0,
                                            (st && !st->sums) ? &st->chunks : nullptr, c.st));
    }
    c.release(m);
}

// input channels of the block's KPConv as the gather kernel sees them (padded to a multiple of 4)
int kp_cin(const pcrcg_block& blk, const Mat& x) { return (blk.kp_w_pad && blk.cin_pad > x.cols) ? blk.cin_pad : x.cols; }

int out_rows(const pcrcg_batch& b, const pcrcg_block& blk) {
    return blk.strided ? b.n_points[blk.layer + 1] : b.n_points[blk.layer];
}

// SimpleBlock.forward (ref:models/blocks.py:578-590)
Mat simple_block(Ctx& c, const pcrcg_batch& b, const pcrcg_block& blk, const Mat& x) {
    Mat y = c.mat(out_rows(b, blk), blk.mid_dim);
    const size_t m = c.mark();
    Mat t = c.gemm_out(y.rows, y.cols, PCRCG_KPOINTS * kp_cin(blk, x));
    Stat ts = stat_buffer(c, t.rows, t.cols);
    kpconv(c, b, blk, x, t, &ts);
    norm_act(c, t, 0.1f, y, &ts);
    c.release(m);
    return y;
}

// u = lrelu(IN(t), slope) for a u that is the input of the block's KPConv: the normalisation kernel leaves the KPConv's
// support records (coordinates + "feature row sums to a positive value" flag) in kp_ws as it goes, so the KPConv needs no
// pass over u of its own (10 launches per S30k forward: +4 % pairs/s with them knocked out).  False: not applicable.
bool norm_act_pack(Ctx& c, const Mat& t, float slope, const Mat& u, Stat* ts, const float* s_pts, void* kp_ws,
                   size_t kp_ws_bytes) {
    if (!debug_opts().fuse_pack || c.bf16 || !instnorm_pack_ok(t.cols, t.ld, u.ld) || u.ld != u.cols) return false;
    const size_t m = c.mark();
    float* stats = static_cast<float*>(c.raw(sizeof(float) * 2 * t.cols));
    const size_t wsb = pcrcg_instnorm_ws_bytes(t.cols);
    void* ws = c.raw(wsb);
    if (c.live()) {
        if (ts && ts->sums && ts->chunks == 0) {        // a split-K product left nothing: one pass into the accumulators
            c.check(pcrcg_instnorm_colsums(t.p, t.rows, t.cols, t.ld, ts->partials, c.st));
            ts->chunks = -1;
        }
        float4* pk = kpconv_pk_ptr(kp_ws, kp_ws_bytes, t.rows);
        if (!pk) c.check(PCRCG_EWORKSPACE);
        else if (ts && ts->chunks == -1)
            c.check(instnorm_apply_pack(t.p, t.rows, t.cols, t.ld, nullptr, static_cast<const double*>(ts->partials),
                                        (double)t.rows, 1e-5f, slope, u.p, u.ld, s_pts, pk, c.st));
        else {
            col_stats(c, t, ts, stats, ws, wsb);
            c.check(instnorm_apply_pack(t.p, t.rows, t.cols, t.ld, stats, nullptr, 0.0, 1e-5f, slope, u.p, u.ld, s_pts, pk, c.st));
        }
    }
    c.release(m);
    return true;
}

// ResnetBottleneckBlock.forward (ref:models/blocks.py:650-678)
Mat resnet_block(Ctx& c, const pcrcg_batch& b, const pcrcg_block& blk, const Mat& feats) {
    const int nq = out_rows(b, blk);
    Mat y = c.mat(nq, blk.out_dim);
    const size_t m = c.mark();
    Mat x = feats;
    void* kp_ws = nullptr;
    size_t kp_wsb = 0;
    if (blk.unary1) {
        Mat t = c.gemm_out(feats.rows, blk.mid_dim, feats.cols), u = c.mat(feats.rows, blk.mid_dim);
        Stat ts = stat_buffer(c, t.rows, t.cols);
        linear(c, feats, blk.unary1, feats.cols, nullptr, t, &ts);
        // u feeds the KPConv whose supports are this block's input rows: pack its support records on the way
        if (blk.mid_dim % 4 == 0 && !(blk.kp_w_pad && blk.cin_pad > blk.mid_dim)) {
            kp_wsb = pcrcg_kpconv_ws_bytes(feats.rows);
            kp_ws = c.raw(kp_wsb);
            if (!norm_act_pack(c, t, 0.1f, u, &ts, b.points[blk.layer], kp_ws, kp_wsb)) kp_ws = nullptr;
        }
        if (!kp_ws) norm_act(c, t, 0.1f, u, &ts);
        x = u;
    }
    Mat k = c.gemm_out(nq, blk.mid_dim, PCRCG_KPOINTS * kp_cin(blk, x)), kn = c.mat(nq, blk.mid_dim);
    Stat ks = stat_buffer(c, nq, blk.mid_dim);
    kpconv(c, b, blk, x, k, &ks, kp_ws);
    Mat u2 = c.gemm_out(nq, blk.out_dim, blk.mid_dim);
    Stat u2s = stat_buffer(c, nq, blk.out_dim);
    // lrelu(IN(k)) is read by unary2 alone: its product normalises k on load when k's statistics are column sums
    if (!linear_norm(c, k, &ks, 0.1f, blk.unary2, blk.mid_dim, nullptr, u2, &u2s)) {
        norm_act(c, k, 0.1f, kn, &ks);
        linear(c, kn, blk.unary2, blk.mid_dim, nullptr, u2, &u2s);
    }
    Mat sc = feats;
    if (blk.strided) {   // max_pool shortcut (:672-673)
        const pcrcg_table& t = b.pools[blk.layer];
        sc = c.mat(nq, feats.cols);
        if (c.live())
            c.check(pcrcg_gather_max(feats.p, feats.rows, feats.cols, t.idx, nq, t.cols, t.ld, sc.p, c.st));
    }
    if (blk.shortcut) {
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
void edge_norm(Ctx& c, const Mat& cn, int cw, const int* idx, int n, int k, const Mat& emax, const Mat& out, void* sums,
               float* stats, void* ws, size_t wsb) {
    if (sums && sums_apply_ok(emax, out, nullptr)) {
        c.check(pcrcg_edgeconv_reduce_sums(cn.p, cn.ld, cn.p + cw, cn.ld, idx, n, k, cw, emax.p, emax.ld, sums, c.st));
        c.check(pcrcg_instnorm_apply_sums(emax.p, n, cw, emax.ld, sums, (double)n * (double)k, 1e-5f, nullptr, 0, nullptr, 0.2f,
                                          out.p, out.ld, c.st));
        return;
    }
    c.check(pcrcg_edgeconv_reduce(cn.p, cn.ld, cn.p + cw, cn.ld, idx, n, k, cw, 1e-5f, emax.p, emax.ld, stats, ws, wsb, c.st));
    c.check(pcrcg_instnorm_apply(emax.p, n, cw, emax.ld, stats, nullptr, 0, nullptr, 0.2f, out.p, out.ld, c.st));
}

// SelfAttention.forward (ref:models/gcn.py:110-134) on row-major [n, ch]
Mat self_attention(Ctx& c, const pcrcg_model& mdl, const pcrcg_gnn_layer& g, const float* coords, const Mat& f) {
    const int n = f.rows, ch = f.cols;
    Mat y = c.mat(n, ch);
    const size_t m = c.mark();
    const int k = mdl.knn_k < n - 1 ? mdl.knn_k : n - 1;
    int* idx = static_cast<int*>(c.raw(sizeof(int) * (size_t)n * (k > 0 ? k : 1)));
    Mat cat = c.mat(n, 4 * ch);
    const size_t wsb = pcrcg_edgeconv_ws_bytes(2 * ch);
    void* ws = c.raw(wsb);
    float* stats = static_cast<float*>(c.raw(sizeof(float) * 4 * ch));
    Mat cn1 = c.gemm_out(n, 2 * ch, ch), e1 = c.mat(n, ch), cn2 = c.gemm_out(n, 4 * ch, ch), e2 = c.mat(n, 2 * ch);
    Mat x3 = c.gemm_out(n, ch, 4 * ch);
    Stat x3s = stat_buffer(c, n, ch);
    // InstanceNorm2d statistics of the two edge convolutions as fp64 sums (zero arena) when that form applies
    void* sums1 = gemm_colstats_sums_ok() ? c.zraw(2 * sizeof(double) * (size_t)ch) : nullptr;
    void* sums2 = gemm_colstats_sums_ok() ? c.zraw(2 * sizeof(double) * (size_t)(2 * ch)) : nullptr;
    if (c.live()) {
        c.check(pcrcg_knn(coords, n, k, idx, c.st));
        c.check(pcrcg_copy2d(f.p, f.ld, cat.p, cat.ld, n, ch, c.st));                                    // x0
        // x1 = max_k lrelu(IN2d(conv1(cat(f_i, f_j - f_i))))  (:121-125)
        c.check(gemm_bt_colstats(f.p, f.ld, g.edge1, ch, cn1.p, cn1.ld, n, 2 * ch, ch, nullptr, nullptr, nullptr, 0, nullptr,
                                 c.st, cn1.zeroed));
        edge_norm(c, cn1, ch, idx, n, k, e1, cols(cat, ch, ch), sums1, stats, ws, wsb);
        // x2 from x1 with conv2 (:127-129)
        c.check(gemm_bt_colstats(cat.p + ch, cat.ld, g.edge2, ch, cn2.p, cn2.ld, n, 4 * ch, ch, nullptr, nullptr, nullptr, 0,
                                 nullptr, c.st, cn2.zeroed));
        edge_norm(c, cn2, 2 * ch, idx, n, k, e2, cols(cat, 2 * ch, 2 * ch), sums2, stats, ws, wsb);
        // x3 = lrelu(IN(conv3(cat(x0,x1,x2))))  (:131-132)
        c.check(gemm_bt_colstats(cat.p, cat.ld, g.conv3, 4 * ch, x3.p, x3.ld, n, ch, 4 * ch, nullptr, nullptr, x3s.partials,
                                 x3s.bytes, &x3s.chunks, c.st, x3.zeroed, x3s.sums));
    }
    norm_act(c, x3, 0.2f, y, &x3s);
    c.release(m);
    return y;
}

// x + AttentionalPropagation(x, src)  (ref:models/gcn.py:151-185, 213-214)
Mat cross_attention(Ctx& c, const pcrcg_model& mdl, const pcrcg_gnn_layer& g, const Mat& x, const Mat& src) {
    const int n = x.rows, ms = src.rows, ch = x.cols, h = mdl.heads, d = ch / h;
    Mat y = c.mat(n, ch);
    const size_t m = c.mark();
    Mat q = c.gemm_out(n, ch, ch), kk = c.gemm_out(ms, ch, ch), v = c.gemm_out(ms, ch, ch), msg = c.mat(n, ch);
    Mat sc = c.mat(n, ms);
    Mat cat = c.mat(n, 2 * ch), h0 = c.gemm_out(n, 2 * ch, 2 * ch), h1 = c.mat(n, 2 * ch), delta = c.gemm_out(n, ch, 2 * ch);
    linear(c, x, g.wq, ch, g.bq, q);
    linear(c, src, g.wk, ch, g.bk, kk);
    linear(c, src, g.wv, ch, g.bv, v);
    if (c.live() && pcrcg_attention_supported(d) && (long)n * ms <= 1000000) {
        // all heads in one launch (heads are contiguous column blocks after the weight permutation); a latency win on
        // the few hundred coarse points of an indoor pair (49 vs 138 us of kernels + 11 launches less per call,
        // scripts/attention_bench.py) -- beyond ~1000 x 1000 the GEMM path below is faster (1900 x 1900: 228 vs 640 us)
        c.check(pcrcg_attention(q.p, q.ld, kk.p, kk.ld, v.p, v.ld, msg.p, msg.ld, n, ms, h, d, 1.0f / sqrtf((float)d),
                                c.st));
        c.check(pcrcg_copy2d(x.p, x.ld, cat.p, cat.ld, n, ch, c.st));
    } else if (c.live()) {
        for (int i = 0; i < h; ++i) {
            c.check(pcrcg_gemm_f32(q.p + i * d, q.ld, kk.p + i * d, kk.ld, 1, sc.p, sc.ld, n, ms, d, nullptr, nullptr,
                                   c.st));
            c.check(pcrcg_softmax_rows(sc.p, n, ms, sc.ld, 1.0f / sqrtf((float)d), c.st));
            c.check(pcrcg_gemm_f32(sc.p, sc.ld, v.p + i * d, v.ld, 0, msg.p + i * d, msg.ld, n, d, ms, nullptr, nullptr,
                                   c.st));
        }
        c.check(pcrcg_copy2d(x.p, x.ld, cat.p, cat.ld, n, ch, c.st));
    }
    linear(c, msg, g.wm, ch, g.bm, cols(cat, ch, ch));       // merge, written next to x: cat([x, message])
    Stat h0s = stat_buffer(c, n, 2 * ch);
    linear(c, cat, g.w0, 2 * ch, g.b0, h0, &h0s);
    if (!linear_norm(c, h0, &h0s, 0.0f, g.w3, 2 * ch, g.b3, delta)) {      // InstanceNorm1d + ReLU, inside w3's A loads
        norm_act(c, h0, 0.0f, h1, &h0s);
        linear(c, h1, g.w3, 2 * ch, g.b3, delta);
    }
    if (c.live()) c.check(pcrcg_add(x.p, delta.p, y.p, (long)n * ch, c.st));
    c.release(m);
    return y;
}

void forward(Ctx& c, const pcrcg_model& mdl, const pcrcg_batch& b, const pcrcg_outputs& out) {
    const int L = b.n_levels;
    c.bf16 = mdl.feature_bf16 != 0;
    Mat x;
    x.p = const_cast<float*>(b.features);
    x.rows = b.n_points[0];
    x.cols = x.ld = b.feat_dim;
    std::vector<Mat> skips;
    // 1. encoder (:519-524)
    for (int i = 0; i < mdl.n_enc; ++i) {
        if (mdl.enc_skip[i]) skips.push_back(x);
        const pcrcg_block& blk = mdl.enc[i];
        x = blk.type == PCRCG_BLK_SIMPLE ? simple_block(c, b, blk, x) : resnet_block(c, b, blk, x);
    }
    // 2. bottleneck (:527-528) and 3. GNN (:532-536)
    const int nc = b.n_points[L - 1], ns = b.len_src_c, nt = nc - ns, g = mdl.gnn_dim;
    Mat fc = c.gemm_out(nc, g, mdl.enc_out_dim);
    linear(c, x, mdl.bottle_w, mdl.enc_out_dim, mdl.bottle_b, fc);
    Mat d0 = rows(fc, 0, ns), d1 = rows(fc, ns, nt);
    const float* c0 = b.points[L - 1];
    const float* c1 = b.points[L - 1] + 3 * (long)ns;
    for (int i = 0; i < mdl.n_gnn; ++i) {
        const pcrcg_gnn_layer& gl = mdl.gnn[i];
        if (gl.cross) {
            d0 = cross_attention(c, mdl, gl, d0, d1);
            d1 = cross_attention(c, mdl, gl, d1, d0);   // sees the updated d0 (:214)
        } else {
            d0 = self_attention(c, mdl, gl, c0, d0);
            d1 = self_attention(c, mdl, gl, c1, d1);
        }
    }
    // coarse features [score | saliency | proj_gnn feats] (:538-565), rows 16-byte aligned
    const int wc = g + 2;
    Mat xc = c.mat(nc, wc, pad4(wc));
    {
        const size_t m = c.mark();
        Mat gcat = c.mat(nc, g), fn = c.mat(nc, g), pst = c.mat(ns, nt), pts = c.mat(nt, ns);
        if (c.live()) {
            c.check(pcrcg_copy2d(d0.p, d0.ld, gcat.p, gcat.ld, ns, g, c.st));
            c.check(pcrcg_copy2d(d1.p, d1.ld, gcat.p + (long)ns * gcat.ld, gcat.ld, nt, g, c.st));
        }
        Mat feats = cols(xc, 2, g), score = cols(xc, 0, 1), sal = cols(xc, 1, 1);
        linear(c, gcat, mdl.proj_gnn_w, g, mdl.proj_gnn_b, feats);           // :538
        linear(c, feats, mdl.proj_score_w, g, mdl.proj_score_b, score);      // :539
        if (c.live()) {
            c.check(pcrcg_l2norm_rows(feats.p, feats.ld, fn.p, fn.ld, nc, g, c.st));   // :541
            const float inv_t = 1.0f / mdl.temperature;
            const float* fs = fn.p;
            const float* ft = fn.p + (long)ns * fn.ld;
            // s1 = softmax(inner/T) @ tgt_scores, s2 = softmax(inner^T/T) @ src_scores  (:562-563); the softmax and
            // the product with the score column are one pass over each inner-product matrix
            c.check(pcrcg_gemm_f32(fs, fn.ld, ft, fn.ld, 1, pst.p, pst.ld, ns, nt, g, nullptr, nullptr, c.st));
            c.check(pcrcg_softmax_matvec(pst.p, ns, nt, pst.ld, inv_t, score.p + (long)ns * xc.ld, xc.ld, sal.p, xc.ld, c.st));
            c.check(pcrcg_gemm_f32(ft, fn.ld, fs, fn.ld, 1, pts.p, pts.ld, nt, ns, g, nullptr, nullptr, c.st));
            c.check(pcrcg_softmax_matvec(pts.p, nt, ns, pts.ld, inv_t, score.p, xc.ld, sal.p + (long)ns * xc.ld, xc.ld, c.st));
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
            const pcrcg_table& t = b.upsamples[blk.layer - 1];
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
                Mat tt = c.mat(t.rows, next->out_dim, pad4(next->out_dim));    // rows 16-byte aligned: it may be gathered next
                float* zero_row = static_cast<float*>(c.zraw(sizeof(float) * (size_t)(x.cols + 8)));
                if (lazy && !(norm_fuse_on() && lazy_stats_ready(c, x, &xs))) materialise();
                if (c.live()) {
                    GemmExtra g1, g2;
                    if (lazy) {                              // the producer's normalisation, applied to the gathered rows
                        g1.a_sums = static_cast<const double*>(xs.partials);
                        g1.a_count = (double)x.rows;
                        g1.a_slope = 0.1f;
                    }
                    g1.a_idx = reinterpret_cast<const long long*>(t.idx);
                    g1.a_idx_ld = t.ld;
                    g1.a_ns = x.rows;
                    g1.a_zero = zero_row;
                    g2.accumulate = true;
                    c.check(gemm_bt_extra(x.p, x.ld, next->mlp, next->mlp_ld, tt.p, tt.ld, t.rows, next->out_dim, x.cols, c.st,
                                          false, g1));
                    c.check(gemm_bt_extra(sk.p, sk.ld, next->mlp_skip, next->mlp_skip_ld, tt.p, tt.ld, t.rows, next->out_dim,
                                          cs, c.st, true, g2));
                }
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
            Mat y = c.mat(t.rows, x.cols + cs, pad4(x.cols + cs));
            Mat xd = x;
            if (x.ld != x.cols) {   // gather_first reads dense rows
                xd = c.mat(x.rows, x.cols);
                if (c.live()) c.check(pcrcg_copy2d(x.p, x.ld, xd.p, xd.ld, x.rows, x.cols, c.st));
            }
            if (c.live()) {
                c.check(pcrcg_gather_first(xd.p, xd.rows, xd.cols, t.idx, t.rows, t.ld, y.p, y.ld, c.st));
                if (concat) {
                    const Mat& s = skips.back();
                    c.check(pcrcg_copy2d(s.p, s.ld, y.p + x.cols, y.ld, s.rows, s.cols, c.st));
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
        c.check(pcrcg_l2norm_rows(x.p, x.ld, out.feats_f, fd, x.rows, fd, c.st));
        c.check(pcrcg_sigmoid_scores(x.p + fd, x.ld, out.scores_overlap, x.rows, c.st));
        c.check(pcrcg_sigmoid_scores(x.p + fd + 1, x.ld, out.scores_saliency, x.rows, c.st));
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

}  // namespace
}  // namespace pcrcg

using namespace pcrcg;

extern "C" {

size_t pcrcg_kpfcnn_ws_bytes(const pcrcg_model* model, const pcrcg_batch* batch) {
    if (validate(model, batch) != PCRCG_OK) return 0;
    Ctx c;
    c.dry = true;
    pcrcg_outputs none = {nullptr, nullptr, nullptr};
    forward(c, *model, *batch, none);
    return c.peak + c.zoff + 4096;
}

int pcrcg_kpfcnn_forward(const pcrcg_model* model, const pcrcg_batch* batch, const pcrcg_outputs* out, void* ws,
                         size_t ws_bytes, void* stream) {
    PCRCG_PROPAGATE(validate(model, batch));
    PCRCG_CHECK_ARG(out && out->feats_f && out->scores_overlap && out->scores_saliency && ws);
    // pass 1 (no launches): how much of the workspace the zero arena takes for this batch
    Ctx d;
    d.dry = true;
    pcrcg_outputs none = {nullptr, nullptr, nullptr};
    forward(d, *model, *batch, none);
    if (d.zoff + d.peak > ws_bytes) {
        set_error("pcrcg_kpfcnn_forward: workspace too small (%zu needed, %zu given)", d.zoff + d.peak, ws_bytes);
        return PCRCG_EWORKSPACE;
    }
    Ctx c;
    c.zbase = static_cast<char*>(ws);
    c.zcap = d.zoff;
    c.base = c.zbase + d.zoff;
    c.cap = ws_bytes - d.zoff;
    c.st = as_stream(stream);
    if (d.zoff > 0) PCRCG_CHECK_HIP(hipMemsetAsync(ws, 0, d.zoff, c.st));
    forward(c, *model, *batch, *out);
    return c.rc;
}
}
