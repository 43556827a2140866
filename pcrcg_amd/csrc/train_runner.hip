// train_runner.hip -- host-side runner of the TRAIN step's network part: KPFCNN.forward with a tape
// and its backward, each enqueued by ONE C-ABI call (ref:lib/trainer.py:216-265 drives ref:models/architectures.py:181-191,
// 516-610 and torch.autograd through it).  Counterpart of runner.hip for training: no device code of its own, it sequences
// the library's forward kernels and the backward kernels of trainops.hip / the transposed forms of the split-bf16 GEMM.
// Round 2 ran this composition op by op from Python under torch.autograd: 5.4 ms of interpreter time for the forward,
// ~1 070 at::native launches per step (zero fills, gradient accumulation adds, concatenations) and 8.7 ms of host time
// for the backward.  Here the forward records, per operator, a closure that enqueues its backward; activations and
// gradients live in one caller-provided workspace:
//
//   [ values | gradients | backward scratch ]
//
// every differentiable tensor gets a value buffer and a gradient buffer; the gradient region is cleared by ONE memset when
// the backward starts and every backward operator ACCUMULATES into the gradients of its inputs (GEMMs through the
// atomic-accumulate form of the split-bf16 kernel, scatters through the existing atomic kernels), which is what makes
// fan-out (skip connections, the residual shortcut, the descriptors both attention directions read) need no extra pass.
// Parameter gradients are accumulated straight into the caller's buffers (pcrcg_amd/trainer.py: views of the flat
// all-reduce bucket), given as a second pcrcg_model whose pointers are the gradient twins of the first one's weights.
//
// The operator set is the un-fused one of pcrcg_amd/train_forward.py (same kernels, same order), so the values equal
// the inference runner's up to summation order; tests/test_train_step_gpu.py and tests/test_autograd_gpu.py hold both
// the values and every parameter gradient against torch autograd of the CPU oracle.
#include <cmath>
#include <functional>
#include <mutex>
#include <vector>

#include "common.h"
#include "pcrcg_train.h"

namespace pcrcg {
// gemm.hip / trainops.hip
// grad_operand: 1 = A holds gradients, 2 = B does, 0 = neither (see GemmExtra::grad_operand, common.h)
int gemm_general(const float* a, int lda, int trans_a, const float* b, int ldb, int trans_b, float* c, int ldc, int m, int n,
                 int k, const float* row_scale, const float* bias, bool accumulate, hipStream_t st, int grad_operand = 0);
int gemm_bt_colstats(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k,
                     const float* row_scale, const float* bias, void* colstats, size_t colstats_bytes, int* h_chunks,
                     hipStream_t st, bool c_zeroed, bool colstats_sums);
int tr_scale_rows(const float* src, int ld_src, const float* s, float* dst, int rows, int cols, hipStream_t st);
int tr_add_lrelu(const float* a, int lda, const float* b, int ldb, float slope, float* y, int ldy, int rows, int cols,
                 hipStream_t st);
int tr_add_lrelu_bwd(const float* y, int ldy, const float* dy, int ld_dy, float slope, float* ga, int lga, float* gb, int lgb,
                     int rows, int cols, hipStream_t st);
int tr_add2d(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols, hipStream_t st);
int tr_bias_grad(const float* dy, int ld, int rows, int cols, float* db, hipStream_t st);
int tr_l2norm_bwd(const float* x, int ldx, const float* dy, int ld_dy, float* dx, int ld_dx, int rows, int cols, hipStream_t st);
int tr_sigmoid_bwd(const float* s, const float* ds, float* dx, int ld_dx, int rows, hipStream_t st);
int tr_dot_acc(const float* a, const float* b, long n, float scale, float* out, hipStream_t st);
bool instnorm_backward_sums_ok(const float* x, int n, int c, int ldx, const float* dy, int ld_dy, const float* dx, int ld_dx);
int instnorm_backward_sums(const float* x, int n, int c, int ldx, const float* stats, const float* dy, int ld_dy, float slope,
                           float* dx, int ld_dx, double* sums, hipStream_t st);

namespace {

struct TT {                 // a tensor of the tape: value, gradient (NULL: not differentiable), row-major view
    float* p = nullptr;
    float* g = nullptr;
    int rows = 0, cols = 0, ld = 0;
};
struct Wt {                 // a weight and the buffer its gradient is accumulated into (NULL: frozen)
    const float* p = nullptr;
    float* g = nullptr;
};

struct Region {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    size_t top = 0;            // bytes handed out from the END of the region (take_top)
    // from the end, growing downwards: the gradient region keeps there the tensors whose gradient the backward WRITES before
    // anything reads it -- its one memset covers [0, off) only
    void* take_top(size_t bytes) {
        bytes = (bytes + 255) & ~size_t(255);
        top += bytes;
        return (base && top <= cap) ? base + ((cap - top) & ~size_t(255)) : nullptr;
    }
    void* take(size_t bytes) {
        bytes = (bytes + 255) & ~size_t(255);
        void* q = base ? base + off : nullptr;
        off += bytes;
        if (off > peak) peak = off;
        return q;
    }
};

// The backward's second stream (one per device, created at first use and kept): weight gradients are off the critical
// path -- nothing in the backward reads them -- so their products (dW = dY^T X, dW_kp = wf^T dY: 1/3 of the backward's GEMM
// time) run beside the activation-gradient chain instead of inside it.  Fork = an event on the main stream the side stream
// waits for (what the product reads is complete by then: gradients of an operator's output are final when its backward
// starts, forward values never change); join = one event at the end of the backward.
struct SideStream {
    hipStream_t st = nullptr;
    std::vector<hipEvent_t> ev;
    hipEvent_t joined = nullptr;
    unsigned next = 0;
};
// The side stream must not share the main stream's hardware DISPATCHER (round 6, profiles/r06_queue_pipes.txt: gfx950 has
// four; two busy streams on one take turns kernel by kernel -- the weight-gradient products would then run BETWEEN the
// activation-gradient kernels instead of beside them: 14.2 instead of 12.3 ms per step, which is what the step cost whenever
// a few engines had created streams in the process before the first backward and the runtime's round-robin put the new stream
// on the main stream's dispatcher).  So: up to six candidates, classified against the main stream by measurement
// (pcrcg_stream_pipe_classes, once per device, ~10 ms; the first backward pays it), the first one of another class is kept.
static SideStream* side_stream(hipStream_t main_stream) {
    static std::mutex mu;
    static SideStream* per_device[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!per_device[dev]) {
        SideStream* s = new SideStream();
        constexpr int kCand = 6;
        hipStream_t cand[kCand] = {};
        int made = 0;
        for (; made < kCand; ++made)
            if (hipStreamCreateWithFlags(&cand[made], hipStreamNonBlocking) != hipSuccess) break;
        if (made == 0) { delete s; return nullptr; }
        int pick = 0;
        void* scratch = nullptr;
        if (hipMalloc(&scratch, 64) == hipSuccess) {
            void* all[kCand + 1];
            int cls[kCand + 1];
            all[0] = main_stream;
            for (int i = 0; i < made; ++i) all[i + 1] = cand[i];
            if (pcrcg_stream_pipe_classes(all, made + 1, cls, scratch) == PCRCG_OK)
                for (int i = 0; i < made; ++i)
                    if (cls[i + 1] != cls[0]) { pick = i; break; }
            (void)hipFree(scratch);
        }
        s->st = cand[pick];
        for (int i = 0; i < made; ++i)
            if (i != pick) (void)hipStreamDestroy(cand[i]);
        s->ev.resize(64);
        bool ok = hipEventCreateWithFlags(&s->joined, hipEventDisableTiming) == hipSuccess;
        for (auto& e : s->ev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (!ok) { delete s; return nullptr; }       // (a failed creation leaks a few handles once; the backward then runs on one stream)
        per_device[dev] = s;
    }
    return per_device[dev];
}

struct Tape {
    Region val, grad, scratch;
    SideStream* side = nullptr;         // set by the backward when the second stream is in use
    bool forked = false;
    // the stream for work nothing later in the backward depends on (see SideStream); the main stream without one
    hipStream_t off_path() {
        if (!side) return st;
        hipEvent_t e = side->ev[side->next++ % side->ev.size()];
        if (hipEventRecord(e, st) != hipSuccess || hipStreamWaitEvent(side->st, e, 0) != hipSuccess) return st;
        forked = true;
        return side->st;
    }
    void join() {
        if (!forked) return;
        forked = false;
        if (hipEventRecord(side->joined, side->st) != hipSuccess || hipStreamWaitEvent(st, side->joined, 0) != hipSuccess) {
            (void)hipStreamSynchronize(side->st);
        }
    }
    bool dry = true;
    int rc = PCRCG_OK;
    hipStream_t st = nullptr;
    std::vector<std::function<void(Tape&)>> bw;
    size_t bw_scratch = 0;              // largest scratch need of one backward operator
    float* d_inv_t = nullptr;           // gradient of 1 / temperature (one float in the gradient region)
    float inv_t = 1.0f;
    TT x_final;                         // [N0, final_dim + 2] before the heads
    float *feats_f = nullptr, *s_ov = nullptr, *s_sal = nullptr;
    int n0 = 0, fd = 0;

    bool live() const { return !dry && rc == PCRCG_OK; }
    void check(int r) { if (r != PCRCG_OK && rc == PCRCG_OK) rc = r; }
    bool fits() {
        if (dry) return true;
        if ((val.off > val.cap || grad.off + grad.top + 256 > grad.cap) && rc == PCRCG_OK) {
            set_error("pcrcg_kpfcnn_train_forward: workspace too small");
            rc = PCRCG_EWORKSPACE;
        }
        return rc == PCRCG_OK;
    }
    // grad_written: the tensor's one consumer WRITES its gradient (an InstanceNorm behind a product): it need not be cleared
    TT tensor(int rows, int cols, bool want_grad = true, int ld = 0, bool grad_written = false) {
        TT t;
        t.rows = rows; t.cols = cols; t.ld = ld ? ld : cols;
        const size_t bytes = sizeof(float) * (size_t)(rows > 0 ? rows : 1) * t.ld;
        t.p = static_cast<float*>(val.take(bytes));
        if (want_grad) t.g = static_cast<float*>(grad_written ? grad.take_top(bytes) : grad.take(bytes));
        fits();
        return t;
    }
    void* value_bytes(size_t bytes) { void* q = val.take(bytes); fits(); return q; }
    // [2][c] fp64 column-sum accumulators for a product whose epilogue leaves its output's InstanceNorm statistics (round 5):
    // carved from the head of the SCRATCH region, which only the backward uses and which the forward clears once up to
    // kSumsBytes; NULL when that head is used up (the norm then makes its own pass, as before)
    static constexpr size_t kSumsBytes = 1 << 20;
    size_t sums_off = 0;
    double* sums_slot(int c) {
        const size_t bytes = ((size_t)2 * c * sizeof(double) + 255) & ~size_t(255);
        if (dry || !scratch.base || !debug_opts().stat_sums || sums_off + bytes > kSumsBytes || sums_off + bytes > scratch.cap)
            return nullptr;              // (stat_sums = 0 -- the deterministic mode -- keeps the atomics-free statistics pass)
        double* q = reinterpret_cast<double*>(scratch.base + sums_off);
        sums_off += bytes;
        return q;
    }
    void need_scratch(size_t bytes) { if (bytes > bw_scratch) bw_scratch = bytes; }
    template <typename F> void record(F&& f) { if (!dry) bw.emplace_back(std::forward<F>(f)); }
    // backward-time scratch (stack discipline inside one operator)
    float* tmp(size_t floats) {
        float* q = static_cast<float*>(scratch.take(floats * sizeof(float)));
        if (scratch.off > scratch.cap && rc == PCRCG_OK) {
            set_error("pcrcg_kpfcnn_train_backward: scratch too small");
            rc = PCRCG_EWORKSPACE;
        }
        return q;
    }
};

inline TT cols(const TT& t, int c0, int n) {
    TT r = t;
    r.p = t.p ? t.p + c0 : nullptr;
    r.g = t.g ? t.g + c0 : nullptr;
    r.cols = n;
    return r;
}
inline TT rows(const TT& t, int r0, int n) {
    TT r = t;
    r.p = t.p ? t.p + (long)r0 * t.ld : nullptr;
    r.g = t.g ? t.g + (long)r0 * t.ld : nullptr;
    r.rows = n;
    return r;
}
inline size_t fbytes(long rows, long cols) { return sizeof(float) * (size_t)(rows > 0 ? rows : 1) * (size_t)cols + 256; }

// ---- operators: forward now, backward recorded ---------------------------------------------------------------------

// y = x @ W^T (+ bias); W [out, in] with leading dimension ldw   (nn.Linear / 1x1 convolution on row-major features)
// sums_out (optional): receives fp64 column sums [2][out] of y left by the product's epilogue (for the InstanceNorm that
// follows), or NULL when this product could not leave them (split over K, no slot)
// y_grad_written: y goes into ONE InstanceNorm and nowhere else (its backward writes y's gradient: see Tape::tensor)
TT linear(Tape& t, const TT& x, Wt w, int ldw, Wt bias, int out, TT* into = nullptr, const double** sums_out = nullptr,
          bool y_grad_written = false) {
    TT y = into ? *into : t.tensor(x.rows, out, true, 0, y_grad_written);
    if (sums_out) *sums_out = nullptr;
    if (t.live()) {
        double* slot = sums_out ? t.sums_slot(out) : nullptr;
        if (slot) {
            int chunks = 0;
            t.check(gemm_bt_colstats(x.p, x.ld, w.p, ldw, y.p, y.ld, x.rows, out, x.cols, nullptr, bias.p, slot,
                                     (size_t)2 * out * sizeof(double), &chunks, t.st, false, true));
            if (chunks == -1) *sums_out = slot;
        } else {
            t.check(gemm_general(x.p, x.ld, 0, w.p, ldw, 1, y.p, y.ld, x.rows, out, x.cols, nullptr, bias.p, false, t.st));
        }
    }
    t.record([x, y, w, ldw, bias, out](Tape& b) {
        if (x.g)     // dx += dy @ W
            b.check(gemm_general(y.g, y.ld, 0, w.p, ldw, 0, x.g, x.ld, x.rows, x.cols, out, nullptr, nullptr, true, b.st, 1));
        if (w.g || bias.g) {
            hipStream_t side = b.off_path();
            if (w.g)     // dW += dy^T @ x
                b.check(gemm_general(y.g, y.ld, 1, x.p, x.ld, 0, w.g, ldw, out, x.cols, x.rows, nullptr, nullptr, true, side, 1));
            if (bias.g) b.check(tr_bias_grad(y.g, y.ld, y.rows, out, bias.g, side));
        }
    });
    return y;
}

// y = a @ b^T  (both k-contiguous), gradients to both
TT matmul_bt(Tape& t, const TT& a, const TT& bm) {
    TT y = t.tensor(a.rows, bm.rows);
    if (t.live())
        t.check(gemm_general(a.p, a.ld, 0, bm.p, bm.ld, 1, y.p, y.ld, a.rows, bm.rows, a.cols, nullptr, nullptr, false, t.st));
    t.record([a, bm, y](Tape& b) {
        if (a.g) b.check(gemm_general(y.g, y.ld, 0, bm.p, bm.ld, 0, a.g, a.ld, a.rows, a.cols, bm.rows, nullptr, nullptr, true, b.st, 1));
        if (bm.g) b.check(gemm_general(y.g, y.ld, 1, a.p, a.ld, 0, bm.g, bm.ld, bm.rows, a.cols, a.rows, nullptr, nullptr, true, b.st, 1));
    });
    return y;
}

// y = p @ v  (v [m, d] row-major = k-major B), into `into` when given
TT matmul_nn(Tape& t, const TT& p, const TT& v, TT* into = nullptr) {
    TT y = into ? *into : t.tensor(p.rows, v.cols);
    if (t.live())
        t.check(gemm_general(p.p, p.ld, 0, v.p, v.ld, 0, y.p, y.ld, p.rows, v.cols, p.cols, nullptr, nullptr, false, t.st));
    t.record([p, v, y](Tape& b) {
        if (p.g) b.check(gemm_general(y.g, y.ld, 0, v.p, v.ld, 1, p.g, p.ld, p.rows, p.cols, v.cols, nullptr, nullptr, true, b.st, 1));
        if (v.g) b.check(gemm_general(p.p, p.ld, 1, y.g, y.ld, 0, v.g, v.ld, p.cols, v.cols, p.rows, nullptr, nullptr, true, b.st, 2));
    });
    return y;
}

// InstanceNorm over the rows + LeakyReLU(slope)  (ref:models/blocks.py:448-462); x has this one consumer
// sums: the producing product's column sums (linear / kpconv with sums_out), or NULL: the statistics take their own pass
TT instnorm_lrelu(Tape& t, const TT& x, float slope, TT* into = nullptr, const double* sums = nullptr) {
    TT y = into ? *into : t.tensor(x.rows, x.cols);
    float* stats = static_cast<float*>(t.value_bytes(sizeof(float) * 2 * x.cols));
    const size_t wsb = pcrcg_instnorm_ws_bytes(x.cols), bwb = pcrcg_instnorm_backward_ws_bytes(x.cols);
    void* ws = t.value_bytes(wsb);
    if (t.live()) {
        const int c4 = x.cols / 4;
        const bool tiles = x.cols % 4 == 0 && c4 >= 1 && (c4 <= 256 ? 256 % c4 == 0 : c4 % 256 == 0) && x.ld % 4 == 0 && y.ld % 4 == 0 &&
                           ((reinterpret_cast<uintptr_t>(x.p) | reinterpret_cast<uintptr_t>(y.p)) & 15) == 0;
        if (!sums && tiles && x.rows > 0) {
            // no sums from the producer (a product split over K, a row slice of a stacked tensor): ONE launch that leaves them
            // in a zeroed slot (fp64 atomics of a few row chunks) instead of partial sums + a finishing launch
            double* slot = t.sums_slot(x.cols);
            const float* xs[1] = {x.p};
            double* ss[1] = {slot};
            const int ns[1] = {x.rows};
            if (slot) {
                t.check(instnorm_colsums_multi(xs, ss, ns, 1, x.cols, x.ld, t.st));
                sums = slot;
            }
        }
        if (sums && tiles && x.rows > 0) {
            // statistics straight from the product's column sums inside the normalising kernel, which also leaves the
            // (mean, rstd) pairs for the backward: one launch
            NormJob one{x.p, sums, nullptr, nullptr, y.p, nullptr, nullptr, x.rows, (double)x.rows};
            one.stats_out = stats;
            t.check(instnorm_apply_sums_multi(&one, 1, x.cols, x.ld, 1e-5f, 0, slope, y.ld, false, t.st));
        } else {
            if (sums) t.check(pcrcg_instnorm_stats_from_partials(sums, 1, x.cols, (double)x.rows, 1e-5f, stats, t.st));
            else t.check(pcrcg_instnorm_stats(x.p, x.rows, x.cols, x.ld, 1e-5f, stats, ws, wsb, t.st));
            t.check(pcrcg_instnorm_apply(x.p, x.rows, x.cols, x.ld, stats, nullptr, 0, nullptr, slope, y.p, y.ld, t.st));
        }
    }
    // few-row tensors (the coarse levels, the GNN): the backward's two column sums meet by atomics in a slot of the gradient
    // region (cleared by the backward's one memset): no finishing launch
    double* bw_sums = x.rows <= 32 * 32 ? static_cast<double*>(t.grad.take(sizeof(double) * 2 * x.cols)) : nullptr;
    t.fits();
    t.need_scratch(bwb + 256);
    t.record([x, y, stats, slope, bwb, bw_sums](Tape& b) {
        if (!x.g) return;
        if (bw_sums && instnorm_backward_sums_ok(x.p, x.rows, x.cols, x.ld, y.g, y.ld, x.g, x.ld)) {
            b.check(instnorm_backward_sums(x.p, x.rows, x.cols, x.ld, stats, y.g, y.ld, slope, x.g, x.ld, bw_sums, b.st));
            return;
        }
        void* w = b.tmp((bwb + 3) / 4);
        b.check(pcrcg_instnorm_backward(x.p, x.rows, x.cols, x.ld, stats, y.g, y.ld, slope, x.g, x.ld, w, bwb, b.st));
    });
    return y;
}

// y = lrelu(a + b, slope)   (slope 1: a plain sum)
TT add_lrelu(Tape& t, const TT& a, const TT& bb, float slope, TT* into = nullptr) {
    TT y = into ? *into : t.tensor(a.rows, a.cols);
    if (t.live()) t.check(tr_add_lrelu(a.p, a.ld, bb.p, bb.ld, slope, y.p, y.ld, a.rows, a.cols, t.st));
    t.record([a, bb, y, slope](Tape& b) {
        b.check(tr_add_lrelu_bwd(y.p, y.ld, y.g, y.ld, slope, a.g, a.ld, bb.g, bb.ld, a.rows, a.cols, b.st));
    });
    return y;
}

// dst[:, c0:c0+src.cols] = src (a piece of torch.cat); the gradient of the slice flows back into src
void copy_into(Tape& t, const TT& src, const TT& dst_slice) {
    if (t.live()) t.check(pcrcg_copy2d(src.p, src.ld, dst_slice.p, dst_slice.ld, src.rows, src.cols, t.st));
    t.record([src, dst_slice](Tape& b) {
        if (src.g && dst_slice.g) b.check(tr_add2d(dst_slice.g, dst_slice.ld, src.g, src.ld, src.rows, src.cols, b.st));
    });
}

// KPConv.forward (ref:models/blocks.py:229-374): aggregate, contract, divide by the neighbour count
TT kpconv(Tape& t, const pcrcg_batch& b, const pcrcg_block& blk, Wt w, const TT& x, const double** sums_out = nullptr) {
    if (sums_out) *sums_out = nullptr;
    const int l = blk.layer;
    const pcrcg_table& tab = blk.strided ? b.pools[l] : b.neighbors[l];
    const float* q = blk.strided ? b.points[l + 1] : b.points[l];
    const int nq = blk.strided ? b.n_points[l + 1] : b.n_points[l];
    const int ns = b.n_points[l], cin = x.cols, cout = blk.mid_dim, kc = PCRCG_KPOINTS * cin;
    TT y = t.tensor(nq, cout, true, 0, true);       // (every KPConv of the network feeds one InstanceNorm: encoder_block)
    float* wf = static_cast<float*>(t.value_bytes(fbytes(nq, kc)));
    float* inv_n = static_cast<float*>(t.value_bytes(fbytes(nq, 1)));
    const size_t wsb = pcrcg_kpconv_ws_bytes(ns);
    void* ws = t.value_bytes(wsb);
    if (t.live()) {
        t.check(pcrcg_kpconv_aggregate(q, nq, b.points[l], ns, tab.idx, tab.cols, tab.ld, x.p, cin, blk.kp, blk.extent, wf,
                                       inv_n, ws, wsb, t.st));
        // the contraction: with the caller's K-contiguous copy of the weights (pcrcg_block.kp_wt, [cout, 15 cin]) the
        // C = A B^T form with both operands k-contiguous -- the inference runner's product; without it the k-major form
        double* slot = (sums_out && blk.kp_wt && kc % 4 == 0) ? t.sums_slot(cout) : nullptr;
        if (slot) {
            int chunks = 0;
            t.check(gemm_bt_colstats(wf, kc, blk.kp_wt, kc, y.p, y.ld, nq, cout, kc, inv_n, nullptr, slot,
                                     (size_t)2 * cout * sizeof(double), &chunks, t.st, false, true));
            if (chunks == -1) *sums_out = slot;
        } else if (blk.kp_wt && kc % 4 == 0)
            t.check(gemm_general(wf, kc, 0, blk.kp_wt, kc, 1, y.p, y.ld, nq, cout, kc, inv_n, nullptr, false, t.st));
        else
            t.check(gemm_general(wf, kc, 0, w.p, cout, 0, y.p, y.ld, nq, cout, kc, inv_n, nullptr, false, t.st));
    }
    t.need_scratch(fbytes(nq, kc));
    float* dys = static_cast<float*>(t.value_bytes(fbytes(nq, cout)));          // dy / n: read by the off-path dW product
    const float* s_pts = b.points[l];
    const float* kp = blk.kp;
    const float extent = blk.extent;
    t.record([=](Tape& bk) {
        bk.check(tr_scale_rows(y.g, y.ld, inv_n, dys, nq, cout, bk.st));          // dy / n
        if (w.g)                                                                  // dW += wf^T @ (dy / n)
            bk.check(gemm_general(wf, kc, 1, dys, cout, 0, w.g, cout, kc, cout, nq, nullptr, nullptr, true, bk.off_path(), 2));
        if (x.g) {                                                                // d wf = (dy / n) @ W^T, scattered through w
            float* d_wf = bk.tmp((size_t)(nq > 0 ? nq : 1) * kc + 64);
            bk.check(gemm_general(dys, cout, 0, w.p, cout, 1, d_wf, kc, nq, kc, cout, nullptr, nullptr, false, bk.st, 1));
            bk.check(pcrcg_kpconv_backward_dx(q, nq, s_pts, ns, tab.idx, tab.cols, tab.ld, d_wf, cin, kp, extent, x.g, bk.st));
        }
    });
    return y;
}

TT max_pool(Tape& t, const TT& x, const pcrcg_table& tab) {
    TT y = t.tensor(tab.rows, x.cols);
    if (t.live()) t.check(pcrcg_gather_max(x.p, x.rows, x.cols, tab.idx, tab.rows, tab.cols, tab.ld, y.p, t.st));
    t.record([x, y, tab](Tape& b) {
        if (x.g)
            b.check(pcrcg_gather_max_backward(x.p, x.rows, x.cols, tab.idx, tab.rows, tab.cols, tab.ld, y.p, y.g, x.g, b.st));
    });
    return y;
}

// closest_pool into a column slice of `dst` (the first part of the decoder's concatenation)
void closest_pool_into(Tape& t, const TT& x, const pcrcg_table& tab, const TT& dst) {
    if (t.live()) t.check(pcrcg_gather_first(x.p, x.rows, x.cols, tab.idx, tab.rows, tab.ld, dst.p, dst.ld, t.st));
    t.record([x, tab, dst](Tape& b) {
        if (x.g) b.check(pcrcg_gather_first_backward(dst.g, dst.ld, x.cols, tab.idx, tab.rows, tab.ld, x.rows, x.g, b.st));
    });
}

// p = softmax(s * scale) row-wise, in place on s's buffer; `scale_grad` (optional): accumulates dL/dscale
TT softmax_rows(Tape& t, const TT& s, float scale, float* scale_grad = nullptr) {
    // the raw scores are needed for dL/dscale only; keep a copy then
    float* raw = nullptr;
    if (scale_grad) raw = static_cast<float*>(t.value_bytes(fbytes(s.rows, s.cols)));
    if (t.live()) {
        if (raw) t.check(pcrcg_copy2d(s.p, s.ld, raw, s.cols, s.rows, s.cols, t.st));
        t.check(pcrcg_softmax_rows(s.p, s.rows, s.cols, s.ld, scale, t.st));
    }
    t.need_scratch(fbytes(s.rows, s.cols));
    t.record([s, scale, raw, scale_grad](Tape& b) {
        // ds = scale * p * (dp - sum p dp): written over dp in place is not possible (reads the whole row): scratch
        float* ds = b.tmp((size_t)(s.rows > 0 ? s.rows : 1) * s.cols + 64);
        b.check(pcrcg_softmax_rows_backward(s.p, s.ld, s.g, s.ld, s.rows, s.cols, scale, ds, s.cols, b.st));
        if (scale_grad)     // z = raw * scale: dL/dscale = sum (ds / scale) * raw
            b.check(tr_dot_acc(ds, raw, (long)s.rows * s.cols, 1.0f / scale, scale_grad, b.st));
        b.check(pcrcg_copy2d(ds, s.cols, s.g, s.ld, s.rows, s.cols, b.st));          // the gradient of the raw scores
    });
    return s;
}

// ---- blocks (ref:models/blocks.py) ---------------------------------------------------------------------------------
TT unary(Tape& t, const TT& x, Wt w, int ldw, int out, float slope) {
    const double* sums = nullptr;
    TT y = linear(t, x, w, ldw, Wt(), out, nullptr, &sums, true);
    return instnorm_lrelu(t, y, slope, nullptr, sums);
}

Wt wt(const float* p, const float* g) { Wt w; w.p = p; w.g = const_cast<float*>(g); return w; }

TT encoder_block(Tape& t, const pcrcg_batch& b, const pcrcg_block& blk, const pcrcg_block& gb, const TT& x) {
    const double* ksums = nullptr;
    if (blk.type == PCRCG_BLK_SIMPLE) {                                          // :579-590
        TT c0 = kpconv(t, b, blk, wt(blk.kp_w, gb.kp_w), x, &ksums);
        return instnorm_lrelu(t, c0, 0.1f, nullptr, ksums);
    }
    TT y = x;                                                                    // resnetb :650-678
    if (blk.unary1) y = unary(t, x, wt(blk.unary1, gb.unary1), x.cols, blk.mid_dim, 0.1f);
    {
        TT c1 = kpconv(t, b, blk, wt(blk.kp_w, gb.kp_w), y, &ksums);
        y = instnorm_lrelu(t, c1, 0.1f, nullptr, ksums);
    }
    y = unary(t, y, wt(blk.unary2, gb.unary2), blk.mid_dim, blk.out_dim, 1.0f);  // no_relu
    TT sc = blk.strided ? max_pool(t, x, b.pools[blk.layer]) : x;
    if (blk.shortcut) sc = unary(t, sc, wt(blk.shortcut, gb.shortcut), sc.cols, blk.out_dim, 1.0f);
    return add_lrelu(t, y, sc, 0.1f);
}

// ---- GNN head (ref:models/gcn.py) ----------------------------------------------------------------------------------
// max_j lrelu(IN2d(ctr_i + nbr_idx[i,j]), 0.2), statistics over all n*k edges (:37-64, 121-129)
// the part behind the two weight products: ctr / nbr [n, cout] dense, y = *into (a column slice of the concatenation)
void edge_conv_core(Tape& t, const TT& ctr, const TT& nbr, int cout, const int* idx, int k, const TT& y) {
    const int n = ctr.rows;
    TT emax = t.tensor(n, cout, false);
    float* stats = static_cast<float*>(t.value_bytes(sizeof(float) * 2 * cout));
    const size_t wsb = pcrcg_edgeconv_ws_bytes(cout), bwb = pcrcg_edgeconv_backward_ws_bytes(cout);
    void* ws = t.value_bytes(wsb);
    if (t.live()) {
        t.check(pcrcg_edgeconv_reduce(ctr.p, ctr.ld, nbr.p, nbr.ld, idx, n, k, cout, 1e-5f, emax.p, emax.ld, stats, ws, wsb, t.st));
        t.check(pcrcg_instnorm_apply(emax.p, n, cout, emax.ld, stats, nullptr, 0, nullptr, 0.2f, y.p, y.ld, t.st));
    }
    t.need_scratch(bwb + fbytes(n, cout) + 512);
    t.record([=](Tape& b) {
        // the kernel wants a dense dy: the output is a column slice of the concatenation
        float* dy = b.tmp((size_t)n * cout + 64);
        b.check(pcrcg_copy2d(y.g, y.ld, dy, cout, n, cout, b.st));
        void* w = b.tmp((bwb + 3) / 4 + 64);
        // dctr is written, dnbr accumulated (both gradients are zero before: single consumers)
        b.check(pcrcg_edgeconv_backward(ctr.p, nbr.p, idx, n, k, cout, stats, dy, 0.2f, ctr.g, nbr.g, w, bwb, b.st));
    });
}
// f holds the rows of `clouds` clouds one after the other (rows_of[c] each): the two weight products run ONCE over all rows,
// the edge maxima and their statistics per cloud (idx[c]: that cloud's kNN graph, k[c] columns)
void edge_conv(Tape& t, const TT& f, Wt w_packed, int cout, int clouds, const int* rows_of, int* const* idx, const int* k,
               const TT& into) {
    // packed weight [2*cout, cin]: rows [0, cout) = Wa - Wb (centre term), rows [cout, 2 cout) = Wb (neighbour term)
    const int cin = f.cols;
    Wt wc = w_packed, wn = w_packed;
    wn.p = w_packed.p + (long)cout * cin;
    wn.g = w_packed.g ? w_packed.g + (long)cout * cin : nullptr;
    TT ctr = linear(t, f, wc, cin, Wt(), cout), nbr = linear(t, f, wn, cin, Wt(), cout);
    for (int c = 0, r0 = 0; c < clouds; r0 += rows_of[c], ++c)
        edge_conv_core(t, rows(ctr, r0, rows_of[c]), rows(nbr, r0, rows_of[c]), cout, idx[c], k[c], rows(into, r0, rows_of[c]));
}

// One DGCNN self-attention layer for the stacked rows of `clouds` clouds (round 5: the pair's two clouds share every weight
// product -- one launch forward, one dX and one dW backward instead of two each; kNN graphs, edge maxima and InstanceNorm
// statistics stay per cloud, so the clouds never mix)
TT self_attention(Tape& t, const pcrcg_model& m, const pcrcg_gnn_layer& g, const pcrcg_gnn_layer& gg, int clouds,
                  const float* const* coords, const int* rows_of, const TT& f) {
    const int n = f.rows, ch = f.cols;
    int* idx[2];
    int k[2];
    for (int c = 0; c < clouds; ++c) {
        k[c] = m.knn_k < rows_of[c] - 1 ? m.knn_k : rows_of[c] - 1;
        idx[c] = static_cast<int*>(t.value_bytes(sizeof(int) * (size_t)rows_of[c] * (k[c] > 0 ? k[c] : 1) + 256));
        if (t.live()) t.check(pcrcg_knn(coords[c], rows_of[c], k[c], idx[c], t.st));
    }
    TT cat = t.tensor(n, 4 * ch);
    copy_into(t, f, cols(cat, 0, ch));                                                           // x0
    TT s1 = cols(cat, ch, ch), s2 = cols(cat, 2 * ch, 2 * ch);
    edge_conv(t, f, wt(g.edge1, gg.edge1), ch, clouds, rows_of, idx, k, s1);                     // x1 :121-125
    edge_conv(t, s1, wt(g.edge2, gg.edge2), 2 * ch, clouds, rows_of, idx, k, s2);                // x2 :127-129
    TT x3 = linear(t, cat, wt(g.conv3, gg.conv3), 4 * ch, Wt(), ch, nullptr, nullptr, true);     // :131-132
    TT out = t.tensor(n, ch);
    for (int c = 0, r0 = 0; c < clouds; r0 += rows_of[c], ++c) {
        TT oc = rows(out, r0, rows_of[c]);
        instnorm_lrelu(t, rows(x3, r0, rows_of[c]), 0.2f, &oc);
    }
    return out;
}

// x + AttentionalPropagation(x, src)   (:151-185, 213-214); weights head-major (runner.py permutes them)
TT cross_attention(Tape& t, const pcrcg_model& m, const pcrcg_gnn_layer& g, const pcrcg_gnn_layer& gg, const TT& x,
                   const TT& src, TT* into = nullptr) {
    const int n = x.rows, ms = src.rows, ch = x.cols, h = m.heads, d = ch / h;
    TT q = linear(t, x, wt(g.wq, gg.wq), ch, wt(g.bq, gg.bq), ch);
    TT kk = linear(t, src, wt(g.wk, gg.wk), ch, wt(g.bk, gg.bk), ch);
    TT v = linear(t, src, wt(g.wv, gg.wv), ch, wt(g.bv, gg.bv), ch);
    TT msg = t.tensor(n, ch);
    const float scale = 1.0f / sqrtf((float)d);
    AttnCloud one{q.p, kk.p, v.p, msg.p, n, ms};
    // every head in ONE launch each way on the fp32 matrix cores (round 5; the sizing pass keeps the per-head path's
    // workspace, which is the larger one)
    const bool fused = !t.dry && q.g && kk.g && v.g && attention_mfma_ok(&one, 1, q.ld, kk.ld, v.ld, d) &&
                       attention_bwd_mfma_ok(n, ms, d, q.ld, kk.ld, v.ld, msg.ld);
    if (fused) {
        if (t.live()) t.check(attention_mfma_multi(&one, 1, q.ld, kk.ld, v.ld, msg.ld, h, d, scale, t.st));
        t.record([q, kk, v, msg, n, ms, h, d, scale](Tape& b) {
            b.check(attention_bwd_mfma(q.p, q.ld, kk.p, kk.ld, v.p, v.ld, msg.p, msg.ld, msg.g, msg.ld, q.g, q.ld, kk.g, kk.ld, v.g,
                                       v.ld, n, ms, h, d, scale, b.st));
        });
    } else {
        for (int i = 0; i < h; ++i) {
            TT qi = cols(q, i * d, d), ki = cols(kk, i * d, d), vi = cols(v, i * d, d), mi = cols(msg, i * d, d);
            TT prob = softmax_rows(t, matmul_bt(t, qi, ki), scale);
            matmul_nn(t, prob, vi, &mi);
        }
    }
    TT cat = t.tensor(n, 2 * ch);
    copy_into(t, x, cols(cat, 0, ch));
    TT merged = cols(cat, ch, ch);
    linear(t, msg, wt(g.wm, gg.wm), ch, wt(g.bm, gg.bm), ch, &merged);
    TT h0 = linear(t, cat, wt(g.w0, gg.w0), 2 * ch, wt(g.b0, gg.b0), 2 * ch, nullptr, nullptr, true);
    TT h1 = instnorm_lrelu(t, h0, 0.0f);                                                        // InstanceNorm1d + ReLU
    TT delta = linear(t, h1, wt(g.w3, gg.w3), 2 * ch, wt(g.b3, gg.b3), ch);
    return add_lrelu(t, x, delta, 1.0f, into);
}

void forward(Tape& t, const pcrcg_model& m, const pcrcg_model& gm, const pcrcg_batch& b) {
    const int L = b.n_levels;
    TT x;                                   // input features: a constant (no gradient)
    x.p = const_cast<float*>(b.features);
    x.rows = b.n_points[0];
    x.cols = x.ld = b.feat_dim;
    std::vector<TT> skips;
    for (int i = 0; i < m.n_enc; ++i) {                                          // encoder :519-524
        if (m.enc_skip[i]) skips.push_back(x);
        x = encoder_block(t, b, m.enc[i], gm.enc[i], x);
    }
    // bottleneck + GNN (:527-536)
    const int nc = b.n_points[L - 1], ns = b.len_src_c, nt = nc - ns, g = m.gnn_dim;
    TT fc = linear(t, x, wt(m.bottle_w, gm.bottle_w), m.enc_out_dim, wt(m.bottle_b, gm.bottle_b), g);
    // the GNN state of both clouds lives in ONE [ns + nt, g] tensor (source rows first): self-attention layers take it whole
    TT dd = fc;
    const float* coords[2] = {b.points[L - 1], b.points[L - 1] + 3 * (long)ns};
    const int rows_of[2] = {ns, nt};
    for (int i = 0; i < m.n_gnn; ++i) {
        if (m.gnn[i].cross) {
            TT nxt = t.tensor(nc, g);
            TT n0 = rows(nxt, 0, ns), n1 = rows(nxt, ns, nt);
            cross_attention(t, m, m.gnn[i], gm.gnn[i], rows(dd, 0, ns), rows(dd, ns, nt), &n0);
            cross_attention(t, m, m.gnn[i], gm.gnn[i], rows(dd, ns, nt), n0, &n1);          // sees the updated source cloud (:214)
            dd = nxt;
        } else {
            dd = self_attention(t, m, m.gnn[i], gm.gnn[i], 2, coords, rows_of, dd);
        }
    }
    // coarse head (:538-565): x = [score | saliency | proj_gnn feats]
    const TT& gcat = dd;
    const int wc = g + 2;
    TT xc = t.tensor(nc, wc);
    TT feats = cols(xc, 2, g), score = cols(xc, 0, 1), sal = cols(xc, 1, 1);
    linear(t, gcat, wt(m.proj_gnn_w, gm.proj_gnn_w), g, wt(m.proj_gnn_b, gm.proj_gnn_b), g, &feats);      // :538
    linear(t, feats, wt(m.proj_score_w, gm.proj_score_w), g, wt(m.proj_score_b, gm.proj_score_b), 1, &score);   // :539
    TT fn = t.tensor(nc, g);                                                      // F.normalize (:541)
    if (t.live()) t.check(pcrcg_l2norm_rows(feats.p, feats.ld, fn.p, fn.ld, nc, g, t.st));
    t.record([feats, fn, nc, g](Tape& bk) { bk.check(tr_l2norm_bwd(feats.p, feats.ld, fn.g, fn.ld, feats.g, feats.ld, nc, g, bk.st)); });
    TT fs = rows(fn, 0, ns), ft = rows(fn, ns, nt);
    t.inv_t = 1.0f / m.temperature;
    t.d_inv_t = static_cast<float*>(t.grad.take(256));
    t.fits();
    // s1 = softmax(inner / T) @ tgt_scores, s2 = softmax(inner^T / T) @ src_scores (:562-563)
    TT p_st = softmax_rows(t, matmul_bt(t, fs, ft), t.inv_t, t.d_inv_t);
    TT p_ts = softmax_rows(t, matmul_bt(t, ft, fs), t.inv_t, t.d_inv_t);
    TT sal_s = rows(sal, 0, ns), sal_t = rows(sal, ns, nt);
    matmul_nn(t, p_st, rows(score, ns, nt), &sal_s);
    matmul_nn(t, p_ts, rows(score, 0, ns), &sal_t);
    x = xc;
    // decoder (:567-570)
    for (int j = 0; j < m.n_dec; ++j) {
        const pcrcg_block& blk = m.dec[j];
        const pcrcg_block& gb = gm.dec[j];
        if (blk.type == PCRCG_BLK_UPSAMPLE) {
            const pcrcg_table& tab = b.upsamples[blk.layer - 1];
            const bool concat = j + 1 < m.n_dec && m.dec_concat[j + 1];
            const int cs = concat ? skips.back().cols : 0;
            TT y = t.tensor(tab.rows, x.cols + cs);
            closest_pool_into(t, x, tab, cols(y, 0, x.cols));
            if (concat) {
                copy_into(t, skips.back(), cols(y, x.cols, cs));
                skips.pop_back();
            }
            x = y;
        } else if (blk.type == PCRCG_BLK_UNARY) {
            x = unary(t, x, wt(blk.mlp, gb.mlp), blk.mlp_ld, blk.out_dim, 0.1f);
        } else {                                                                   // last_unary: Linear only
            x = linear(t, x, wt(blk.mlp, gb.mlp), blk.mlp_ld, Wt(), blk.out_dim);
        }
    }
    t.x_final = x;
    t.n0 = x.rows;
    t.fd = m.final_dim;
}

int validate(const pcrcg_model* m, const pcrcg_model* g, const pcrcg_batch* b) {
    PCRCG_CHECK_ARG(m && g && b);
    PCRCG_CHECK_ARG(m->n_enc >= 1 && m->n_enc <= PCRCG_MAX_BLOCKS && m->n_dec >= 1 && m->n_dec <= PCRCG_MAX_BLOCKS);
    PCRCG_CHECK_ARG(m->n_gnn >= 0 && m->n_gnn <= PCRCG_MAX_GNN && b->n_levels >= 1 && b->n_levels <= PCRCG_MAX_LEVELS);
    PCRCG_CHECK_ARG(b->len_src_c >= 1 && b->len_src_c < b->n_points[b->n_levels - 1]);
    PCRCG_CHECK_ARG(m->heads >= 1 && m->gnn_dim % m->heads == 0 && m->temperature > 0.0f);
    for (int i = 0; i < m->n_enc; ++i) {
        PCRCG_CHECK_ARG(m->enc[i].layer >= 0 && m->enc[i].layer + m->enc[i].strided < b->n_levels);
        PCRCG_CHECK_ARG(m->enc[i].kp && m->enc[i].kp_w);
    }
    return PCRCG_OK;
}

}  // namespace
}  // namespace pcrcg

using namespace pcrcg;

extern "C" {

int pcrcg_kpfcnn_train_ws_bytes(const pcrcg_model* model, const pcrcg_model* grads, const pcrcg_batch* batch,
                                size_t* value_bytes, size_t* grad_bytes, size_t* scratch_bytes) {
    PCRCG_PROPAGATE(validate(model, grads, batch));
    PCRCG_CHECK_ARG(value_bytes && grad_bytes && scratch_bytes);
    Tape t;
    forward(t, *model, *grads, *batch);
    const int n0 = batch->n_points[0];
    *value_bytes = t.val.peak + 3 * fbytes(n0, model->final_dim) + 4096;      // + the three outputs
    *grad_bytes = t.grad.peak + t.grad.top + 4096;
    *scratch_bytes = t.bw_scratch + 4096;
    return PCRCG_OK;
}

int pcrcg_kpfcnn_train_forward(const pcrcg_model* model, const pcrcg_model* grads, const pcrcg_batch* batch, void* ws,
                               size_t value_bytes, size_t grad_bytes, size_t scratch_bytes, pcrcg_train_outputs* out,
                               void** tape, void* stream) {
    PCRCG_PROPAGATE(validate(model, grads, batch));
    PCRCG_CHECK_ARG(ws && tape && out);
    *tape = nullptr;
    Tape* t = new Tape();
    t->dry = false;
    t->st = as_stream(stream);
    char* base = static_cast<char*>(ws);
    t->val.base = base;
    t->val.cap = value_bytes;
    t->grad.base = base + value_bytes;
    t->grad.cap = grad_bytes;
    t->scratch.base = base + value_bytes + grad_bytes;
    t->scratch.cap = scratch_bytes;
    {   // the column-sum slots of the forward's products (Tape::sums_slot)
        const size_t head = scratch_bytes < Tape::kSumsBytes ? scratch_bytes : Tape::kSumsBytes;
        if (head && hipMemsetAsync(t->scratch.base, 0, head, t->st) != hipSuccess) {
            delete t;
            set_error("pcrcg_kpfcnn_train_forward: clearing the statistics slots failed");
            return PCRCG_ELAUNCH;
        }
    }
    forward(*t, *model, *grads, *batch);
    if (t->rc == PCRCG_OK) {
        // heads (:571-582): L2-normalised descriptors, sigmoid scores
        const int n0 = t->n0, fd = t->fd;
        t->feats_f = static_cast<float*>(t->val.take(fbytes(n0, fd)));
        t->s_ov = static_cast<float*>(t->val.take(fbytes(n0, 1)));
        t->s_sal = static_cast<float*>(t->val.take(fbytes(n0, 1)));
        t->fits();
        if (t->rc == PCRCG_OK) {
            const TT& x = t->x_final;
            t->check(pcrcg_l2norm_rows(x.p, x.ld, t->feats_f, fd, n0, fd, t->st));
            t->check(pcrcg_sigmoid_scores(x.p + fd, x.ld, t->s_ov, n0, t->st));
            t->check(pcrcg_sigmoid_scores(x.p + fd + 1, x.ld, t->s_sal, n0, t->st));
        }
    }
    const int rc = t->rc;
    if (rc != PCRCG_OK) {
        delete t;
        return rc;
    }
    out->feats_f = t->feats_f;
    out->scores_overlap = t->s_ov;
    out->scores_saliency = t->s_sal;
    out->d_inv_temperature = t->d_inv_t;
    out->n_points = t->n0;
    out->final_dim = t->fd;
    *tape = t;
    return PCRCG_OK;
}

int pcrcg_kpfcnn_train_backward(void* tape, const float* d_feats_f, const float* d_scores_overlap,
                                const float* d_scores_saliency, void* stream) {
    PCRCG_CHECK_ARG(tape);
    Tape& t = *static_cast<Tape*>(tape);
    t.st = as_stream(stream);
    t.rc = PCRCG_OK;
    t.side = debug_opts().train_side_stream ? side_stream(t.st) : nullptr;
    t.forked = false;
    PCRCG_CHECK_HIP(hipMemsetAsync(t.grad.base, 0, t.grad.off, t.st));
    // heads: the gradients of the three outputs into the gradient of x_final
    const TT& x = t.x_final;
    const int n0 = t.n0, fd = t.fd;
    if (d_feats_f) t.check(tr_l2norm_bwd(x.p, x.ld, d_feats_f, fd, x.g, x.ld, n0, fd, t.st));
    if (d_scores_overlap) t.check(tr_sigmoid_bwd(t.s_ov, d_scores_overlap, x.g + fd, x.ld, n0, t.st));
    if (d_scores_saliency) t.check(tr_sigmoid_bwd(t.s_sal, d_scores_saliency, x.g + fd + 1, x.ld, n0, t.st));
    for (size_t i = t.bw.size(); i-- > 0 && t.rc == PCRCG_OK;) {
        t.scratch.off = 0;
        t.bw[i](t);
    }
    t.join();                 // the weight gradients are complete when the main stream passes this point
    return t.rc;
}

void pcrcg_kpfcnn_train_free(void* tape) { delete static_cast<Tape*>(tape); }
}
