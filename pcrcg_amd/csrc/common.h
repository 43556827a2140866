// common.h -- shared host-side helpers of libpcrcg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>

#include "pcrcg.h"

namespace pcrcg {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Bump allocator over the caller's workspace; every carve is 256-byte aligned.
struct Carver {
    char* base;
    size_t cap;
    size_t off = 0;
    Carver(void* ws, size_t bytes) : base(static_cast<char*>(ws)), cap(bytes) {}
    template <typename T>
    T* take(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
        T* p = reinterpret_cast<T*>(base + off);
        off += bytes;
        return p;
    }
    bool ok() const { return off <= cap; }
};
inline size_t carve_bytes(size_t count, size_t elem) { return (count * elem + 255) & ~size_t(255); }

// Exclusive prefix sum of int32 (device-wide, any n >= 0).  out may alias in.  If total != nullptr
// the grand total is stored there.  ws: scan_ws_bytes(n).
size_t scan_ws_bytes(int n);
int exclusive_scan_i32(const int* in, int* out, int n, int* total, void* ws, hipStream_t stream);

constexpr int kWave = 64;

// grid_subsample.hip: pcrcg_grid_subsample_batch with the number of points given as a host-side BOUND (the cloud
// lengths on the device say how many there are) and room for out_cap output rows (more: *overflow = 1, output cut)
int grid_subsample_bound(const float* pts, int n_bound, const int* len, int nb, float dl, int max_p, float* out_pts,
                         int* out_len, int* out_m, int out_cap, int* overflow, void* ws, size_t ws_bytes, hipStream_t stream);

// pointops.hip: InstanceNorm + LeakyReLU that also leaves the KPConv support records of its output (kpconv.hip: pk)
bool instnorm_pack_ok(int c, int ldx, int ldy);
int instnorm_apply_pack(const float* x, int n, int c, int ldx, const float* stats, const double* sums, double count, float eps,
                        float slope, float* y, int ldy, const float* s_pts, float4* pk, hipStream_t st);
// kpconv.hip: where the support records live inside a pcrcg_kpconv_ws_bytes(ns) workspace
float4* kpconv_pk_ptr(void* ws, size_t ws_bytes, int ns);

// morton_knock.hip (measurement aid, DebugOpts::pyr_morton): a subsampled level's rows into Z order, in place
size_t morton_knock_ws_bytes(int cap);
int morton_knock_level(float* pts, int cap, const int* len, int nb, void* ws, size_t ws_bytes, hipStream_t st);

// the deterministic debug mode's scratch (gemm_x6.hip, trainops.hip): freed by pcrcg_debug_release()
void gemm_x6_release_det();
void trainops_release_det();

// tieorder.hip: pcrcg_kdforest_build over clouds that are LEVELS of per_level clouds each, level l's rows starting at row
// level_base[l] of sup (per_level = 0: one contiguous stack, the public entry point)
int kdforest_build_levels(const float* sup, int ns, const int* slen, int nb, int per_level, const int* level_base, void* forest,
                          size_t forest_bytes, hipStream_t stream);

// radius.hip: pcrcg_radius_query_groups by pass (0 both kernels, 1 the first, 2 the redo of rows with > radius_fast_cap()
// hits that the first one marked)
int radius_fast_cap();
int radius_query_pass(const float* q, int nq, const int* qlen, int ns, const int* slen, int nb, int group, float radius,
                      const void* grid, int cols, int64_t* out_idx, int* out_count, int* out_max_count, int* status,
                      int* out_tie_rows, int* out_tie_count, hipStream_t st, int pass);

// radius.hip: the cell-cooperative search over a query grid (pass 0: + the per-query second pass, 1: the cell kernel only)
// gemm_x6.hip: the calling host thread enqueues beside other streams (see x6_plan_for; C ABI: pcrcg_thread_shares_gpu)
bool gemm_x6_shared();
void gemm_x6_set_shared(int on);
int radius_cells_pass(const void* qgrid, const float* q, int nq, const int* qlen, const void* sgrid, int ns, const int* slen,
                      int nb, int group, float radius, int cols, int64_t* out_idx, int* out_count, int* out_max_count,
                      int* status, int* out_tie_rows, int* out_tie_count, hipStream_t st, int pass);
constexpr int kRadiusRedoStatus = 4;   // status bit: rows were handed to the per-query second pass (cleared by that pass)

// gemm_x6.hip: optional extras of a C = A * B^T product (the decoder's fused upsample + concat, runner.hip)
struct GemmExtra {
    const long long* a_idx = nullptr;   // != NULL: output row r reads A row a_idx[r * a_idx_ld] (first column of a table)
    int a_idx_ld = 0, a_ns = 0;         // an index outside [0, a_ns) reads a_zero instead (the shadow row)
    const float* a_zero = nullptr;      // >= k zero floats
    bool accumulate = false;            // C += product (fp32 atomics) instead of C = product
    const double* a_sums = nullptr;     // != NULL: A is normalised on load, a' = lrelu((a - mean_k) * rstd_k, a_slope), with
    double a_count = 0.0;               // the statistics of its columns given as fp64 sums [2][k] over a_count rows
    float a_eps = 1e-5f, a_slope = 1.0f;
    int grad_operand = 0;               // train step: 1 = A holds gradients, 2 = B does (the fp16 form scales that operand by 2^16)
};

// gemm_x6.hip: a SECOND product C1 = f(A1) * B^T that shares B (and the bias, the leading dimensions, n, k and every
// GemmExtra setting except the per-product pointers below) with the first and runs in the SAME launch: the same layer
// of a second fragment pair (runner.hip, pcrcg_kpfcnn_forward_group).  Small products fill the chip twice as well
// and every product costs one launch per two pairs.
struct GemmPair {
    const float* a = nullptr;
    float* c = nullptr;
    int m = 0;
    const float* row_scale = nullptr;
    void* colstats = nullptr;           // statistics of C1, same form and size as the first product's
    int* h_chunks = nullptr;
    bool c_zeroed = false;
    const long long* a_idx = nullptr;   // gather form: its own table and source row count (A1 = its source matrix)
    int a_ns = 0;
    const double* a_sums = nullptr;     // normalise-on-load form: its own column sums and row count
    double a_count = 0.0;
};
struct GemmGroup {                      // up to 3 further products in the launch (4 fragment pairs per call)
    int n = 0;
    GemmPair p[3];
};

// kpconv.hip: row-positive flags + packed (x, y, z, flag) support records into a pcrcg_kpconv_ws_bytes(ns) workspace
// (x_bf16 != NULL: also the bf16 round-to-nearest-even copy of x, [ns, cin])
int kpconv_pack(const float* x, int ns, int cin, const float* s_pts, void* ws, size_t ws_bytes, hipStream_t st,
                unsigned short* x_bf16 = nullptr);
int kpconv_aggregate_rows(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx, int h, int ld_idx,
                          const float* x, int cin, const float* kp, float extent, float* wf, float* inv_n, void* ws,
                          size_t ws_bytes, hipStream_t st, bool pack, bool stream_out, int c1_ld = 0);   // c1_ld = 16: cin = 1 with wf rows of 16 floats (15 + a zero)
int kpconv_aggregate_bf16(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx, int h, int ld_idx,
                          const float* x, unsigned short* x_bf16, int cin, const float* kp, float extent,
                          unsigned short* wf_bf16, float* inv_n, void* ws, size_t ws_bytes, hipStream_t st);

// InstanceNorm + LeakyReLU from fp64 column sums for up to four tensors of one width in one launch (pointops.hip)
struct NormJob {
    const float* x; const double* sums; const float* res; const double* res_sums; float* y;
    const float* s_pts; float4* pk;      // pack form only: the KPConv support records of the output rows
    int n; double count;
    float* stats_out = nullptr;          // optional: receives the (mean, rstd) pairs [c][2] the sums stand for (the train tape keeps them)
};
int instnorm_apply_sums_multi(const NormJob* jobs, int count, int c, int ldx, float eps, int ldr, float slope, int ldy, bool pack,
                              hipStream_t st);

struct GatherJob { const float* x; const int64_t* idx; float* out; int ns, nq, h, ld_idx; };
int gather_max_multi(const GatherJob* jobs, int count, int c, hipStream_t st);
int heads_multi(const float* const* x, const int* rows, int count, int ld, int fd, float* const* feats, float* const* s_ov,
                float* const* s_sal, hipStream_t st);
int copy2d_multi(const float* const* src, float* const* dst, const int* rows, int count, int ld_src, int ld_dst, int cols,
                 hipStream_t st);
int instnorm_colsums_multi(const float* const* x, double* const* sums, const int* n, int count, int c, int ldx, hipStream_t st);

// DGCNN edge convolution (gnn.hip): emax + InstanceNorm2d statistics of up to four clouds in one launch (k_edgeconv_rows)
struct EdgeCloud {
    const float* ctr; const float* nbr; const int* idx; float* emax; double* sums; int n, k;
};
bool edgeconv_rows_ok(const EdgeCloud* cl, int count, int ld_ctr, int ld_nbr, int ld_emax, int c);
int edgeconv_rows_multi(const EdgeCloud* cl, int count, int ld_ctr, int ld_nbr, int ld_emax, int c, bool atomic, int* nchunks_out,
                        hipStream_t st);

// multi-head attention on the fp32 matrix cores (gnn.hip): up to four clouds in one launch (k_attention_mfma)
struct AttnCloud { const float* q; const float* k; const float* v; float* out; int n, ms; };
bool attention_mfma_ok(const AttnCloud* cl, int count, int ldq, int ldk, int ldv, int d);
int attention_mfma_multi(const AttnCloud* cl, int count, int ldq, int ldk, int ldv, int ldo, int heads, int d, float scale,
                         hipStream_t st);
// its backward (train step): gradients ADDED to dq / dk / dv by float atomics; false: keys beyond what the score tile of a
// 32-query workgroup holds in LDS (1216), head widths other than 32 / 64 / 128, deterministic=1
bool attention_bwd_mfma_ok(int n, int ms, int d, int ldq, int ldk, int ldv, int ld_do);
int attention_bwd_mfma(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, int ldo,
                       const float* d_o, int ld_do, float* dq, int ld_dq, float* dk, int ld_dk, float* dv, int ld_dv, int n, int ms,
                       int heads, int d, float scale, hipStream_t st);

// Tuning / A-B switches of the library, in ONE place.  Every field defaults to the product behaviour; they are set by
// pcrcg_debug_set("name=value,name=value") or, once at first use, from the environment variable PCRCG_DEBUG (same
// syntax) -- include/pcrcg.h lists the names.  Nothing else in csrc/ reads the environment, except PCRCG_GEMM_MODE
// (the documented arithmetic selector, pcrcg_gemm_set_mode).
struct DebugOpts {
    int zero_arena = 1;        // runner: split-K outputs from one pre-zeroed arena (0: each product clears its own C)
    int stat_sums = 1;         // runner: InstanceNorm statistics as fp64 column sums added by the GEMM epilogues
    int stat_sums_rows = 1 << 30;   //   ... only for outputs of up to that many rows (above: deterministic partials)
    int fuse_norm = 1;         // runner: normalise-on-load inside the consuming product
    int fuse_pack = 1;         // runner: the normalisation that feeds a KPConv also packs its support records
    int fuse_upsample = 1;     // runner: nearest_upsample -> cat(skip) -> unary as two products into one output
    int knock_tail = 0;        // MEASUREMENT AID (wrong results): resnet blocks of layers < knock_tail skip the shortcut product and its half of the closing pass -- the traffic a fused block tail would save
    int c1_rows16 = 1;         // runner: the first layer's (cin = 1) aggregate in rows of 16 floats, its contraction the grouped A B^T
                               // product with epilogue statistics (round 6); 0: rows of 15, one k-major launch per pair + a column-sum pass
    int gnn_merge = 1;         // runner: source and target clouds of a self-attention layer through ONE pass (round 5); needs
                               // 2 x pairs <= 4 clouds per launch, i.e. forward calls of one or two pairs (include/pcrcg.h)
    int edge_rows = 1;         // edge convolution: the row-parallel multi-cloud kernel (0: the per-cloud chunked kernel)
    int att_mfma = 1;          // attention: the fp32-MFMA kernel, all clouds of a call in one launch (0: the VALU kernel per cloud)
    int radius_blocks = 0;     // radius search grid (0: 512 workgroups)
    int radius_eager_redo = 0; // pyramid builder: launch the >256-hit redo pass unconditionally
    int kd_blocks = 0;
    int radius_prof = 0;       // cell-cooperative search: per-phase shader-cycle counters, printed at exit (measurement aid)
    int radius_cells = 1;      // pyramid builder: cell-cooperative LDS-staged search (0: the per-query kernel of rounds 1-3)
    int pyr_wait = 1;          // pyramid builder host round trip: 0 stream sync, 1 event, 2 device-posted flag
    int pyr_trace = 0;         // pyramid builder: host enqueue / wait microseconds at exit
    int pyr_morton = 0;        // MEASUREMENT AID: every subsampled level sorted along a Z curve before anything reads it (the level
                               // rows are then not the reference's: a knock-out that prices an internal spatial order, morton_knock.hip)
    int att_tq = 16;           // attention kernel: queries per workgroup (8 or 16)
    int kd_spin_limit = 0;     // KD-forest task queue: spin bound (0: default)
    int gemm_log = 0;          // print every GEMM's shape and grid
    int x6_tile = -1, x6_splitk = 0, x6_t1 = 1, x6_t2 = 1, x6_order = -1, x6_big = 0, x6_h2 = 1;   // split-bf16 GEMM plan overrides
    int train_side_stream = 1; // train-step backward: weight-gradient products on a second stream
    int gemm_tile = -1, gemm_splitk = 0, gemm_split_target = 768;        // fp32-MFMA GEMM plan overrides
    // deterministic=1: results that are a function of the inputs alone, bit for bit, run after run -- no floating-point
    // atomics anywhere on the path: split-K products store their partial tiles and add them in split order (a second pass,
    // gemm_x6.hip; the fp32-MFMA kernel does not split: gemm_splitk = 1), InstanceNorm statistics come from stored partials
    // and a fixed-order finishing pass (stat_sums = 0), and the train step's scatter kernels accumulate in 64-bit fixed
    // point (integer addition is associative; trainops.hip).  Slower (DESIGN.md has the price); the default keeps the
    // atomics.  Setting it overrides the two switches it implies.
    int deterministic = 0;
    int bwd_mfma = 1;               // KPConv backward's scatter on the matrix cores (k_kpconv_bwd_dx_mfma); 0: the VALU kernel
};
const DebugOpts& debug_opts();

// Optional start / stop events of a KPConv kernel (bench.py roofline); see pcrcg_profile_kpconv in
// include/pcrcg.h.  The events are handed to hipExtLaunchKernelGGL, so they stamp the kernel's own begin and
// end (what rocprofv3 reports), not the time its dispatch waited behind other streams.  a/b are NULL when
// profiling is off (a plain launch).  kind 0 = gather/aggregate kernel, 1 = fused kernel, 2 = bf16-storage gather kernel,
// 3 = a GEMM of the k_gemm_x6 family (nq = M, h = N, cin = K, cout = bf16 products per element: 6, or 3 with a bf16 A),
// 4 = a radius search (nq = queries, h = columns, cin = supports, cout = 1 cell-cooperative kernel / 0 per-query kernel);
// pcrcg_profile_kpconv(flags): 1 = KPConv kernels, 2 = GEMMs, 4 = radius searches.
struct KpProfScope {
    hipStream_t st;
    hipEvent_t a, b;
    int nq, h, cin, cout, kind;
    bool on;
    KpProfScope(hipStream_t s, int nq, int h, int cin, int cout, int kind);
    ~KpProfScope();
};

}  // namespace pcrcg

#define PCRCG_CHECK_ARG(cond)                                                        \
    do {                                                                             \
        if (!(cond)) {                                                               \
            pcrcg::set_error("%s: bad argument: %s", __func__, #cond);               \
            return PCRCG_EBADARG;                                                    \
        }                                                                            \
    } while (0)

#define PCRCG_CHECK_WS(carver)                                                                     \
    do {                                                                                           \
        if (!(carver).ok()) {                                                                      \
            pcrcg::set_error("%s: workspace too small (%zu needed, %zu given)", __func__,          \
                             (carver).off, (carver).cap);                                          \
            return PCRCG_EWORKSPACE;                                                               \
        }                                                                                          \
    } while (0)

#define PCRCG_CHECK_HIP(expr)                                                                  \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            pcrcg::set_error("%s: %s failed: %s", __func__, #expr, hipGetErrorString(e_));     \
            return PCRCG_ELAUNCH;                                                              \
        }                                                                                      \
    } while (0)

#define PCRCG_CHECK_LAUNCH() PCRCG_CHECK_HIP(hipGetLastError())

// A kernel that may ask for more than 64 KB of dynamic LDS is granted, ONCE per kernel and under std::call_once, everything a
// gfx950 workgroup can have (160 KB minus the kernel's static LDS).  The grant only lifts the launch-time limit -- what a
// launch allocates is its own `lds` argument -- and it never shrinks, so host threads that launch the same kernel with
// different sizes (the pair engine's model threads) cannot undo one another's setting.
namespace pcrcg {
inline hipError_t grant_max_dyn_lds(const void* kern) {
    hipFuncAttributes at;
    hipError_t e = hipFuncGetAttributes(&at, kern);
    if (e != hipSuccess) return e;
    const int cap = 160 * 1024 - (int)at.sharedSizeBytes;
    return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
}
}  // namespace pcrcg
#define PCRCG_GRANT_LDS(kern)                                                                                   \
    do {                                                                                                        \
        static std::once_flag once_;                                                                            \
        static hipError_t granted_ = hipSuccess;                                                                \
        std::call_once(once_, [&] { granted_ = pcrcg::grant_max_dyn_lds(reinterpret_cast<const void*>(kern)); }); \
        PCRCG_CHECK_HIP(granted_);                                                                              \
    } while (0)

#define PCRCG_PROPAGATE(expr)      \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != PCRCG_OK) return rc_; \
    } while (0)
