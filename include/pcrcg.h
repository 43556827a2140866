/* pcrcg.h -- C ABI of libpcrcg_hip.so, the MI355X (gfx950) implementation of PCR-CG's
 * feature-extraction hot path.
 *
 * The reference has no C-level FFI for this path: its native boundary is two CPython extension
 * modules (NumPy C-API glue) and, for the model, stock PyTorch ops.  Every entry point below names
 * the reference interface it replaces; INTEGRATION.md shows the binding a maintainer of the
 * reference would add (a ctypes stub, since the reference's host language is Python).
 *
 * Conventions
 *   - plain pointers and sizes only; every data pointer is a DEVICE pointer unless the parameter
 *     name starts with `h_`;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued on
 *     it and nothing synchronises unless documented;
 *   - no hidden allocation: scratch memory is a caller-provided workspace whose size comes from the
 *     matching *_ws_bytes() query; workspaces may be reused across calls on the same stream;
 *   - return value: PCRCG_OK (0) or a negative PCRCG_E* code; pcrcg_last_error() gives the
 *     message of the calling thread's last failure;
 *   - index tables use the reference's batched-neighbour contract: int64, row-major [Nq, cols],
 *     shadow (padding) value == number of stacked support points (ref:neighbors.cpp:319-325).
 */
#ifndef PCRCG_H
#define PCRCG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCRCG_OK 0
#define PCRCG_EBADARG (-1)   /* null pointer / negative size / unsupported shape */
#define PCRCG_EWORKSPACE (-2) /* workspace too small */
#define PCRCG_ELAUNCH (-3)   /* HIP runtime error while enqueuing */
#define PCRCG_ECAPACITY (-4) /* device-side capacity overflow reported by pcrcg_check_status */

#define PCRCG_KPOINTS 15 /* kernel points per KPConv (ref:configs/test/indoor.yaml num_kernel_points) */

const char* pcrcg_last_error(void);
/* ABI version of this header: bumped on any signature change, on any change of a caller-held workspace's layout or size,
 * and on any change of the default arithmetic.  A binding compares pcrcg_abi_version() (what the loaded library was built
 * from) with the PCRCG_ABI_VERSION it was written against (pcrcg_amd/_lib.py does, at load time).
 *   2 (round 3)  forward groups, train-step runner, correspondences, debug switches; one-kernel KPConv entries removed
 *   3 (round 4/5) cell-grid workspace: 64-bit run cursor, 16-byte slots, compact cell list and ticket block (a grid built
 *                by a version-2 library cannot be walked); new entries pcrcg_radius_query_cells,
 *                pcrcg_pyramid_build_parts, pcrcg_gemm_f32_grad, pcrcg_thread_shares_gpu, pcrcg_gather_jobs;
 *                pcrcg_profile_kpconv flag bits;
 *                forward products in the fp16 two-term form with both range ends handled in the kernel (see
 *                pcrcg_gemm_set_mode); deterministic=1 debug switch
 *   4 (round 6)  pyramid builder: levels sized from pcrcg_pyramid_cfg::shrink, ONE host round trip per call, the KD-forests
 *                (level 0's, and one over the subsampled levels) on pcrcg_pyramid_cfg::side_stream (new cfg fields; pcrcg_pyramid_ws_bytes lost its `shrink`
 *                argument; pcrcg_pyramid_restore and pcrcg_reorder_job changed layout) */
#define PCRCG_ABI_VERSION 4
int pcrcg_abi_version(void);

/* Tuning / A-B switches, for measurements only: "name=value,name=value" (NULL: back to what the process started with --
 * the defaults, or what PCRCG_DEBUG made of them).  The same string is
 * read once from the environment variable PCRCG_DEBUG at first use; nothing else in the library reads the environment
 * except PCRCG_GEMM_MODE (pcrcg_gemm_set_mode).  Every switch defaults to the product behaviour:
 *   zero_arena=1 stat_sums=1 stat_sums_rows=2^30 fuse_norm=1 fuse_pack=1 fuse_upsample=1 gnn_merge=1 edge_rows=1 att_mfma=1 c1_rows16=1   network runner fusions
 *                  (gnn_merge: the source and target clouds of a self-attention layer through one pass and both query
 *                  projections of a cross layer in one launch -- applies to forward calls of ONE or TWO pairs: the merged
 *                  pass holds 2 x pairs clouds and the multi-cloud kernels take four; with four pairs per call the two
 *                  sides run as two passes of four clouds.  The fused key / value projection and the once-per-forward kNN
 *                  graphs apply at every group size.)
 *   radius_blocks=0 radius_eager_redo=0 radius_cells=1 radius_prof=0 pyr_wait=1 pyr_trace=0 kd_spin_limit=0 kd_blocks=0   front end
 *   pyr_morton=0   1: MEASUREMENT AID -- every subsampled level sorted along a Z curve before anything reads it; the level rows
 *                  are then not the reference's (a knock-out that prices an internal spatial order: csrc/morton_knock.hip)
 *   att_tq=16                                                                              attention tile
 *   gemm_log=0 x6_tile=-1 x6_splitk=0 x6_t1=1 x6_t2=1 x6_order=-1 x6_big=0 x6_h2=1 gemm_tile=-1 gemm_splitk=0 gemm_split_target=768   GEMM plans
 *   train_side_stream=1 bwd_mfma=1                                                         train-step backward
 *   deterministic=0    1: bit-reproducible results -- no floating-point atomics (split-K partial tiles stored and added in
 *                      split order by a second pass, InstanceNorm statistics from stored partials, fixed-point scatter sums
 *                      in the train step); implies stat_sums=0 gemm_splitk=1; allocates its scratch itself.  Together with a fixed pairing of the pair engine (PairStreams(adaptive_jobs=False))
 *                      outputs are a function of the inputs alone.
 * Returns PCRCG_EBADARG (and changes nothing) on an unknown name. */
int pcrcg_debug_set(const char* spec);
/* deterministic=1 allocates its scratch itself (partial tiles of split-K products, 64-bit fixed-point sums of the train
 * step's scatters): one buffer per stream it has run on, keyed by the stream handle, grown on demand and kept until the
 * process ends.  This frees all of it (after a device-wide synchronise): call it before destroying streams the mode has
 * used -- a recycled handle would otherwise inherit a stale entry -- or to get the memory back. */
int pcrcg_debug_release(void);

/* ------------------------------------------------------------------------------------------------
 * Front end: grid subsampling
 * Replaces cpp_wrappers.cpp_subsampling.grid_subsampling.subsample_batch(points, batches,
 * sampleDl=, max_p=) (zip:cpp_subsampling/wrapper.cpp:62-330) whose core is batch_grid_subsampling
 * (zip:cpp_subsampling/grid_subsampling/grid_subsampling.cpp:109-211).
 *   pts      [n,3] f32 stacked clouds           len      [nb] i32 points per cloud
 *   out_pts  [n,3] f32 (capacity n rows)        out_len  [nb] i32 cells kept per cloud
 *   out_m    [1]   i32 total rows written (= sum(out_len)); read it back after the stream drains
 * Output rows are the cell barycentres in the exact order the reference emits them (libstdc++
 * unordered_map iteration order per cloud), bit-identical in fp32.
 * ---------------------------------------------------------------------------------------------- */
size_t pcrcg_grid_subsample_ws_bytes(int n, int nb);
int pcrcg_grid_subsample_batch(const float* pts, int n, const int* len, int nb, float dl, int max_p,
                               float* out_pts, int* out_len, int* out_m, void* ws, size_t ws_bytes,
                               void* stream);

/* Iteration order of a libstdc++ std::unordered_map<size_t,T> (identity hash) after inserting the
 * m DISTINCT keys in order; order[j] = insertion rank of the j-th visited element.  This is the
 * ordering step of the subsampling above, exported for unit tests
 * (zip:.../grid_subsampling.cpp:48,59-61,85).  keys [m] u64, order [m] i32. */
size_t pcrcg_umap_order_ws_bytes(int m);
int pcrcg_umap_order(const uint64_t* keys, int m, int* order, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Front end: radius neighbours
 * Replaces cpp_wrappers.cpp_neighbors.radius_neighbors.batch_query(queries, supports, q_batches,
 * s_batches, radius=) (ref:cpp_wrappers/cpp_neighbors/wrapper.cpp:58-238) whose core is
 * batch_nanoflann_neighbors (ref:cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-333), plus
 * the `[:, :max_neighbors]` truncation and `.long()` cast of batch_neighbors_kpconv
 * (ref:datasets/dataloader.py:54-69, 347-349).
 *
 * The search structure (a hashed uniform cell grid over the supports, cell edge = radius) is built
 * once per support set and can serve several query sets of the same radius: in the KPConv pyramid
 * the conv, pool and upsample searches over one level all use that level's radius
 * (ref:datasets/dataloader.py:273,298,301).
 *   sup [ns,3] f32, slen [nb] i32, grid = workspace of pcrcg_cellgrid_ws_bytes(ns, nb) bytes.
 * ---------------------------------------------------------------------------------------------- */
size_t pcrcg_cellgrid_ws_bytes(int ns, int nb);
int pcrcg_cellgrid_build(const float* sup, int ns, const int* slen, int nb, float radius, void* grid,
                         size_t grid_bytes, void* stream);
/*   q [nq,3] f32, qlen [nb] i32; out_idx [nq, cols] i64: row = supports with d2 < radius^2 (fp32,
 *   unfused) in ascending (d2, index) order, truncated to `cols`, padded with ns;
 *   out_count [nq] i32 (may be NULL) = untruncated list length per query;
 *   out_max_count [1] i32 = max over queries (the reference's column count), accumulated with
 *   atomicMax: zero it (or let pcrcg_radius_neighbors_batch do so) before the first query call;
 *   status [1] i32 (may be NULL): set non-zero if a list exceeded the kernel's staging capacity. */
int pcrcg_radius_query(const float* q, int nq, const int* qlen, int ns, const int* slen, int nb,
                       float radius, const void* grid, int cols, int64_t* out_idx, int* out_count,
                       int* out_max_count, int* status, void* stream);
/* pcrcg_radius_query that also reports the rows whose REFERENCE order is not the (d2, index) order: rows that hold
 * two neighbours of EXACTLY equal d2, the first of them inside the kept `cols` columns.
 *   out_tie_rows [nq] i32: the row numbers (unordered), out_tie_count [1] i32: how many -- accumulated with
 *   atomicAdd, zero it before the call.  Both NULL = pcrcg_radius_query. */
int pcrcg_radius_query_ex(const float* q, int nq, const int* qlen, int ns, const int* slen, int nb,
                          float radius, const void* grid, int cols, int64_t* out_idx, int* out_count,
                          int* out_max_count, int* status, int* out_tie_rows, int* out_tie_count, void* stream);
/* pcrcg_radius_query_ex for several INDEPENDENT groups of clouds stacked into one call (e.g. two fragment pairs =
 * four clouds, group = 2): indices are written relative to the first support of the query's group, rows are padded
 * with the group's support count, and out_max_count is an array with one entry per group ([ceil(nb / group)], zeroed
 * by the caller) -- the table is the groups' own tables stacked on top of each other, row for row what separate
 * calls would have written.  group = 0: one group (pcrcg_radius_query_ex). */
int pcrcg_radius_query_groups(const float* q, int nq, const int* qlen, int ns, const int* slen, int nb, int group,
                              float radius, const void* grid, int cols, int64_t* out_idx, int* out_count,
                              int* out_max_count, int* status, int* out_tie_rows, int* out_tie_count, void* stream);
/* The CELL-COOPERATIVE search (round 4; the kernel the pyramid builder uses): the queries are walked cell by cell through
 * a grid of their own, `qgrid` = pcrcg_cellgrid_build over the QUERY points (any radius: it only sets the size of the
 * query cells; a level's conv grid serves), nb clouds matching the support grid's.  One workgroup per occupied query cell
 * resolves the hash probes of the support cells within reach once, stages their candidates and the cell's queries in LDS,
 * and its wavefronts answer the cell's queries from LDS -- same result set, order, padding, counts, tie rows and group
 * semantics as pcrcg_radius_query_groups, row for row (replaces the hot loop of
 * ref:cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:268-301 and :319-325).  Rows of more than 128 hits and cells
 * whose neighbourhood exceeds the staging capacity are finished by the per-query kernel inside the same call
 * (status bit 4 is set and cleared on the way; q / qlen / slen are what that pass reads).
 * The kernel's work counters live in `qgrid` (it leaves them zeroed): searches that walk the SAME query grid must be
 * ordered on one stream (or by events) -- two of them running at once would share the counters.  Searches over
 * different query grids, and any number of searches reading the same SUPPORT grid, may overlap freely. */
int pcrcg_radius_query_cells(const void* qgrid, const float* q, int nq, const int* qlen, const void* sgrid, int ns,
                             const int* slen, int nb, int group, float radius, int cols, int64_t* out_idx, int* out_count,
                             int* out_max_count, int* status, int* out_tie_rows, int* out_tie_count, void* stream);
/* Convenience: zero out_max_count/status, build the grid in `ws`, run one query. */
size_t pcrcg_radius_neighbors_ws_bytes(int ns, int nb);
int pcrcg_radius_neighbors_batch(const float* q, int nq, const float* sup, int ns, const int* qlen,
                                 const int* slen, int nb, float radius, int cols, int64_t* out_idx,
                                 int* out_count, int* out_max_count, int* status, void* ws,
                                 size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Front end: the reference's order inside groups of exactly equal distance
 * batch_nanoflann_neighbors sorts its hits by distance only (IndexDist_Sorter, zip:cpp_utils/nanoflann/
 * nanoflann.hpp:208-214) with an unstable std::sort over nanoflann's traversal order, and
 * batch_neighbors_kpconv then cuts the rows at `[:, :limit]` (ref:datasets/dataloader.py:65-69): which members of
 * a tie group survive, and which of two equidistant coarse points lands in column 0 of an upsample table, is
 * decided by that order.  These calls reproduce it entry for entry (pcrcg_amd/csrc/tieorder.hip):
 *
 *   pcrcg_kdforest_build   nanoflann 1.3.0's KD-tree (leaf size 10, ref:.../neighbors.cpp:245) for each of the
 *                          nb clouds of `sup` -- the clouds of ALL pyramid levels can be stacked into one forest.
 *                          forest = workspace of pcrcg_kdforest_ws_bytes(ns, nb) bytes; one launch.
 *   pcrcg_radius_reorder   rewrites rows of a table produced by pcrcg_radius_query(_ex): the query clouds
 *                          0..nbq-1 search the forest's clouds cloud0..cloud0+nbq-1; indices are written relative
 *                          to the first of them and rows are padded with their total size (the table's shadow
 *                          index).  rows [nrows] i32 = the rows to redo (out_tie_rows; NULL = all nq rows);
 *                          count [nq] i32 (may be NULL) = out_count of the query, cross-checked;
 *                          max_count = staging width (>= the longest list among `rows`, <= 8192);
 *                          idx [nq, cols] i64.
 *   pcrcg_radius_reorder_jobs  the same for up to PCRCG_MAX_REORDER_JOBS tables over one forest in ONE launch
 *                          (all tables of a pair).
 *                          status [1] i32 (may be NULL): 1 forest kernel gave up waiting for work, 2 traversal stack
 *                          overflow (more than 128 pending branches), 3 hit count differs from `count`,
 *                          4 list longer than max_count.
 * ---------------------------------------------------------------------------------------------- */
#define PCRCG_MAX_REORDER_JOBS 12
typedef struct pcrcg_reorder_job {
    const float* q;       /* [nq,3] queries of the table */
    const int* qlen;      /* [nbq] */
    const int* rows;      /* [nrows] rows to redo, NULL = all */
    const int* count;     /* [nq] or NULL */
    int64_t* idx;         /* [nq, cols] table, rewritten in place */
    int nq, nbq, cloud0, nrows, max_count, cols;
    float radius;
    int group;            /* > 0: the nbq clouds are independent groups of `group` clouds (pcrcg_radius_query_groups):
                             indices relative to the group's first support, padding = the group's support count */
    const float* sup;     /* version 4: forest != NULL: this job searches a forest of its OWN -- built over `sup` with */
    const void* forest;   /*   pcrcg_kdforest_build(sup, forest_ns, .., forest_nb, ..) -- instead of the call's (the  */
    int forest_ns, forest_nb; /* pyramid builder keeps one forest per level); NULL: the call's sup / forest / ns / nb */
} pcrcg_reorder_job;
size_t pcrcg_kdforest_ws_bytes(int ns, int nb);
int pcrcg_kdforest_build(const float* sup, int ns, const int* slen, int nb, void* forest, size_t forest_bytes,
                         void* stream);
int pcrcg_radius_reorder(const float* q, int nq, const int* qlen, int nbq, const float* sup, int ns, int nb,
                         const void* forest, int cloud0, float radius, const int* rows, int nrows, const int* count,
                         int max_count, int cols, int64_t* idx, int* status, void* stream);
int pcrcg_radius_reorder_jobs(const pcrcg_reorder_job* jobs, int njobs, const float* sup, int ns, int nb,
                              const void* forest, int* status, void* stream);

/* ------------------------------------------------------------------------------------------------
 * KPConv (rigid, linear influence, sum aggregation) -- replaces KPConv.forward
 * (ref:models/blocks.py:229-374):
 *   out[q,:] = (1/n_q) * sum_k ( sum_h max(0, 1 - |s[idx[q,h]] - q_pts[q] - kp[k]| / extent)
 *                                 * x[idx[q,h],:] ) @ W[k]
 *   n_q = max(1, #{h : sum_c x[idx[q,h],c] > 0}); idx == ns addresses the shadow support
 *   (point 1e6,1e6,1e6; feature 0).
 * Stage 1 (this call) gathers and aggregates:  wf [nq, 15*cin] f32 (kernel-point major: column
 * k*cin + c) and inv_n [nq] f32 = 1/n_q.  Stage 2 is the dense contraction wf @ W[15*cin, cout]
 * scaled per row by inv_n (pcrcg_gemm_f32 with row_scale).
 *   q_pts [nq,3], s_pts [ns,3], idx [nq, h] i64 (row stride ld_idx elements), x [ns, cin] f32,
 *   kp [15,3] f32.
 * ---------------------------------------------------------------------------------------------- */
size_t pcrcg_kpconv_ws_bytes(int ns);
int pcrcg_kpconv_aggregate(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx,
                           int h, int ld_idx, const float* x, int cin, const float* kp, float extent,
                           float* wf, float* inv_n, void* ws, size_t ws_bytes, void* stream);
/* Stage 1 of the bf16 feature-storage VARIANT (pcrcg_model.feature_bf16): x is fp32 at the boundary; x_bf16
 * ([ns, cin] u16 scratch) receives its round-to-nearest-even bf16 copy, which is what the neighbour gathers read, and
 * wf_bf16 ([nq, 15*cin] u16) the aggregate rounded the same way.  Geometry, influences, n_q and the accumulation are
 * fp32.  cin % 4 == 0.  Stage 2 is pcrcg_gemm_bf16a_f32_colstats. */
int pcrcg_kpconv_aggregate_bf16(const float* q_pts, int nq, const float* s_pts, int ns, const int64_t* idx, int h,
                                int ld_idx, const float* x, int cin, const float* kp, float extent, void* x_bf16,
                                void* wf_bf16, float* inv_n, void* ws, size_t ws_bytes, void* stream);

/* Measurement aid for bench.py: when enabled, the gather/aggregate kernel of every
 * pcrcg_kpconv_aggregate call (kind 0; kind 1 is reserved for a one-kernel KPConv)
 * are launched with HIP start / stop events (hipExtLaunchKernel) on their own stream, i.e. the kernel's own
 * execution time as rocprofv3 reports it, excluding the time its dispatch waited behind other streams; _read
 * waits for them and returns up to `cap` records (milliseconds, nq / h / cin of the launch, cout for kind 1).
 * `enable` is a bit mask: 1 = the KPConv kernels (kinds 0, 1, 2 = bf16-storage gather), 2 = every GEMM of the
 * split-bf16 family (kind 3: nq = M, h = N, cin = K, cout = bf16 matrix products per element, 6 or 3), 0 = off.
 * Safe to call from several host threads; off by default. */
void pcrcg_profile_kpconv(int enable);
int pcrcg_profile_kpconv_read(float* ms, int* nq, int* h, int* cin, int* cout, int* kind, int cap);

/* C[m,n] = (A[m,k] @ Bop[k,n]) * row_scale[m] + bias[n]   (fp32 in, fp32 MFMA accumulate; row_scale
 * and bias may be NULL).  Row-major with leading dimensions in elements.
 *   trans_b = 0: b is [k,n] (ldb >= n), Bop = b      -- KPConv weights [15*cin, cout], P @ V
 *   trans_b = 1: b is [n,k] (ldb >= k), Bop = b^T    -- nn.Linear / 1x1 conv weights [out,in], Q @ K^T
 * Replaces the torch.matmul / nn.Linear / 1x1 nn.Conv1d contractions of the path
 * (ref:models/blocks.py:361-366, 487; ref:models/architectures.py:528,538-539; ref:models/gcn.py:123-173). */
int pcrcg_gemm_f32(const float* a, int lda, const float* b, int ldb, int trans_b, float* c, int ldc, int m,
                   int n, int k, const float* row_scale, const float* bias, void* stream);
/* Same GEMM; additionally, when every output element is written exactly once (no split-K), the
 * epilogue leaves per-column partial sums / sums of squares of C in `colstats`
 * (pcrcg_gemm_colstats_bytes(m, n) bytes, layout [2][n][chunks] fp64) and stores the chunk count in the
 * HOST integer *h_chunks; otherwise *h_chunks = 0 and the caller computes the statistics with
 * pcrcg_instnorm_stats.  Every GEMM of the path feeds an InstanceNorm (ref:models/blocks.py:456-463), so
 * this saves one full read of C per layer. */
size_t pcrcg_gemm_colstats_bytes(int m, int n);
int pcrcg_gemm_f32_colstats(const float* a, int lda, const float* b, int ldb, int trans_b, float* c, int ldc,
                            int m, int n, int k, const float* row_scale, const float* bias, void* colstats,
                            size_t colstats_bytes, int* h_chunks, void* stream);
/* C = (A @ B^T) * row_scale + bias with A stored as bf16 ([m, k] u16, lda in ELEMENTS, lda % 8 == 0, 16-byte aligned,
 * k % 32 == 0) and B fp32 [n, k]: B is split exactly into three bf16 terms as in mode 1 below and the three products
 * against the single A term run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- exact in B, bf16-rounded in A
 * only by A's storage format.  Column partials as pcrcg_gemm_f32_colstats. */
int pcrcg_gemm_bf16a_f32_colstats(const void* a_bf16, int lda, const float* b, int ldb, float* c, int ldc, int m, int n,
                                  int k, const float* row_scale, const float* bias, void* colstats,
                                  size_t colstats_bytes, int* h_chunks, void* stream);

/* C (+)= f(A)[rows] @ B^T + bias with B [n, k] k-contiguous: the products with which pcrcg_kpfcnn_forward folds a
 * neighbouring operator into a GEMM's A loads instead of running it as a pass of its own.
 *   idx != NULL    output row r uses row idx[r * ld_idx] of A [ns, k] (lda), the zero row `zero_row` (>= k floats of 0)
 *                  when that index is outside [0, ns) -- nearest_upsample with its shadow index
 *                  (ref:models/blocks.py:77-87); with accumulate the decoder's nearest_upsample -> cat(skip) -> unary
 *                  (ref:models/architectures.py:568-569) is two products into one output and neither the upsampled
 *                  matrix nor the concatenation is written;
 *   a_sums != NULL A is the RAW output of a product; its InstanceNorm + LeakyReLU (ref:models/blocks.py:456-470) is
 *                  applied on load, f(a) = lrelu((a - mean_k) * rstd_k, a_slope), from the fp64 column sums
 *                  a_sums [2][k] over a_count rows (biased variance, a_eps); a gathered shadow row stays zero;
 *   accumulate     != 0 adds the product to C (fp32 atomics) instead of storing it.
 * Split-bf16 arithmetic only (mode 1 below). */
int pcrcg_gemm_f32_fused(const float* a, int lda, const int64_t* idx, int ld_idx, int ns, const float* zero_row,
                         const void* a_sums, double a_count, float a_eps, float a_slope, const float* b, int ldb,
                         const float* bias, float* c, int ldc, int m, int n, int k, int accumulate, void* stream);

/* Arithmetic of the C = A @ B^T products (trans_b = 1) behind pcrcg_gemm_f32 / _colstats / _ex:
 *   0: v_mfma_f32_32x32x2_f32 on the fp32 operands (the fp32 matrix rate, 157 TF on MI355X);
 *   1: (default) the fp32 operands are split into 16-bit terms and the products run on the 16-bit matrix cores with
 *      fp32 accumulation.  The forward products (k-contiguous fp32 operands) use the TWO-term fp16 form
 *      x = h + 2^-11 l, three v_mfma_f32_32x32x16_f16 per 16-deep chunk (representation and dropped term: 2^-22
 *      relative per product; measured against float64 no worse than the bf16 form or mode 0).  fp16's normal range is
 *      2^-14 .. 65504, and BOTH ends are handled inside the kernel, per tile, with no flag for the host and no different
 *      result contract (round 5):
 *        above   an operand value beyond 65504 leaves a non-finite partial sum behind;
 *        below   every thread keeps the largest |x| of the values it splits, per tile row of A and of B (an output row /
 *                an output column); a row that is not all zeros and holds no value of at least 2^-14 is one whose h terms
 *                are all fp16 subnormals (absolute floor 2^-36 per value instead of 2^-22 relative);
 *      a workgroup that finds either after its loop throws its sums away and redoes its tile in the exact bf16 form
 *      below.  What the fp16 form keeps is therefore, for every finite fp32 operand: each element's representation error
 *      is at most 2^-22 of the LARGEST value of its row (not of the element itself: a 1e-9 entry beside O(1) entries of
 *      the same row is carried with an absolute error of 2^-36) -- a normwise-per-row fp32-class bound, measured
 *      <= 5e-7 of a float64 product at every whole-operand scale from 1 to 1e-12 (tests/test_gemm_range_gpu.py).
 *      The other products (the training rows' k-major forms without a named gradient operand, bf16-stored operands) and
 *      PCRCG_DEBUG=x6_h2=0 use the EXACT three-term bf16 form: three bf16 terms per value, the six leading cross products
 *      on v_mfma_f32_32x32x16_bf16, dropped terms below 2^-23 relative per product, at 16/6 of the fp32 matrix rate (the
 *      fp16 form: 16/3).
 * Process-wide (one atomic word); also read once from the environment variable PCRCG_GEMM_MODE.  Interface: fp32 in,
 * fp32 out in both modes.
 * Non-finite operands: mode 1 turns an operand value of +-inf into NaN in every output it touches (the split's residual
 * is inf - inf), where mode 0 / an fp32 GEMM would produce +-inf or, against a zero, NaN as well; NaN operands give NaN in
 * both modes.  Finite operands keep fp32's range in both forms: the bf16 form's three terms are exact for every finite
 * fp32 value whose magnitude is at least 2^-110 (below that the third term leaves fp32's own subnormal range), and the fp16
 * form hands every tile it cannot hold to it.  The train step's non-finite check (pcrcg_amd/trainer.py: the reference's
 * validate_gradient) treats inf and NaN alike, so the skip decision does not depend on the mode. */
void pcrcg_gemm_set_mode(int mode);
int pcrcg_gemm_get_mode(void);
/* Diagnostics of the fp16 form's range checks (synchronous; not for the hot path): out2[0] = tiles redone in the bf16 form
 * since the last reset because an operand value lay beyond fp16's range, out2[1] = because a row lay below it.  reset != 0
 * clears the counters afterwards.  out2 may be NULL (reset only). */
int pcrcg_gemm_redo_counts(unsigned long long* out2, int reset);
/* Declares that the CALLING HOST THREAD enqueues its network calls beside other streams that keep the GPU busy (on = 1;
 * 0 takes it back; per thread, default off).  The products of such a thread are planned WITHOUT split-K:
 * alone on the GPU a small product is split until ~200 workgroups exist, which fills the chip; beside other streams
 * their kernels fill the idle CUs anyway and every slice only costs fp32 atomics, a zeroed output and the column statistics
 * its epilogue could have left (+5 % pairs/s in the four-stream pair engine, whose model threads make this call; -1.9 %
 * for a forward that does run alone; round 4-5: a few slices, x6_t1 / x6_t2 = 32 / 128; round 6: none, 1 / 1).  Results
 * differ by summation order only.  No reference counterpart (ref:main.py:15 one process, one stream). */
void pcrcg_thread_shares_gpu(int on);

/* ------------------------------------------------------------------------------------------------
 * Point-wise blocks
 * ---------------------------------------------------------------------------------------------- */
/* max_pool (ref:models/blocks.py:86-102): out[q,c] = max_h xpad[idx[q,h],c], shadow row = 0. */
int pcrcg_gather_max(const float* x, int ns, int c, const int64_t* idx, int nq, int h, int ld_idx,
                     float* out, void* stream);
/* closest_pool (ref:models/blocks.py:71-83): out[q,:] = xpad[idx[q,0],:]; writes into a row-major
 * destination with leading dimension ld_out (so it can fill the left part of a concat buffer). */
int pcrcg_gather_first(const float* x, int ns, int c, const int64_t* idx, int nq, int ld_idx, float* out,
                       int ld_out, void* stream);
/* InstanceNorm over all rows + LeakyReLU (BatchNormBlock/UnaryBlock/ResnetBottleneckBlock,
 * ref:models/blocks.py:433-470, 493-500, 650-678):
 *   stats[2*c] receives per-channel (mean, 1/sqrt(var+eps)) of x [n,c] (biased variance);
 *   y = lrelu( (x - mean_x) * rstd_x + res_term, slope )
 *   res_term = 0                                   if res == NULL
 *            = res                                 if res_stats == NULL
 *            = (res - mean_r) * rstd_r             otherwise (res_stats from a previous call)
 * x, res, y are row-major with leading dimensions ldx, ldr, ldy.  slope 1.0 = no activation.
 * ws: pcrcg_instnorm_ws_bytes(c) bytes. */
size_t pcrcg_instnorm_ws_bytes(int c);
/* (mean, rstd) pairs from [2][c][chunks] fp64 partial sums over `count` rows (see
 * pcrcg_gemm_f32_colstats); fixed summation order, deterministic. */
int pcrcg_instnorm_stats_from_partials(const void* partials, int chunks, int c, double count, float eps,
                                       float* stats, void* stream);
int pcrcg_instnorm_stats(const float* x, int n, int c, int ldx, float eps, float* stats, void* ws,
                         size_t ws_bytes, void* stream);
int pcrcg_instnorm_apply(const float* x, int n, int c, int ldx, const float* stats, const float* res,
                         int ldr, const float* res_stats, float slope, float* y, int ldy, void* stream);
/* The same with the statistics given as fp64 column SUMS: sums [2][c] = (sum_r x[r,ch], sum_r x[r,ch]^2) over `count`
 * rows (what pcrcg_kpfcnn_forward's GEMM epilogues leave), mean / biased variance / rstd
 * derived on the fly -- no finishing launch between the producing GEMM and the normalisation.  res_sums (may be NULL):
 * the residual is normalised by its own sums, else added as is.  c % 4 == 0 and c / 4 a divisor or a multiple of 256
 * (every width of the architecture); rows 16-byte aligned. */
/* sums [2][c] f64 += (column sums, column sums of squares) of x [n, c]; the caller zeroes `sums` (one launch: the
 * statistics pass for outputs whose producing GEMM could not leave them, e.g. split-K products). */
int pcrcg_instnorm_colsums(const float* x, int n, int c, int ldx, void* sums, void* stream);
int pcrcg_instnorm_apply_sums(const float* x, int n, int c, int ldx, const void* sums, double count, float eps,
                              const float* res, int ldr, const void* res_sums, float slope, float* y, int ldy,
                              void* stream);

/* ------------------------------------------------------------------------------------------------
 * PCR-CG's image-feature injection (ref:models/architectures.py:195-514): the geometric input feature (a column of
 * ones) becomes [N, c+1] = ones, and every 3-D point that projects into a colour image receives that pixel's 2-D
 * feature in columns 0..c-1 (column c stays 1):
 *     x[inds3d[j] + row_offset, 0:c] = fmap[:, inds2d[j,1], inds2d[j,0]] * valid[inds2d[j,0], inds2d[j,1]]
 * fmap [c, h, w] f32 is the 2-D backbone's output for one image (the backbone itself is outside the path; any
 * producer of such a map plugs in), valid [w, h] f32 its valid-pixel mask as the reference stores it (NULL: none,
 * the three-image branch :196-252), inds2d [n, 2] i64 = (column, row), inds3d [n] i64 = point index inside its cloud,
 * row_offset = 0 for the source cloud / the source cloud's size for the target cloud (:239-241).  The reference
 * writes image 3, 2, 1 in that order so that image 1 wins where projections overlap: call once per image in the
 * same order on the same stream.  pcrcg_fill2d writes the constant (ones) matrix first.
 * ---------------------------------------------------------------------------------------------- */
int pcrcg_fill2d(float* dst, int ld, int rows, int cols, float value, void* stream);
int pcrcg_inject_image_features(const float* fmap, int c, int h, int w, const float* valid, const int64_t* inds2d,
                                const int64_t* inds3d, int n, long row_offset, long n_rows, float* x, int ldx,
                                void* stream);

/* ------------------------------------------------------------------------------------------------
 * GNN head helpers (ref:models/gcn.py)
 * ---------------------------------------------------------------------------------------------- */
/* get_graph_feature's kNN (ref:models/gcn.py:15-34,48-51): dist = -2ab + a^2 + b^2 clamped at 1e-12,
 * the k+1 smallest by (dist, index), first dropped.  coords [n,3] f32 -> idx [n,k] i32. */
int pcrcg_knn(const float* coords, int n, int k, int* idx, void* stream);
/* DGCNN edge conv after splitting the 1x1 conv over cat(f_i, f_j - f_i) into a centre term and a
 * neighbour term (ref:models/gcn.py:61-62,123-129):  e[i,j,c] = ctr[i,c] + nbr[idx[i,j],c];
 *   emax[i,c] = max_j e[i,j,c];  stats[2*c] = (mean, rstd) of e over all (i,j) (InstanceNorm2d).
 * The caller finishes with pcrcg_instnorm_apply(emax, stats, slope 0.2) -- max commutes with the
 * monotone normalise + LeakyReLU.  ctr/nbr/emax row-major [n,c] with leading dims. */
size_t pcrcg_edgeconv_ws_bytes(int c);
int pcrcg_edgeconv_reduce(const float* ctr, int ld_ctr, const float* nbr, int ld_nbr, const int* idx,
                          int n, int k, int c, float eps, float* emax, int ld_emax, float* stats,
                          void* ws, size_t ws_bytes, void* stream);
/* The same reduction leaving the statistics as fp64 sums: sums [2][c] f64 (zeroed by the caller) += (sum, sum of squares)
 * of e over all (i,j); finish with pcrcg_instnorm_apply_sums(emax, ..., sums, count = n * k, ...).  One launch. */
int pcrcg_edgeconv_reduce_sums(const float* ctr, int ld_ctr, const float* nbr, int ld_nbr, const int* idx, int n, int k,
                               int c, float* emax, int ld_emax, void* sums, void* stream);
/* Row softmax in place: x [rows, cols] (ld), x = softmax(x * scale) (ref:models/gcn.py:151-155,
 * ref:models/architectures.py:562-563). */
int pcrcg_softmax_rows(float* x, int rows, int cols, int ld, float scale, void* stream);
/* Multi-head attention in ONE launch (ref:models/gcn.py:151-155: scores = q k^T / sqrt(d), prob = softmax(scores),
 * message = prob v, per head):  out[:, h*d:(h+1)*d] = softmax(scale * q_h k_h^T) v_h  for h < heads, where head h of
 * q [n, heads*d], k / v [ms, heads*d] and out [n, heads*d] is the column block [h*d, (h+1)*d) (row-major, leading
 * dimensions ldq/ldk/ldv multiples of 4, 16-byte aligned bases).  d in {16, 32, 48, 64, 128}
 * (pcrcg_attention_supported); other widths: one pcrcg_gemm_f32 / pcrcg_softmax_rows / pcrcg_gemm_f32 per head. */
int pcrcg_attention_supported(int d);
int pcrcg_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, int n,
                    int ms, int heads, int d, float scale, void* stream);
/* y[r * ldy] = sum_j softmax(scale * x[r, :])_j * vec[j * ldv]: the saliency scores
 * (ref:models/architectures.py:562-563, softmax(inner / T) @ scores) without storing the probabilities. */
int pcrcg_softmax_matvec(const float* x, int rows, int cols, int ld, float scale, const float* vec, int ldv, float* y,
                         int ldy, void* stream);

/* Copy a device status word to the host after draining `stream`; returns PCRCG_ECAPACITY if it is
 * non-zero. */
int pcrcg_check_status(const int* status, void* stream);

/* Small element-wise helpers used by the network runner (also exported for tests).
 *   copy2d : dst[r, 0:cols] = src[r, 0:cols]                       (torch.cat pieces)
 *   add    : dst = a + b                                          (residual of ref:models/gcn.py:213-214)
 *   l2norm : dst[r,:] = src[r,:] / max(|src[r,:]|, 1e-12)          (F.normalize, ref:architectures.py:541,582)
 *   scores : dst[r] = scrub(clamp(sigmoid(src[r*ld]), 0, 1))      (ref:architectures.py:176-179,576-579) */
int pcrcg_copy2d(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols, void* stream);
int pcrcg_add(const float* a, const float* b, float* dst, long n, void* stream);
int pcrcg_l2norm_rows(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols, void* stream);
int pcrcg_sigmoid_scores(const float* src, int ld_src, float* dst, int rows, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-network runner: KPFCNN.forward (ref:models/architectures.py:181-191, 516-610) enqueued by ONE
 * call, so that the host cost per pair is a single FFI crossing instead of several hundred.
 * The descriptors are plain C structs of device pointers and sizes; pcrcg_amd/runner.py builds them
 * from the nn.Module tree (weights as stored in the reference state_dict, plus a few re-packed copies
 * noted below).
 * ---------------------------------------------------------------------------------------------- */
#define PCRCG_MAX_LEVELS 8
#define PCRCG_MAX_BLOCKS 32
#define PCRCG_MAX_GNN 8

enum { PCRCG_BLK_SIMPLE = 0, PCRCG_BLK_RESNETB = 1, PCRCG_BLK_UNARY = 2, PCRCG_BLK_LAST_UNARY = 3,
       PCRCG_BLK_UPSAMPLE = 4 };

typedef struct pcrcg_block {
    int type;            /* PCRCG_BLK_* */
    int layer;           /* layer_ind of the reference block */
    int strided;         /* queries = level layer+1, table = pools[layer] */
    int in_dim, out_dim; /* feature widths seen by the block (out_dim = produced width) */
    int mid_dim;         /* KPConv width of a resnetb block (out_dim/4); KPConv output of a simple block */
    float extent;        /* KP_extent of the block's KPConv */
    const float* kp;     /* [15,3]  ...KPConv.kernel_points */
    const float* kp_w;   /* [15*cin, cout]  ...KPConv.weights */
    const float* kp_wt;  /* [cout, 15*cin]  K-contiguous copy: the contraction then is a C = A * B^T product, or NULL.
                            cin = 1: [cout, 16], the 16th column zero (the gather kernel then writes rows of 16 floats) */
    const float* kp_w_pad; /* [15*cin_pad, cout]: kp_w with the input channels zero-padded to cin_pad (a multiple of
                              4), or NULL.  Set when cin % 4 != 0 (PCR-CG's 129-channel first layer,
                              ref:models/architectures.py:195-514): the runner pads the features likewise, so the
                              MFMA gather kernel applies instead of the scalar fallback. */
    int cin_pad;
    const float* unary1; /* [mid, in] or NULL (nn.Identity) */
    const float* unary2; /* [out, mid] */
    const float* shortcut; /* [out, in] or NULL (nn.Identity) */
    const float* mlp;    /* unary / last_unary: [out, in] with leading dimension mlp_ld (rows 16-B aligned) */
    int mlp_ld;
    const float* mlp_skip; /* unary / last_unary that consumes cat([upsampled x, skip]) (model.dec_concat): a dense,
                              16-byte aligned copy of the weight's skip columns [out, skip_dim] (= mlp[:, in - skip_dim:]),
                              or NULL.  With it the runner never materialises the upsampled matrix or the concatenation:
                              it runs mlp[:, :in - skip_dim] on rows gathered through the upsample table and adds the
                              skip part's product (ref:models/blocks.py:77-87, ref:models/architectures.py:568-569). */
    int mlp_skip_ld, skip_dim;
} pcrcg_block;

typedef struct pcrcg_gnn_layer {
    int cross;             /* 0 = SelfAttention (ref:models/gcn.py:96-134), 1 = AttentionalPropagation (:176-185) */
    /* self: 1x1 conv weights [cout, 2*cin] = [Wa | Wb] re-packed as [2*cout, cin] = [Wa-Wb ; Wb] (rows: the centre
     * term's output channels, then the neighbour term's), k-contiguous like every other weight */
    const float* edge1;    /* [2c, c] */
    const float* edge2;    /* [4c, c] */
    const float* conv3;    /* [c, 4c] as stored ([out, in]) */
    /* cross: projection weights with output channels permuted head-major, [c, c] as [out, in] */
    const float *wq, *bq, *wk, *bk, *wv, *bv;
    const float *wm, *bm;  /* merge, input channels permuted head-major */
    const float *w0, *b0;  /* mlp.0 [2c, 2c] */
    const float *w3, *b3;  /* mlp.3 [c, 2c] */
} pcrcg_gnn_layer;

typedef struct pcrcg_model {
    int n_enc, n_dec, n_gnn;
    pcrcg_block enc[PCRCG_MAX_BLOCKS];
    pcrcg_block dec[PCRCG_MAX_BLOCKS];
    pcrcg_gnn_layer gnn[PCRCG_MAX_GNN];
    int enc_skip[PCRCG_MAX_BLOCKS];   /* 1 if the input of encoder block i is saved as a skip (:521-522) */
    int dec_concat[PCRCG_MAX_BLOCKS]; /* 1 if decoder block i consumes cat([x, skip]) (:568-569) */
    int enc_out_dim, gnn_dim, heads, knn_k, final_dim;
    const float *bottle_w, *bottle_b;         /* [gnn, enc_out], [gnn] */
    const float *proj_gnn_w, *proj_gnn_b;     /* [gnn, gnn], [gnn] */
    const float *proj_score_w, *proj_score_b; /* [1, gnn], [1] */
    float temperature;                        /* exp(epsilon) + 0.03 (:561) */
    int feature_bf16;                         /* 0: fp32 feature storage (the parity-tested path, default).  1: the
                                                 bf16 feature-storage VARIANT (BASELINE.json configs[1] "bf16/fp32"):
                                                 inside every KPConv with cin % 32 == 0 the neighbour gathers read a
                                                 bf16 copy of the features and the aggregated [nq, 15*cin] matrix is
                                                 bf16 in HBM; weights, accumulation, norms, outputs stay fp32.  NOT
                                                 within the 1e-4 parity bound: tests/test_bf16_gpu.py states its error */
} pcrcg_model;

typedef struct pcrcg_table { const int64_t* idx; int rows, cols, ld; } pcrcg_table;

typedef struct pcrcg_batch {
    int n_levels;
    const float* points[PCRCG_MAX_LEVELS];
    int n_points[PCRCG_MAX_LEVELS];
    pcrcg_table neighbors[PCRCG_MAX_LEVELS], pools[PCRCG_MAX_LEVELS], upsamples[PCRCG_MAX_LEVELS];
    const float* features; /* [n_points[0], feat_dim] */
    int feat_dim;
    int len_src_c;         /* stack_lengths[-1][0] */
    const int* stack_lengths[PCRCG_MAX_LEVELS]; /* [nb] i32 per level (device); filled by pcrcg_pyramid_build,
                                                   not read by pcrcg_kpfcnn_forward (may be NULL) */
} pcrcg_batch;

typedef struct pcrcg_outputs {
    float* feats_f;         /* [n0, final_dim] */
    float* scores_overlap;  /* [n0] */
    float* scores_saliency; /* [n0] */
} pcrcg_outputs;

size_t pcrcg_kpfcnn_ws_bytes(const pcrcg_model* model, const pcrcg_batch* batch);
int pcrcg_kpfcnn_forward(const pcrcg_model* model, const pcrcg_batch* batch, const pcrcg_outputs* out, void* ws,
                         size_t ws_bytes, void* stream);

/* The same forward for n (1 to 4) INDEPENDENT fragment pairs in one call: batches[n], outs[n] (e.g. the batches one
 * pcrcg_pyramid_build call with cfg.group = 2 returns for 2n clouds).  Pairs never mix -- InstanceNorm statistics, neighbour tables,
 * the GNN's kNN and attention stay per pair (ref:datasets/dataloader.py:207 asserts one pair per batch;
 * ref:models/blocks.py:456-463 normalises over the batch's stacked points), so the outputs are those of n separate
 * pcrcg_kpfcnn_forward calls up to fp32 summation order -- but every product with a weight matrix runs ONCE for all
 * pairs (their row ranges sharing the weight operand in one launch): the coarse-level products, which fill a fraction of
 * the chip for one pair, fill twice that at the same duration, and a pair costs half the GEMM launches.
 * Per-pair kernels (gathers, normalisation, kNN, attention) are enqueued once per pair.  With the fp32-MFMA arithmetic
 * (pcrcg_gemm_set_mode(0)) the products run pair by pair as well. */
size_t pcrcg_kpfcnn_group_ws_bytes(const pcrcg_model* model, const pcrcg_batch* batches, int n);
int pcrcg_kpfcnn_forward_group(const pcrcg_model* model, const pcrcg_batch* batches, const pcrcg_outputs* outs, int n,
                               void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pyramid builder: the whole front end of one fragment pair behind one call -- the pyramid loop of
 * collate_fn_descriptor (ref:datasets/dataloader.py:230-361: per level a conv table, then grid subsampling, a pool
 * and an upsample table; radius and cell size double per level) with batch_grid_subsampling_kpconv /
 * batch_neighbors_kpconv (:14-69) underneath.  It sequences the front-end entry points above over one arena and
 * fills the pcrcg_batch that pcrcg_kpfcnn_forward consumes; every pointer in `out` points into `ws`.
 *   cfg        level plan: r_conv / r_pool / dl per level (ref:datasets/dataloader.py:239,286,301,357), has_conv,
 *              pooled (every level but the last), limit = neighborhood_limits; tie_order 0 = ascending index inside
 *              groups of exactly equal distance, 1 = the reference's order (rows holding such groups are redone
 *              through the KD-forest, as pcrcg_radius_reorder_jobs documents).
 *   out        one pcrcg_batch (cfg->group == 0) or nb / cfg->group of them
 *   ws         arena of pcrcg_pyramid_ws_bytes(n0, nb, cfg) bytes.  cfg->shrink in (0,1] is the caller's bound on
 *              rows(level l+1) / rows(level l) (1.0 = always enough; 3DMatch-like clouds keep about a quarter): level l is
 *              given room for n0 * shrink^l rows, and since round 6 that bound -- not a row count read back from the GPU --
 *              sizes every buffer, table and launch of the level.  A cloud that keeps more rows than the bound allows is
 *              reported as PCRCG_EWORKSPACE at the end of the call -- nothing is corrupted, call again with a larger
 *              shrink (and the arena that goes with it).
 *   h_scratch  HOST scratch of >= 256 ints, pinned for best latency; h_lengths HOST [n_levels * nb] receives the
 *              per-level cloud lengths (stack_lengths); h_status HOST int (pinned; may be NULL) receives the tie-order
 *              restore status word ASYNCHRONOUSLY -- valid once `stream` has drained, 0 = fine, else as documented at
 *              pcrcg_radius_reorder_jobs.
 *   deferred   NULL: the tie-order restore step (KD-forest + reorder of the rows holding ties) is enqueued on `stream`
 *              by this call.  Otherwise the call only DESCRIBES it in *deferred (pointers into `ws`) and the caller
 *              enqueues it with pcrcg_pyramid_restore_run on a stream of its choice that is ordered after `stream`'s
 *              work and before the first reader of the tables (the pair engine runs it on the pair's model stream:
 *              the front-end stream is the pipeline's bottleneck).
 * The call WAITS for `stream` ONCE, at the end of the kernel chain, for the row counts of the levels and the tables'
 * column counts (one copy of ~100 words): the levels are sized from cfg->shrink, every kernel takes its row count from
 * the cloud lengths on the device.  (Rounds 2-5 read every subsampled level's row count back before sizing the next:
 * four round trips per call, ~1.1 ms of idle front-end stream per four-pair chain.)  It holds no global state, so several
 * host threads may build pyramids on several streams concurrently.  The last launches (tie-order restore) are still in
 * flight when it returns.
 * ---------------------------------------------------------------------------------------------- */
typedef struct pcrcg_pyramid_cfg {
    int n_levels;
    float r_conv[PCRCG_MAX_LEVELS], r_pool[PCRCG_MAX_LEVELS], dl[PCRCG_MAX_LEVELS];
    int has_conv[PCRCG_MAX_LEVELS], pooled[PCRCG_MAX_LEVELS], limit[PCRCG_MAX_LEVELS];
    int tie_order;
    int group;   /* 0: all nb clouds form ONE batch (the reference's contract).  g > 0: the clouds are nb / g independent
                    groups of g clouds (g = 2: several fragment pairs stacked into one call); `out` then is an array of
                    nb / g batches, each with its own tables (indices relative to the group's supports, the group's own
                    column counts) -- what nb / g separate calls would have produced, from ONE chain of kernels: the
                    chain is latency-bound, so two pairs cost little more than one */
    int up_nearest; /* 0: upsample tables with `limit` columns, as collate_fn_descriptor builds them
                       (ref:datasets/dataloader.py:287-297).  1: ONE column -- the nearest coarse point within the radius
                       (the reference's order among equidistant ones), which is all that the network reads of them
                       (closest_pool, ref:models/blocks.py:77-87): for hosts that feed pcrcg_kpfcnn_forward and nothing
                       else, the search keeps one candidate per row instead of sorting and writing `limit` of them */
    double shrink;  /* bound on rows(level l+1) / rows(level l), in (0,1]; 0 or out of range = 1.0 (see `ws` above) */
    void* side_stream;  /* NULL, or further streams (hipStream_t) of the caller's: the chain is a DAG -- a level's subsampling  */
    void* side_stream2; /* needs only the level's points, as do its cell grid and conv search; a KD-forest of the restore step
                       needs only its levels' points.  side_stream: the subsamplings run there, back to back from the moment
                       the input is in place, beside the grids and searches on `stream`, which waits for each subsampled
                       level when it first reads it.  side_stream2: the KD-forests (tie_order = 1) -- level 0's at once, the
                       subsampled levels' when the last of them exists; NULL: behind the subsamplings on side_stream.
                       `stream` waits for all of it before the call returns.  Both NULL: one stream, one line of kernels.
                       Which streams: gfx950 has four hardware dispatchers and a dispatcher hands out one kernel's workgroups at
                       a time (pcrcg_stream_pipe_classes).  Side streams on OTHER dispatchers than `stream`'s -- idle ones --
                       halve the chain (one 2 x 30 000-point pair 1.01 against 1.65 ms, four pairs 1.85 against 3.17:
                       profiles/r06_chain_latency_alone.txt); side streams on `stream`'s own dispatcher gain nothing; side
                       streams on a dispatcher that also serves a stream of large kernels stand behind every one of those
                       (the pair engine, whose other three dispatchers run the forwards, builds its chains in line). */
} pcrcg_pyramid_cfg;
typedef struct pcrcg_pyramid_restore {
    int njobs;                                   /* 0: no row holds a tie, nothing to do but post the status word */
    pcrcg_reorder_job jobs[PCRCG_MAX_REORDER_JOBS]; /* every job names its forest (sup / forest / forest_ns / forest_nb): the
                                                    builder keeps two, level 0's and one over the subsampled levels, and has
                                                    enqueued their builds -- they exist once `stream`'s work has passed */
    int* tie_status;                             /* device status word */
} pcrcg_pyramid_restore;
/* Threads: calls from several host threads are independent (arena, h_scratch and outputs are the caller's).  Calls that name
 * the SAME stream hold that stream's enqueue lock while they enqueue their chain and release it before they wait for their
 * round trip: the chains stay whole, one behind the other, and one call's wait overlaps the next call's enqueue (with
 * `deferred` set nothing of a call is left in the stream when it returns; without it the restore step follows whatever
 * another thread has enqueued meanwhile).  profiles/r06_ab_overlapped_builds.txt. */
size_t pcrcg_pyramid_ws_bytes(int n0, int nb, const pcrcg_pyramid_cfg* cfg);
int pcrcg_pyramid_build(const float* pts, int n0, const int* len, int nb, const pcrcg_pyramid_cfg* cfg, void* ws,
                        size_t ws_bytes, int* h_scratch, pcrcg_batch* out, int* h_lengths, int* h_status,
                        pcrcg_pyramid_restore* deferred, void* stream);
/* The same build with the input clouds given in `parts` pieces (e.g. the two pairs of a cfg.group = 2 build, each where
 * its producer left it): part i has n_parts[i] rows at pts_parts[i] and nb_parts[i] cloud lengths at len_parts[i] (device);
 * the builder copies them behind each other into its arena -- it copies its input anyway -- so the caller runs no
 * concatenation kernel.  Row for row the result of pcrcg_pyramid_build on the concatenated input. */
int pcrcg_pyramid_build_parts(const float* const* pts_parts, const int* n_parts, const int* const* len_parts,
                              const int* nb_parts, int parts, const pcrcg_pyramid_cfg* cfg, void* ws, size_t ws_bytes,
                              int* h_scratch, pcrcg_batch* out, int* h_lengths, int* h_status,
                              pcrcg_pyramid_restore* deferred, void* stream);
int pcrcg_pyramid_restore_run(const pcrcg_pyramid_restore* r, int* h_status, void* stream);

/* A non-blocking HIP stream created by the library (hipStreamCreateWithPriority(hipStreamNonBlocking); priority 0 =
 * default, -1 = high) for hosts that have no stream abstraction of their own; *stream receives a hipStream_t. */
int pcrcg_stream_create(void** stream, int priority);
int pcrcg_stream_destroy(void* stream);
/* Which of the caller's n (<= 64) streams share a hardware dispatcher.  gfx950's command processor has four compute
 * dispatchers; every stream's hardware queue belongs to one of them, and a dispatcher hands out the workgroups of one kernel
 * at a time -- two busy streams on the same dispatcher take turns, kernel by kernel, instead of running side by side
 * (profiles/r06_queue_pipes.txt).  cls[i] receives the class of streams[i] (0 = streams[0]'s class, then 1, 2, .. by first
 * appearance).  The classes are MEASURED (~1.5 ms per test on an idle GPU: a dispatch-bound probe kernel on one stream, a
 * one-workgroup kernel on the other); scratch = 16 bytes of device memory.  A host that runs several streams side by side
 * (pcrcg_amd/pairstream.py) picks them from different classes, and puts streams of small, latency-bound kernels that should
 * not wait for each other's resources -- the front-end chain and its KD-forests -- into one. */
int pcrcg_stream_pipe_classes(void* const* streams, int n, int* cls, void* scratch);

#ifdef __cplusplus
}
#endif
#endif /* PCRCG_H */
