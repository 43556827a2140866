// gnn.hip -- helpers of the GNN / cross-attention overlap head on gfx950 (ref:models/gcn.py).
//
//   pcrcg_knn             get_graph_feature's neighbour selection (:15-34, :48-51) without the N x N
//                         distance matrix: one wavefront per point re-evaluates the distances in each
//                         of the k+1 selection rounds (N is a few hundred at the coarsest level).
//   pcrcg_edgeconv_reduce the DGCNN edge convolution (:37-64, :123-129) without materialising the
//                         [1, C, N, N] tensor of :55 nor the [1, 2C, N, k] edge tensor: the 1x1 conv
//                         over cat(f_i, f_j - f_i) is linear, so the host splits it into a centre term
//                         and a neighbour term (two N x C GEMMs) and this kernel forms
//                         e[i,j,c] = ctr[i,c] + nbr[idx[i,j],c] on the fly, producing max_j e and the
//                         InstanceNorm2d statistics over all (i,j).
//   pcrcg_softmax_rows    attention / saliency softmax (:151-155; ref:models/architectures.py:562-563).
#include "common.h"
#include "pcrcg_train.h"

namespace pcrcg {

size_t colstats_ws_bytes(int c);
int colstats_finalize(const double* partial, int nchunks, int c, double count, float eps, float* stats,
                      hipStream_t st);
int colstats_chunks();

namespace {

typedef unsigned long long u64;

// The reference's kNN distance (ref:models/gcn.py:15-34) is badly conditioned -- -2ab + a^2 + b^2 in fp32 cancels to a few
// ulps of |a|^2, 1e-3 m^2 on KITTI-sized coordinates -- so WHICH points are the k nearest depends on every rounding.
// The expression is therefore evaluated exactly as the reference's CPU run rounds it (probed entry for entry on the
// 1936-point coarse clouds of the K120k pair): the [N,3]x[3,N] product as an FMA chain over x, y, z (MKL sgemm),
// |a|^2 as (x^2 + y^2) + z^2 from rounded squares (torch.sum of src**2), then (-2*dot + |a|^2) + |b|^2 and the clamp,
// each step rounded.  No contraction of the written operations.
__device__ __forceinline__ float knn_sq(float x, float y, float z) {
#pragma clang fp contract(off)
    return (x * x + y * y) + z * z;
}
__device__ __forceinline__ float knn_dist(float ax, float ay, float az, float sa, float bx, float by, float bz) {
#pragma clang fp contract(off)
    const float dot = __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));
    const float sb = (bx * bx + by * by) + bz * bz;
    const float d = (-2.0f * dot + sa) + sb;          // square_distance :26-31
    return fmaxf(d, 1e-12f);                          // clamp :33
}

// ---- torch.topk's order among EXACTLY equal distances -------------------------------------------------------------
// The reference takes `dist.topk(k+1, largest=False, sorted=True)` (ref:models/gcn.py:49).  With the ill-conditioned
// distance above, equal fp32 values are common on large coordinates (K120k: ~1 % of the rows hold one among their k+2
// smallest), and which of them survive the cut is decided by PyTorch's CPU kernel: (value, index) pairs through
// std::partial_sort when (k+1)*64 <= n, else std::nth_element + std::sort of the first k (libstdc++; comparator on the
// value alone).  Five such rows moved 4 % of the K120k pair's output rows by more than 1e-4.  A row whose k+2 smallest
// distances are all different has ONE answer and comes from the rank selection below; a row that holds a tie replays
// the library algorithms step by step (restated in oracle/topk_replay.py, pinned there against torch.topk itself).
// The replay is written as sequential code that all 64 lanes of the row's wavefront execute redundantly on a
// wavefront-private LDS array (identical writes to identical addresses); only the scan of __heap_select is spread
// over the lanes.  Keys: (distance bits << 32) | index; distances are > 0, so the bit patterns order like the values.
#define KNN_RB_MAX 704                     // regime B (nth_element) holds the whole row: n < 64 * (k + 1) <= 704
#define KNN_K_MAX 10
__device__ __forceinline__ bool kless(u64 x, u64 y) { return (x >> 32) < (y >> 32); }

__device__ void rp_adjust_heap(u64* a, int hole, int len, u64 value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (kless(a[child], a[child - 1])) --child;
        a[hole] = a[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[hole] = a[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;           // __push_heap
    while (hole > top && kless(a[parent], value)) {
        a[hole] = a[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[hole] = value;
}
__device__ void rp_make_heap(u64* a, int len) {
    if (len < 2) return;
    for (int parent = (len - 2) / 2;; --parent) {
        rp_adjust_heap(a, parent, len, a[parent]);
        if (parent == 0) return;
    }
}
__device__ void rp_sort_heap(u64* a, int len) {
    while (len > 1) {
        --len;
        const u64 value = a[len];
        a[len] = a[0];
        rp_adjust_heap(a, 0, len, value);
    }
}
__device__ void rp_insertion_sort(u64* a, int first, int last) {
    for (int i = first + 1; i < last; ++i) {
        const u64 v = a[i];
        if (kless(v, a[first])) {
            for (int j = i; j > first; --j) a[j] = a[j - 1];
            a[first] = v;
        } else {
            int j = i;
            while (kless(v, a[j - 1])) {
                a[j] = a[j - 1];
                --j;
            }
            a[j] = v;
        }
    }
}
__device__ int rp_partition_pivot(u64* a, int first, int last) {
    const int mid = first + (last - first) / 2;
    const int x = first + 1, y = mid, z = last - 1;
    int pick;                               // __move_median_to_first
    if (kless(a[x], a[y])) pick = kless(a[y], a[z]) ? y : (kless(a[x], a[z]) ? z : x);
    else pick = kless(a[x], a[z]) ? x : (kless(a[y], a[z]) ? z : y);
    u64 t = a[first];
    a[first] = a[pick];
    a[pick] = t;
    int lo = first + 1, hi = last;
    const u64 pivot = a[first];
    for (;;) {                              // __unguarded_partition
        while (kless(a[lo], pivot)) ++lo;
        --hi;
        while (kless(pivot, a[hi])) --hi;
        if (!(lo < hi)) return lo;
        t = a[lo];
        a[lo] = a[hi];
        a[hi] = t;
        ++lo;
    }
}
// one row: writes the k indices torch.topk(k+1 smallest, sorted)[1:] holds on the CPU.  `a`: KNN_RB_MAX keys of LDS
// private to this wavefront.
__device__ void knn_replay_row(const float* __restrict__ coords, int n, int i, int k, u64* a, int* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int K = k + 1;
    const float ax = coords[3 * (long)i], ay = coords[3 * (long)i + 1], az = coords[3 * (long)i + 2];
    const float sa = knn_sq(ax, ay, az);
    auto key_of = [&](int j) -> u64 {
        const float d = knn_dist(ax, ay, az, sa, coords[3 * (long)j], coords[3 * (long)j + 1], coords[3 * (long)j + 2]);
        return ((u64)__float_as_uint(d) << 32) | (unsigned)j;
    };
    if (K * 64 <= n) {
        // std::partial_sort = __heap_select + __sort_heap; the heap is a[0..K)
        if (lane < K) a[lane] = key_of(lane);
        rp_make_heap(a, K);
        for (int base = K; base < n; base += 64) {
            const int j = base + lane;
            const u64 kj = j < n ? key_of(j) : ~0ull;
            u64 m = __ballot(kless(kj, a[0]));
            while (m) {                     // candidates in index order; the heap's top changes with every pop
                const int l = __ffsll((long long)m) - 1;
                const u64 value = __shfl(kj, l, 64);
                rp_adjust_heap(a, 0, K, value);           // __pop_heap(first, middle, i): the old top leaves
                m = __ballot(kless(kj, a[0])) & ~((2ull << l) - 1ull);
            }
        }
        rp_sort_heap(a, K);
    } else {
        // std::nth_element(begin, begin + K - 1, end) + std::sort(begin, begin + K - 1)
        for (int j = lane; j < n; j += 64) a[j] = key_of(j);
        int first = 0, last = n;
        const int nth = K - 1;
        int depth = 2 * (31 - __clz(n));
        bool done = false;
        while (last - first > 3) {
            if (depth == 0) {               // __heap_select(first, nth + 1, last) + iter_swap(first, nth)
                u64* h = a + first;
                const int hl = nth + 1 - first;
                rp_make_heap(h, hl);
                for (int j = nth + 1; j < last; ++j)
                    if (kless(a[j], h[0])) {
                        const u64 value = a[j];
                        a[j] = h[0];
                        rp_adjust_heap(h, 0, hl, value);
                    }
                const u64 t = a[first];
                a[first] = a[nth];
                a[nth] = t;
                done = true;
                break;
            }
            --depth;
            const int cut = rp_partition_pivot(a, first, last);
            if (cut <= nth) first = cut;
            else last = cut;
        }
        if (!done) rp_insertion_sort(a, first, last);
        rp_insertion_sort(a, 0, K - 1);     // std::sort of K - 1 <= 16 elements is its final insertion sort alone
    }
    if (lane >= 1 && lane < K) out[lane - 1] = (int)(a[lane] & 0xFFFFFFFFull);
}

__global__ void __launch_bounds__(256) k_knn(const float* __restrict__ coords, int n, int k, int* __restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float ax = coords[3 * (long)i], ay = coords[3 * (long)i + 1], az = coords[3 * (long)i + 2];
    const float sa = knn_sq(ax, ay, az);
    __shared__ u64 s_q[4][KNN_RB_MAX];
    u64 last = 0;
    bool have_last = false;
    bool tie = false;
    for (int round = 0; round <= k + 1; ++round) {      // one round past the cut: is the cut inside a tie group?
        u64 best = ~0ull;
        for (int j = lane; j < n; j += 64) {
            const float bx = coords[3 * (long)j], by = coords[3 * (long)j + 1], bz = coords[3 * (long)j + 2];
            float d = knn_dist(ax, ay, az, sa, bx, by, bz);
            const u64 key = ((u64)__float_as_uint(d) << 32) | (unsigned)j;
            if ((!have_last || key > last) && key < best) best = key;
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const u64 o = __shfl_xor(best, s, 64);
            best = o < best ? o : best;
        }
        tie |= have_last && best != ~0ull && (best >> 32) == (last >> 32);
        last = best;
        have_last = true;
        // topk(k+1) sorted ascending, first dropped (:48-49)
        if (round >= 1 && round <= k && lane == 0)
            idx[(long)i * k + (round - 1)] = best == ~0ull ? i : (int)(best & 0xFFFFFFFFull);
    }
    if (tie && n > k && k <= KNN_K_MAX) knn_replay_row(coords, n, i, k, s_q[threadIdx.x >> 6], idx + (long)i * k);
}

// The same selection with the candidate keys of a query kept in registers (n <= 64 * PER): the distances are formed
// once instead of once per round (same expression, same keys, same result), and a round is a register scan plus the
// wavefront reduction.  The coarse level of an indoor pair has a few hundred points.
template <int PER>
__global__ void __launch_bounds__(256) k_knn_reg(const float* __restrict__ coords, int n, int k, int* __restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float ax = coords[3 * (long)i], ay = coords[3 * (long)i + 1], az = coords[3 * (long)i + 2];
    const float sa = knn_sq(ax, ay, az);
    u64 key[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int j = lane + 64 * t;
        const int jc = j < n ? j : n - 1;
        const float bx = coords[3 * (long)jc], by = coords[3 * (long)jc + 1], bz = coords[3 * (long)jc + 2];
        float d = knn_dist(ax, ay, az, sa, bx, by, bz);
        key[t] = j < n ? (((u64)__float_as_uint(d) << 32) | (unsigned)j) : ~0ull;
    }
    __shared__ u64 s_q[4][KNN_RB_MAX];
    u64 last = 0;
    bool tie = false;
    for (int round = 0; round <= k + 1; ++round) {      // one round past the cut: is the cut inside a tie group?
        u64 best = ~0ull;
#pragma unroll
        for (int t = 0; t < PER; ++t)
            if ((round == 0 || key[t] > last) && key[t] < best) best = key[t];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const u64 o = __shfl_xor(best, s, 64);
            best = o < best ? o : best;
        }
        tie |= round > 0 && best != ~0ull && (best >> 32) == (last >> 32);
        last = best;
        if (round >= 1 && round <= k && lane == 0)
            idx[(long)i * k + (round - 1)] = best == ~0ull ? i : (int)(best & 0xFFFFFFFFull);
    }
    if (tie && n > k && k <= KNN_K_MAX) knn_replay_row(coords, n, i, k, s_q[threadIdx.x >> 6], idx + (long)i * k);
}

// Round 5: multi-head attention on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: fp32 operands, fp32 accumulation -- the
// reference's arithmetic, no split).  k_attention below is a VALU kernel of 16 lanes per query that walks the keys in
// chunks of 64 behind three barriers each: 63 us for 381 x 382 x 2 heads of 128, the largest single launch of the coarse
// level.  Here one workgroup of four wavefronts takes 32 queries of one head of one cloud (blockIdx = (query tile, head,
// cloud); up to four clouds per launch):
//   1 scores   S = scale * Q K^T as 32 x 32 tiles, a key block per wavefront and turn.  The sum over the head dimension may
//              run in any order as long as both operands agree, so lane (l31, half) takes the CONTIGUOUS half
//              [half * D/2, half * D/2 + D/2) of its query row (kept in registers for the whole kernel) and of its key row
//              (D/8 16-byte loads per block, all in flight) and step kk multiplies element kk of the two halves;
//              S goes to LDS ([32][ms] with an odd row stride);
//   2 softmax  a wavefront per 8 rows: row maximum, exp, row sum (wave reductions), P = exp(S - max) back in place;
//   3 P V      the keys split over the wavefronts; per step one key per half, V rows read 128 bytes per half-wave (coalesced),
//              D/32 output tiles; the four wavefronts' partial tiles meet in LDS and leave as one store, scaled by 1 / sum.
// Needs D % 32 == 0 and the scores of a 32-query tile in LDS: ms <= 1024 (the larger clouds take the per-head GEMM path).
struct AttCloud { const float* q; const float* k; const float* v; float* out; int n, ms; };
struct AttMulti { AttCloud cl[4]; int ldq, ldk, ldv, ldo; float scale; int chunk_blocks; };   // chunk_blocks: 32-key blocks per LDS chunk

typedef float attf16 __attribute__((ext_vector_type(16)));

// Clouds whose scores do not fit LDS at once (K120k: 1936 keys) are walked in CHUNKS of chunk_blocks key blocks with the
// running row maximum / row sum of the online softmax: per chunk the scores tile, then m' = max(m, chunk max),
// alpha = exp(m - m'), P = exp(S - m'), l = l alpha + sum P, and every wavefront's partial output is scaled by alpha (row by
// row) before the chunk's P V is added.  One chunk (every indoor cloud) is the plain three-phase form.
template <int D>
__global__ void __launch_bounds__(256) k_attention_mfma(AttMulti a) {
    constexpr int DH = D / 2, NT = D / 32;
    extern __shared__ __attribute__((aligned(16))) float att2_lds[];
    const AttCloud cl = a.cl[blockIdx.z];
    const int n = cl.n, ms = cl.ms;
    const int q0 = blockIdx.x * 32;
    if (q0 >= n) return;
    const int head = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l31 = lane & 31, half = lane >> 5;
    const int sld = a.chunk_blocks * 32 + 1;                  // odd row stride: a column of S walks all banks
    float* const S = att2_lds;                                 // [32][sld]
    float* const m_run = S + 32 * sld;                         // [32] running row maximum
    float* const l_run = m_run + 32;                           // [32] running row sum
    float* const alpha = l_run + 32;                           // [32] this chunk's rescale of what was accumulated before
    float* const red = alpha + 32;                             // [3][32][D] partial output tiles of wavefronts 1..3
    const int nkb = (ms + 31) / 32;
    if (threadIdx.x < 32) { m_run[threadIdx.x] = -INFINITY; l_run[threadIdx.x] = 0.f; }
    float qr[DH];
    {
        const float* qrow = cl.q + (long)min(q0 + l31, n - 1) * a.ldq + head * D + half * DH;
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) {
            const float4 t = *reinterpret_cast<const float4*>(qrow + 4 * c);
            qr[4 * c] = t.x; qr[4 * c + 1] = t.y; qr[4 * c + 2] = t.z; qr[4 * c + 3] = t.w;
        }
    }
    attf16 o[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    for (int kb_lo = 0; kb_lo < nkb; kb_lo += a.chunk_blocks) {
        const int kb_hi = min(nkb, kb_lo + a.chunk_blocks), cb = kb_hi - kb_lo;      // this chunk: key blocks [kb_lo, kb_hi)
        __syncthreads();                                       // the previous chunk's P is no longer read (and m / l are set)
        // ---- 1: scores of the chunk
        for (int kb = kb_lo + wave; kb < kb_hi; kb += 4) {
            const int key = kb * 32 + l31;
            const float* krow = cl.k + (long)min(key, ms - 1) * a.ldk + head * D + half * DH;
            float kr[DH];
#pragma unroll
            for (int c = 0; c < DH / 4; ++c) {
                const float4 t = *reinterpret_cast<const float4*>(krow + 4 * c);
                kr[4 * c] = t.x; kr[4 * c + 1] = t.y; kr[4 * c + 2] = t.z; kr[4 * c + 3] = t.w;
            }
            attf16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < DH; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qr[kk], kr[kk], acc, 0, 0, 0);
            // C/D layout: column = l31 (the key), rows (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                S[row * sld + (kb - kb_lo) * 32 + l31] = key < ms ? acc[r] * a.scale : -INFINITY;
            }
        }
        __syncthreads();
        // ---- 2: online softmax, rows 8 wave .. 8 wave + 7
        for (int rr = 0; rr < 8; ++rr) {
            const int row = wave * 8 + rr;
            float* const srow = S + row * sld;
            float mx = -INFINITY;
            for (int j = lane; j < cb * 32; j += 64) mx = fmaxf(mx, srow[j]);
#pragma unroll
            for (int s2 = 32; s2 >= 1; s2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s2, 64));
            const float m_old = m_run[row], m_new = fmaxf(m_old, mx);       // finite: every chunk holds at least one key
            float sum = 0.f;
            for (int j = lane; j < cb * 32; j += 64) {
                const float e = expf(srow[j] - m_new);                        // exp(-inf) = 0 for the padded keys
                srow[j] = e;
                sum += e;
            }
#pragma unroll
            for (int s2 = 32; s2 >= 1; s2 >>= 1) sum += __shfl_xor(sum, s2, 64);
            if (lane == 0) {
                const float al = expf(m_old - m_new);                         // 0 on the first chunk (m_old = -inf)
                alpha[row] = al;
                l_run[row] = l_run[row] * al + sum;
                m_run[row] = m_new;
            }
        }
        __syncthreads();
        // ---- 3: o = o alpha + P V over this chunk's keys; wavefront w takes the blocks [w * per, (w + 1) * per) of the chunk
        if (kb_lo > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float al = alpha[(r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
                for (int t = 0; t < NT; ++t) o[t][r] *= al;
            }
        }
        {
            const int per = (cb + 3) / 4;
            const int b0 = min(wave * per, cb), b1 = min(b0 + per, cb);
            // within a block of 32 keys, half 0 takes keys 0..15 and half 1 keys 16..31: step j multiplies key (16 half + j)
            for (int bi = b0; bi < b1; ++bi) {
                float pv[16], vv[NT][16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int kloc = bi * 32 + 16 * half + j, key = kb_lo * 32 + kloc;
                    pv[j] = S[l31 * sld + kloc];                 // A[row = l31][k]: P of this lane's query row
                    const float* vrow = cl.v + (long)min(key, ms - 1) * a.ldv + head * D;
#pragma unroll
                    for (int t = 0; t < NT; ++t) vv[t][j] = key < ms ? vrow[t * 32 + l31] : 0.f;     // B[k][col = l31]
                }
#pragma unroll
                for (int j = 0; j < 16; ++j)
#pragma unroll
                    for (int t = 0; t < NT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[j], vv[t][j], o[t], 0, 0, 0);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                red[((wave - 1) * 32 + row) * D + t * 32 + l31] = o[t][r];
            }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int at = row * D + t * 32 + l31;
                const float v = ((o[t][r] + red[at]) + (red[32 * D + at] + red[64 * D + at])) / l_run[row];
                if (q0 + row < n) cl.out[(long)(q0 + row) * a.ldo + head * D + t * 32 + l31] = v;
            }
    }
}

// ---- the attention's backward on the same tiles (train step) --------------------------------------------------------------
// out_h = softmax(scale q_h k_h^T) v_h (ref:models/gcn.py:151-155).  With P the softmax and D_i = sum_c dO_ic O_ic (= sum_j
// P_ij dP_ij, so P need not be kept from the forward):
//     dP = dO v^T,   dS = scale P o (dP - D),   dq = dS k,   dk = dS^T q,   dv = P^T dO.
// A workgroup takes 32 queries of one head: it re-forms S and P for them in LDS (as the forward does), then dS in its place,
// and its eight wavefronts split the key blocks for dP, for dk / dv (contractions over the tile's 32 queries, partial over
// the query tiles: float atomics into the zeroed gradients) and for dq (partial over a wavefront's keys: atomics as well).
// fp32 operands on v_mfma_f32_32x32x2_f32 throughout.  The score tile must fit LDS (dS overwrites P once dv has read it):
// ms <= 1216 keys (the indoor coarse clouds have 350-800); the tape keeps the per-head product path for anything larger
// and for deterministic=1.
struct AttBwd {
    const float* q; const float* k; const float* v; const float* o; const float* d_o;
    float* dq; float* dk; float* dv;
    int n, ms, ldq, ldk, ldv, ldo, ld_do, ld_dq, ld_dk, ld_dv;
    float scale;
};
template <int D>
__global__ void __launch_bounds__(512) k_attention_bwd_mfma(AttBwd a) {
    constexpr int DH = D / 2, NT = D / 32, NW = 8;          // eight wavefronts: the workgroups are few (n / 32 x heads), their key loops long
    extern __shared__ __attribute__((aligned(16))) float attb_lds[];
    const int n = a.n, ms = a.ms;
    const int q0 = blockIdx.x * 32;
    if (q0 >= n) return;
    const int head = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l31 = lane & 31, half = lane >> 5;
    const int nkb = (ms + 31) / 32;
    const int sld = nkb * 32 + 1;
    float* const P = attb_lds;                     // [32][sld]: scores, then P, then (in place) dS
    float* const dS = P;
    float* const Di = P + 32 * sld;                // [32]
    const int qrow = min(q0 + l31, n - 1);
    // ---- 1: scores
    {
        float qr[DH];
        const float* qp = a.q + (long)qrow * a.ldq + head * D + half * DH;
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) {
            const float4 t = *reinterpret_cast<const float4*>(qp + 4 * c);
            qr[4 * c] = t.x; qr[4 * c + 1] = t.y; qr[4 * c + 2] = t.z; qr[4 * c + 3] = t.w;
        }
        for (int kb = wave; kb < nkb; kb += NW) {
            const int key = kb * 32 + l31;
            const float* krow = a.k + (long)min(key, ms - 1) * a.ldk + head * D + half * DH;
            float kr[DH];
#pragma unroll
            for (int c = 0; c < DH / 4; ++c) {
                const float4 t = *reinterpret_cast<const float4*>(krow + 4 * c);
                kr[4 * c] = t.x; kr[4 * c + 1] = t.y; kr[4 * c + 2] = t.z; kr[4 * c + 3] = t.w;
            }
            attf16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < DH; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qr[kk], kr[kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                P[row * sld + key] = key < ms ? acc[r] * a.scale : -INFINITY;
            }
        }
    }
    __syncthreads();
    // ---- 2: P = softmax rows (rows past the cloud: zeros, so that they add nothing to dk / dv), D_i
    for (int rr = 0; rr < 32 / NW; ++rr) {
        const int row = wave * (32 / NW) + rr;
        float* const prow = P + row * sld;
        const bool real = q0 + row < n;
        float mx = -INFINITY;
        for (int j = lane; j < nkb * 32; j += 64) mx = fmaxf(mx, prow[j]);
#pragma unroll
        for (int s2 = 32; s2 >= 1; s2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s2, 64));
        float sum = 0.f;
        for (int j = lane; j < nkb * 32; j += 64) {
            const float e = expf(prow[j] - mx);
            prow[j] = e;
            sum += e;
        }
#pragma unroll
        for (int s2 = 32; s2 >= 1; s2 >>= 1) sum += __shfl_xor(sum, s2, 64);
        const float inv = real ? 1.0f / sum : 0.f;
        for (int j = lane; j < nkb * 32; j += 64) prow[j] *= inv;
        float dd = 0.f;
        const long ro = (long)min(q0 + row, n - 1);
        for (int c = lane; c < D; c += 64) dd = fmaf(a.d_o[ro * a.ld_do + head * D + c], a.o[ro * a.ldo + head * D + c], dd);
#pragma unroll
        for (int s2 = 32; s2 >= 1; s2 >>= 1) dd += __shfl_xor(dd, s2, 64);
        if (lane == 0) Di[row] = dd;
    }
    __syncthreads();
    // ---- 3: dv += P^T dO over this tile's 32 queries; wavefront w takes key blocks w, w + 8, ...
    for (int kb = wave; kb < nkb; kb += NW) {
        const int key_lo = kb * 32;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            attf16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int qi = 2 * s + half;                                   // the contraction index: a query of the tile
                const long ro = (long)min(q0 + qi, n - 1);
                const float av = P[qi * sld + key_lo + l31];                                      // A[row = key][k = query]
                const float bv = a.d_o[ro * a.ld_do + head * D + 32 * t + l31];                        // B[k = query][col = channel]
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key_lo + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (key < ms) atomicAdd(a.dv + (long)key * a.ld_dv + head * D + 32 * t + l31, acc[r]);
            }
        }
    }
    __syncthreads();
    // ---- 4: dP = dO v^T per key block, dS = scale P (dP - D) IN PLACE of P (an element is read and written by one lane)
    {
        float gr[DH];
        const float* gp = a.d_o + (long)qrow * a.ld_do + head * D + half * DH;
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) {
            const float4 t = *reinterpret_cast<const float4*>(gp + 4 * c);
            gr[4 * c] = t.x; gr[4 * c + 1] = t.y; gr[4 * c + 2] = t.z; gr[4 * c + 3] = t.w;
        }
        for (int kb = wave; kb < nkb; kb += NW) {
            const int key = kb * 32 + l31;
            const float* vrow = a.v + (long)min(key, ms - 1) * a.ldv + head * D + half * DH;
            float vr[DH];
#pragma unroll
            for (int c = 0; c < DH / 4; ++c) {
                const float4 t = *reinterpret_cast<const float4*>(vrow + 4 * c);
                vr[4 * c] = t.x; vr[4 * c + 1] = t.y; vr[4 * c + 2] = t.z; vr[4 * c + 3] = t.w;
            }
            attf16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < DH; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gr[kk], vr[kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                dS[row * sld + key] = a.scale * P[row * sld + key] * (acc[r] - Di[row]);      // P = 0 on padded keys and rows
            }
        }
    }
    __syncthreads();
    // ---- 5: dk += dS^T q, the same contraction with dS and the queries
    for (int kb = wave; kb < nkb; kb += NW) {
        const int key_lo = kb * 32;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            attf16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int qi = 2 * s + half;                                   // the contraction index: a query of the tile
                const long ro = (long)min(q0 + qi, n - 1);
                const float av = dS[qi * sld + key_lo + l31];                                      // A[row = key][k = query]
                const float bv = a.q[ro * a.ldq + head * D + 32 * t + l31];                        // B[k = query][col = channel]
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key_lo + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (key < ms) atomicAdd(a.dk + (long)key * a.ld_dk + head * D + 32 * t + l31, acc[r]);
            }
        }
    }
    // ---- 6: dq += dS k over a wavefront's quarter of the key blocks
    {
        attf16 o[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
        const int per = (nkb + NW - 1) / NW;
        const int b0 = min(wave * per, nkb), b1 = min(b0 + per, nkb);
        for (int bi = b0; bi < b1; ++bi) {
            float pv[16], kv[NT][16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int key = bi * 32 + 16 * half + j;
                pv[j] = dS[l31 * sld + key];
                const float* krow = a.k + (long)min(key, ms - 1) * a.ldk + head * D;
#pragma unroll
                for (int t = 0; t < NT; ++t) kv[t][j] = key < ms ? krow[t * 32 + l31] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int t = 0; t < NT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[j], kv[t][j], o[t], 0, 0, 0);
        }
        if (b1 > b0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (q0 + row < n) atomicAdd(a.dq + (long)(q0 + row) * a.ld_dq + head * D + t * 32 + l31, o[t][r]);
                }
        }
    }
}

// Round 5: the same reduction with the ROWS in parallel.  k_edgeconv_reduce below walks its rows with one 4-byte load per
// lane and neighbour (a dependent chain of rows_per_chunk / 4 x (index -> k loads): 20 us for 381 x 10 x 512 values that
// fit L2 ten times over).  Here a wavefront takes one row at a time: lanes 0 .. k-1 load the row's neighbour indices, which
// are then wave-uniform scalars, and every lane carries FOUR channels, so a neighbour row is one 16-byte load per lane
// with all k of them in flight; 64 lanes x 4 = a 256-channel block (blockIdx.y), 4 wavefronts x RPW rows per workgroup
// (blockIdx.x), up to four clouds per launch (blockIdx.z: the source and target clouds of the pairs of one forward call
// share the layer).  Column sums: fp64 per lane, the four wavefronts meet in LDS, then one fp64 atomic per channel and
// workgroup (ATOMIC) or a stored partial (the deterministic form: [2][c][nchunks], finished by k_colstats_final).
struct EdgeMulti { EdgeCloud cl[4]; int ld_ctr, ld_nbr, ld_emax, c, rows_per_block, nchunks; };

template <bool ATOMIC>
__global__ void __launch_bounds__(256) k_edgeconv_rows(EdgeMulti a) {
    __shared__ double s_sum[4][256], s_sq[4][256];
    const EdgeCloud cl = a.cl[blockIdx.z];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c0 = blockIdx.y * 256 + lane * 4;
    const bool cok = c0 < a.c;
    const int r0 = blockIdx.x * a.rows_per_block, r1 = min(cl.n, r0 + a.rows_per_block);
    double s[4] = {0.0, 0.0, 0.0, 0.0}, sq[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r = r0 + wave; r < r1; r += 4) {
        const int mine = lane < cl.k ? cl.idx[(long)r * cl.k + lane] : 0;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f), m = q;
        if (cok) q = *reinterpret_cast<const float4*>(cl.ctr + (long)r * a.ld_ctr + c0);
        for (int j0 = 0; j0 < cl.k; j0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {                      // all eight rows in flight (indices past k re-read a valid row)
                const int nb = __builtin_amdgcn_readlane(mine, min(j0 + u, cl.k - 1));
                v[u] = cok ? *reinterpret_cast<const float4*>(cl.nbr + (long)nb * a.ld_nbr + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (j0 + u >= cl.k) continue;
                const float e[4] = {q.x + v[u].x, q.y + v[u].y, q.z + v[u].z, q.w + v[u].w};
                if (j0 + u == 0) m = make_float4(e[0], e[1], e[2], e[3]);
                else { m.x = fmaxf(m.x, e[0]); m.y = fmaxf(m.y, e[1]); m.z = fmaxf(m.z, e[2]); m.w = fmaxf(m.w, e[3]); }
#pragma unroll
                for (int t = 0; t < 4; ++t) { s[t] += (double)e[t]; sq[t] += (double)e[t] * (double)e[t]; }
            }
        }
        if (cok) *reinterpret_cast<float4*>(cl.emax + (long)r * a.ld_emax + c0) = m;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) { s_sum[wave][lane * 4 + t] = s[t]; s_sq[wave][lane * 4 + t] = sq[t]; }
    __syncthreads();
    const int ch = blockIdx.y * 256 + threadIdx.x;
    if (ch < a.c && r0 < r1) {
        const int t = threadIdx.x;
        const double ss = (s_sum[0][t] + s_sum[1][t]) + (s_sum[2][t] + s_sum[3][t]);
        const double qq = (s_sq[0][t] + s_sq[1][t]) + (s_sq[2][t] + s_sq[3][t]);
        if (ATOMIC) {
            unsafeAtomicAdd(&cl.sums[ch], ss);
            unsafeAtomicAdd(&cl.sums[(long)a.c + ch], qq);
        } else {
            cl.sums[(long)ch * a.nchunks + blockIdx.x] = ss;                     // layout [2][c][nchunks]
            cl.sums[((long)a.c + ch) * a.nchunks + blockIdx.x] = qq;
        }
    } else if (!ATOMIC && ch < a.c) {
        cl.sums[(long)ch * a.nchunks + blockIdx.x] = 0.0;
        cl.sums[((long)a.c + ch) * a.nchunks + blockIdx.x] = 0.0;
    }
}

template <bool ATOMIC>      // ATOMIC: the chunk's sums are added into zeroed [2][c] accumulators instead of stored as partials
__global__ void __launch_bounds__(256) k_edgeconv_reduce(const float* __restrict__ ctr, int ld_ctr,
                                                          const float* __restrict__ nbr, int ld_nbr,
                                                          const int* __restrict__ idx, int n, int k, int c,
                                                          float* __restrict__ emax, int ld_emax,
                                                          double* __restrict__ partial) {
    __shared__ double s_sum[4][64], s_sq[4][64];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int ch = blockIdx.y * 64 + lane;
    const int chunk = blockIdx.x, nchunks = gridDim.x;
    const int rows_per = (n + nchunks - 1) / nchunks;
    const int r0 = chunk * rows_per, r1 = min(n, r0 + rows_per);
    double s = 0.0, sq = 0.0;
    if (ch < c)
        for (int r = r0 + rl; r < r1; r += 4) {
            const float q = ctr[(long)r * ld_ctr + ch];
            float m = 0.f;
            for (int j = 0; j < k; ++j) {
                const float v = q + nbr[(long)idx[(long)r * k + j] * ld_nbr + ch];
                m = j == 0 ? v : fmaxf(m, v);
                s += (double)v;
                sq += (double)v * (double)v;
            }
            emax[(long)r * ld_emax + ch] = m;
        }
    s_sum[rl][lane] = s;
    s_sq[rl][lane] = sq;
    __syncthreads();
    if (rl == 0 && ch < c) {
        s = (s_sum[0][lane] + s_sum[1][lane]) + (s_sum[2][lane] + s_sum[3][lane]);
        sq = (s_sq[0][lane] + s_sq[1][lane]) + (s_sq[2][lane] + s_sq[3][lane]);
        if (ATOMIC) {
            if (r0 < r1) {
                unsafeAtomicAdd(&partial[ch], s);
                unsafeAtomicAdd(&partial[(long)c + ch], sq);
            }
        } else {
            partial[(long)ch * nchunks + chunk] = s;                   // layout [2][c][nchunks]
            partial[((long)c + ch) * nchunks + chunk] = sq;
        }
    }
}

// one wavefront per row
__global__ void __launch_bounds__(256) k_softmax_rows(float* __restrict__ x, int rows, int cols, int ld, float scale) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float* row = x + (long)r * ld;
    float m = -INFINITY;
    for (int j = lane; j < cols; j += 64) m = fmaxf(m, row[j] * scale);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    float sum = 0.f;
    for (int j = lane; j < cols; j += 64) {
        const float e = expf(row[j] * scale - m);
        row[j] = e;
        sum += e;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) sum += __shfl_xor(sum, s, 64);
    const float inv = 1.0f / sum;
    for (int j = lane; j < cols; j += 64) row[j] *= inv;
}

// Multi-head attention in one launch: out[:, h*D:(h+1)*D] = softmax(scale * q_h k_h^T) v_h  (ref:models/gcn.py:151-155).
// On the path the operands are tiny (a few hundred coarse points, D = 64): three GEMM-class launches + a softmax per
// head and direction were 24 launches of ~20 us of latency each.  Here a workgroup owns 8 or 16 queries of one head;
// G = 16 or 32 lanes share a query: each lane scores 64/G of the 64 keys of a chunk (the query row lives in registers, the key
// chunk in LDS with rows padded to D + 4 floats), the chunk's probabilities go through LDS, and each lane
// accumulates D/G output channels with the running-max / running-sum rescaling (softmax in one pass over the keys,
// fp32 throughout, plain FMAs: 0.3 GFLOP per call needs no matrix core).
template <int D, int TQ>      // TQ queries per workgroup, G = 256 / TQ lanes per query
__global__ void __launch_bounds__(256) k_attention(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                   const float* __restrict__ v, int ldv, float* __restrict__ out, int ldo,
                                                   int n, int ms, float scale) {
    constexpr int G = 256 / TQ, KC = 64, KT = KC / G, KS = D + 4, DV = D / G, D4 = D / 4;
    static_assert(KC % G == 0 && D % G == 0 && G <= 64, "lane groups must tile the chunk and the head");
    extern __shared__ float4 att_lds[];                   // Ks [KC][D + 4] | Vs [KC][D] | Ps [TQ][KC]
    float* const Ks = reinterpret_cast<float*>(att_lds);
    float* const Vs = Ks + KC * KS;
    float* const Ps = Vs + KC * D;
    const int tid = threadIdx.x, qi = tid / G, j = tid % G, head = blockIdx.y;
    const int row = blockIdx.x * TQ + qi;
    const float* qrow = q + (long)(row < n ? row : n - 1) * ldq + head * D;
    float4 qr[D4];
#pragma unroll
    for (int c = 0; c < D4; ++c) qr[c] = *reinterpret_cast<const float4*>(qrow + 4 * c);
    float m = -INFINITY, l = 0.f, o[DV];
#pragma unroll
    for (int i = 0; i < DV; ++i) o[i] = 0.f;
    for (int kc0 = 0; kc0 < ms; kc0 += KC) {
        __syncthreads();                                  // the previous chunk's readers are done
#pragma unroll
        for (int e = tid; e < KC * D4; e += 256) {
            const int key = e / D4, c4 = e % D4;
            const bool ok = kc0 + key < ms;
            const long r = ok ? kc0 + key : 0;
            const float4 kv = *reinterpret_cast<const float4*>(k + r * ldk + head * D + 4 * c4);
            const float4 vv = *reinterpret_cast<const float4*>(v + r * ldv + head * D + 4 * c4);
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(Ks + key * KS + 4 * c4) = ok ? kv : z;
            *reinterpret_cast<float4*>(Vs + key * D + 4 * c4) = ok ? vv : z;
        }
        __syncthreads();
        float sc[KT], mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const int key = j + G * t;
            const float* kr = Ks + key * KS;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < D4; ++c) {
                const float4 kv = *reinterpret_cast<const float4*>(kr + 4 * c);
                acc = fmaf(qr[c].x, kv.x, acc);
                acc = fmaf(qr[c].y, kv.y, acc);
                acc = fmaf(qr[c].z, kv.z, acc);
                acc = fmaf(qr[c].w, kv.w, acc);
            }
            sc[t] = kc0 + key < ms ? acc * scale : -INFINITY;
            mx = fmaxf(mx, sc[t]);
        }
#pragma unroll
        for (int s = G / 2; s >= 1; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s, 64));
        const float m_new = fmaxf(m, mx);                 // finite: every chunk holds at least one key
        const float alpha = expf(m - m_new);              // exp(-inf) = 0 on the first chunk
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const float pe = expf(sc[t] - m_new);
            Ps[qi * KC + j + G * t] = pe;
            psum += pe;
        }
#pragma unroll
        for (int s = G / 2; s >= 1; s >>= 1) psum += __shfl_xor(psum, s, 64);
        l = l * alpha + psum;
        m = m_new;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < DV; ++i) o[i] *= alpha;
#pragma unroll 4
        for (int k4 = 0; k4 < KC / 4; ++k4) {
            const float4 pp = *reinterpret_cast<const float4*>(Ps + qi * KC + 4 * k4);
            const float pv[4] = {pp.x, pp.y, pp.z, pp.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* vr = Vs + (4 * k4 + u) * D + j * DV;
#pragma unroll
                for (int i = 0; i < DV; ++i) o[i] = fmaf(pv[u], vr[i], o[i]);
            }
        }
    }
    if (row < n) {
        const float inv = 1.0f / l;
        float* orow = out + (long)row * ldo + head * D + j * DV;
#pragma unroll
        for (int i = 0; i < DV; ++i) orow[i] = o[i] * inv;
    }
}

// y[r] = sum_j softmax(scale * x[r, :])_j * vec[j * ldv]: the saliency scores (ref:models/architectures.py:562-563)
// without writing the probabilities back.  One wavefront per row.
__global__ void __launch_bounds__(256) k_softmax_matvec(const float* __restrict__ x, int rows, int cols, int ld, float scale,
                                                        const float* __restrict__ vec, int ldv, float* __restrict__ y,
                                                        int ldy) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* row = x + (long)r * ld;
    float m = -INFINITY;
    for (int j = lane; j < cols; j += 64) m = fmaxf(m, row[j] * scale);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    float sum = 0.f, dot = 0.f;
    for (int j = lane; j < cols; j += 64) {
        const float e = expf(row[j] * scale - m);
        sum += e;
        dot = fmaf(e, vec[(long)j * ldv], dot);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        sum += __shfl_xor(sum, s, 64);
        dot += __shfl_xor(dot, s, 64);
    }
    if (lane == 0) y[(long)r * ldy] = dot / sum;
}

}  // namespace

// the layout rules of k_edgeconv_rows: four channels per lane as one 16-byte access, the row's indices on one wavefront
bool edgeconv_rows_ok(const EdgeCloud* cl, int count, int ld_ctr, int ld_nbr, int ld_emax, int c) {
    if (!debug_opts().edge_rows || count < 1 || count > 4 || c % 4 != 0 || ld_ctr % 4 != 0 || ld_nbr % 4 != 0 || ld_emax % 4 != 0) return false;
    for (int i = 0; i < count; ++i) {
        if (cl[i].k < 1 || cl[i].k > 64 || cl[i].n < 1) return false;
        if ((reinterpret_cast<uintptr_t>(cl[i].ctr) | reinterpret_cast<uintptr_t>(cl[i].nbr) | reinterpret_cast<uintptr_t>(cl[i].emax)) & 15)
            return false;
    }
    return true;
}
// emax + InstanceNorm2d statistics of up to four clouds in ONE launch.  atomic: cl[i].sums = zeroed [2][c] fp64 accumulators;
// else cl[i].sums = a partial buffer of colstats_ws_bytes(c) each, *nchunks_out = the chunk count colstats_finalize takes.
int edgeconv_rows_multi(const EdgeCloud* cl, int count, int ld_ctr, int ld_nbr, int ld_emax, int c, bool atomic, int* nchunks_out,
                        hipStream_t st) {
    EdgeMulti a;
    int nmax = 0;
    for (int i = 0; i < count; ++i) { a.cl[i] = cl[i]; nmax = cl[i].n > nmax ? cl[i].n : nmax; }
    for (int i = count; i < 4; ++i) a.cl[i] = cl[0];
    a.ld_ctr = ld_ctr; a.ld_nbr = ld_nbr; a.ld_emax = ld_emax; a.c = c;
    int rpb = 8;                                             // two rows per wavefront: ~48 row blocks for a 381-point cloud
    while ((nmax + rpb - 1) / rpb > colstats_chunks()) rpb += 4;
    a.rows_per_block = rpb;
    a.nchunks = (nmax + rpb - 1) / rpb;
    if (nchunks_out) *nchunks_out = a.nchunks;
    const dim3 grid(a.nchunks, (c + 255) / 256, count);
    if (atomic) hipLaunchKernelGGL(k_edgeconv_rows<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_edgeconv_rows<false>, grid, dim3(256), 0, st, a);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
// 32-key blocks of scores a workgroup keeps in LDS beside its partial output tiles (160 KB per CU, one workgroup of this
// kernel per CU is enough: it is a latency chain, not a throughput kernel)
static int attention_chunk_blocks(int d) {
    const size_t fixed = sizeof(float) * (96 + (size_t)96 * d);
    const size_t budget = 156 * 1024 - fixed;
    return (int)((budget / (32 * sizeof(float)) - 1) / 32);
}
bool attention_mfma_ok(const AttnCloud* cl, int count, int ldq, int ldk, int ldv, int d) {
    if (!debug_opts().att_mfma || count < 1 || count > 4 || (d != 32 && d != 64 && d != 128)) return false;
    if (ldq % 4 != 0 || ldk % 4 != 0 || ldv % 4 != 0) return false;
    for (int i = 0; i < count; ++i) {
        if (cl[i].n < 1 || cl[i].ms < 1) return false;
        if ((reinterpret_cast<uintptr_t>(cl[i].q) | reinterpret_cast<uintptr_t>(cl[i].k)) & 15) return false;
    }
    return true;
}
// out[:, h d:(h + 1) d] = softmax(scale q_h k_h^T) v_h for every head of up to four clouds in ONE launch (k_attention_mfma)
int attention_mfma_multi(const AttnCloud* cl, int count, int ldq, int ldk, int ldv, int ldo, int heads, int d, float scale,
                         hipStream_t st) {
    AttMulti a;
    int nmax = 0, msmax = 0;
    for (int i = 0; i < 4; ++i) {
        const AttnCloud& c = cl[i < count ? i : 0];
        a.cl[i] = AttCloud{c.q, c.k, c.v, c.out, c.n, c.ms};
        if (i < count) { nmax = c.n > nmax ? c.n : nmax; msmax = c.ms > msmax ? c.ms : msmax; }
    }
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.scale = scale;
    const int nkb = (msmax + 31) / 32, cap = attention_chunk_blocks(d);
    a.chunk_blocks = nkb < cap ? nkb : cap;
    const size_t sld = (size_t)a.chunk_blocks * 32 + 1;
    const size_t lds = sizeof(float) * (32 * sld + 96 + (size_t)3 * 32 * d);
    const dim3 grid((nmax + 31) / 32, heads, count);
#define ATT2(DD)                                                                                                          \
    do {                                                                                                                  \
        PCRCG_GRANT_LDS((k_attention_mfma<DD>));                                                            \
        hipLaunchKernelGGL((k_attention_mfma<DD>), grid, dim3(256), lds, st, a);                                          \
    } while (0)
    if (d == 128) ATT2(128);
    else if (d == 64) ATT2(64);
    else ATT2(32);
#undef ATT2
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
// attention backward on the matrix cores (k_attention_bwd_mfma): gradients ADDED to dq / dk / dv
bool attention_bwd_mfma_ok(int n, int ms, int d, int ldq, int ldk, int ldv, int ld_do) {
    if (!debug_opts().att_mfma || debug_opts().deterministic || n < 1 || ms < 1 || (d != 32 && d != 64 && d != 128)) return false;
    if (ldq % 4 != 0 || ldk % 4 != 0 || ldv % 4 != 0 || ld_do % 4 != 0) return false;
    const size_t sld = (size_t)((ms + 31) / 32) * 32 + 1;
    return sizeof(float) * (32 * sld + 32) <= 156 * 1024;
}
int attention_bwd_mfma(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, int ldo,
                       const float* d_o, int ld_do, float* dq, int ld_dq, float* dk, int ld_dk, float* dv, int ld_dv, int n, int ms,
                       int heads, int d, float scale, hipStream_t st) {
    PCRCG_CHECK_ARG(attention_bwd_mfma_ok(n, ms, d, ldq, ldk, ldv, ld_do) && q && k && v && o && d_o && dq && dk && dv);
    PCRCG_CHECK_ARG(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
                      reinterpret_cast<uintptr_t>(d_o)) & 15) == 0);
    AttBwd a{q, k, v, o, d_o, dq, dk, dv, n, ms, ldq, ldk, ldv, ldo, ld_do, ld_dq, ld_dk, ld_dv, scale};
    const size_t sld = (size_t)((ms + 31) / 32) * 32 + 1;
    const size_t lds = sizeof(float) * (32 * sld + 32);
    const dim3 grid((n + 31) / 32, heads);
#define ATTB(DD)                                                                                                          \
    do {                                                                                                                  \
        PCRCG_GRANT_LDS((k_attention_bwd_mfma<DD>));                                                            \
        hipLaunchKernelGGL((k_attention_bwd_mfma<DD>), grid, dim3(512), lds, st, a);                                      \
    } while (0)
    if (d == 128) ATTB(128);
    else if (d == 64) ATTB(64);
    else ATTB(32);
#undef ATTB
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
}  // namespace pcrcg

using namespace pcrcg;

extern "C" {

int pcrcg_knn(const float* coords, int n, int k, int* idx, void* stream) {
    PCRCG_CHECK_ARG(n >= 0 && k >= 1);
    if (n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(coords && idx);
    if (n <= 64 * 8)
        hipLaunchKernelGGL(k_knn_reg<8>, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), coords, n, k, idx);
    else if (n <= 64 * 16)
        hipLaunchKernelGGL(k_knn_reg<16>, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), coords, n, k, idx);
    else
        hipLaunchKernelGGL(k_knn, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), coords, n, k, idx);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

size_t pcrcg_edgeconv_ws_bytes(int c) { return colstats_ws_bytes(c); }

int pcrcg_edgeconv_reduce(const float* ctr, int ld_ctr, const float* nbr, int ld_nbr, const int* idx,
                          int n, int k, int c, float eps, float* emax, int ld_emax, float* stats,
                          void* ws, size_t ws_bytes, void* stream) {
    PCRCG_CHECK_ARG(n >= 1 && k >= 1 && c >= 1 && ld_ctr >= c && ld_nbr >= c && ld_emax >= c);
    PCRCG_CHECK_ARG(ctr && nbr && idx && emax && stats && ws);
    const int chunks = colstats_chunks();
    Carver cv(ws, ws_bytes);
    double* partial = cv.take<double>((size_t)chunks * 2 * c);
    PCRCG_CHECK_WS(cv);
    hipStream_t st = as_stream(stream);
    {
        const EdgeCloud one = {ctr, nbr, idx, emax, partial, n, k};
        if (edgeconv_rows_ok(&one, 1, ld_ctr, ld_nbr, ld_emax, c)) {
            int nchunks = 0;
            PCRCG_PROPAGATE(edgeconv_rows_multi(&one, 1, ld_ctr, ld_nbr, ld_emax, c, false, &nchunks, st));
            return colstats_finalize(partial, nchunks, c, (double)n * (double)k, eps, stats, st);
        }
    }
    hipLaunchKernelGGL(k_edgeconv_reduce<false>, dim3(chunks, (c + 63) / 64), dim3(256), 0, st, ctr, ld_ctr, nbr, ld_nbr,
                       idx, n, k, c, emax, ld_emax, partial);
    return colstats_finalize(partial, chunks, c, (double)n * (double)k, eps, stats, st);
}

int pcrcg_edgeconv_reduce_sums(const float* ctr, int ld_ctr, const float* nbr, int ld_nbr, const int* idx, int n, int k,
                               int c, float* emax, int ld_emax, void* sums, void* stream) {
    PCRCG_CHECK_ARG(n >= 1 && k >= 1 && c >= 1 && ld_ctr >= c && ld_nbr >= c && ld_emax >= c);
    PCRCG_CHECK_ARG(ctr && nbr && idx && emax && sums);
    {
        const EdgeCloud one = {ctr, nbr, idx, emax, static_cast<double*>(sums), n, k};
        if (edgeconv_rows_ok(&one, 1, ld_ctr, ld_nbr, ld_emax, c))
            return edgeconv_rows_multi(&one, 1, ld_ctr, ld_nbr, ld_emax, c, true, nullptr, as_stream(stream));
    }
    int chunks = (n + 15) / 16;                      // ~16 rows per workgroup row slice
    if (chunks > colstats_chunks()) chunks = colstats_chunks();
    hipLaunchKernelGGL(k_edgeconv_reduce<true>, dim3(chunks, (c + 63) / 64), dim3(256), 0, as_stream(stream), ctr, ld_ctr,
                       nbr, ld_nbr, idx, n, k, c, emax, ld_emax, static_cast<double*>(sums));
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

int pcrcg_attention_backward_supported(int n, int ms, int d, int ldq, int ldk, int ldv, int ld_do) {
    return attention_bwd_mfma_ok(n, ms, d, ldq, ldk, ldv, ld_do) ? 1 : 0;
}
int pcrcg_attention_backward(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* out, int ldo,
                             const float* d_out, int ld_do, float* dq, int ld_dq, float* dk, int ld_dk, float* dv, int ld_dv,
                             int n, int ms, int heads, int d, float scale, void* stream) {
    PCRCG_CHECK_ARG(n >= 0 && ms >= 1 && heads >= 1);
    if (n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(ldq >= heads * d && ldk >= heads * d && ldv >= heads * d && ldo >= heads * d && ld_do >= heads * d &&
                    ld_dq >= heads * d && ld_dk >= heads * d && ld_dv >= heads * d);
    return attention_bwd_mfma(q, ldq, k, ldk, v, ldv, out, ldo, d_out, ld_do, dq, ld_dq, dk, ld_dk, dv, ld_dv, n, ms, heads, d, scale,
                              as_stream(stream));
}

int pcrcg_attention_supported(int d) { return d == 16 || d == 32 || d == 48 || d == 64 || d == 128; }

int pcrcg_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, int n,
                    int ms, int heads, int d, float scale, void* stream) {
    PCRCG_CHECK_ARG(n >= 0 && ms >= 1 && heads >= 1 && pcrcg_attention_supported(d));
    PCRCG_CHECK_ARG(ldq >= heads * d && ldk >= heads * d && ldv >= heads * d && ldo >= heads * d);
    PCRCG_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0);
    if (n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(q && k && v && out);
    PCRCG_CHECK_ARG(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15) == 0);
    hipStream_t st = as_stream(stream);
    {
        const AttnCloud one = {q, k, v, out, n, ms};
        if (attention_mfma_ok(&one, 1, ldq, ldk, ldv, d)) return attention_mfma_multi(&one, 1, ldq, ldk, ldv, ldo, heads, d, scale, st);
    }
#define ATT(DD, TQ)                                                                                                  \
    do {                                                                                                             \
        constexpr size_t lds = sizeof(float) * (64 * (DD + 4) + 64 * DD + TQ * 64);                                  \
        if (lds > 64 * 1024) PCRCG_GRANT_LDS((k_attention<DD, TQ>));                              \
        hipLaunchKernelGGL((k_attention<DD, TQ>), dim3((n + TQ - 1) / TQ, heads), dim3(256), lds, st, q, ldq, k, ldk, v, \
                           ldv, out, ldo, n, ms, scale);                                                             \
    } while (0)
    // d = 128: 16 queries x 16 lanes.  8 x 32 is faster alone (50 vs 62 us on 381 x 382 x 4 heads) but takes twice the
    // workgroups and re-reads K / V twice as often, and inside the four-stream engine that costs more than it gains
    // (440 vs 446 pairs/s; 32 x 8 lanes: 89 us alone, 443) -- DebugOpts::att_tq = 8 selects it for a lone forward.
    const int att_tq = debug_opts().att_tq;
    if (d == 128 && att_tq == 8) ATT(128, 8);
    else if (d == 128) ATT(128, 16);
    else if (d == 64) ATT(64, 8);
    else if (d == 48) ATT(48, 16);
    else if (d == 32) ATT(32, 16);
    else ATT(16, 16);
#undef ATT
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

int pcrcg_softmax_matvec(const float* x, int rows, int cols, int ld, float scale, const float* vec, int ldv, float* y,
                         int ldy, void* stream) {
    PCRCG_CHECK_ARG(rows >= 0 && cols >= 1 && ld >= cols && ldv >= 1 && ldy >= 1);
    if (rows == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(x && vec && y);
    hipLaunchKernelGGL(k_softmax_matvec, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, rows, cols, ld, scale, vec,
                       ldv, y, ldy);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

int pcrcg_softmax_rows(float* x, int rows, int cols, int ld, float scale, void* stream) {
    PCRCG_CHECK_ARG(rows >= 0 && cols >= 1 && ld >= cols);
    if (rows == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(x != nullptr);
    hipLaunchKernelGGL(k_softmax_rows, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, rows, cols, ld, scale);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
}
