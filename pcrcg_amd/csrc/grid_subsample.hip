// grid_subsample.hip -- voxel-grid barycentre subsampling of stacked clouds on gfx950.
//
// Replaces batch_grid_subsampling (zip:cpp_subsampling/grid_subsampling/grid_subsampling.cpp:109-211,
// single-cloud core :5-106).  The result must be bit-identical to the reference INCLUDING the row
// order, which the reference inherits from iterating a std::unordered_map<size_t,SampledData>
// (:48,:85).  Pipeline (all on the caller's stream, no host round trip):
//
//   1 cloud offsets + per-cloud min/max (ordered-uint atomics, wave pre-reduced)
//   2 per point: fp32 cell key exactly as :27-31,:53-56 -> insert into a per-cloud open-addressing
//     hash table (one 64-bit CAS), remember slot, atomicMin first-occurrence index, atomicAdd count
//   3 flag first occurrences, device scan -> cell id in first-occurrence order (= the order in which
//     the reference emplaces keys into its map)
//   4 scan of counts -> per-cell segments; scatter point ids; per cell: sort the (few) ids ascending
//     and add the points sequentially in input order (fp32 sums are order sensitive, :59-70), then
//     multiply by float(1.0/count) (:87)
//   5 per cloud, one workgroup: reproduce libstdc++'s unordered_map iteration order with a
//     data-parallel restatement of its insert/rehash rules (see umap_order_block) and emit rows.
//
// This file is compiled with -ffp-contract=off: the reference is built without FMA.
#include "block_scan.h"
#include "common.h"

namespace pcrcg {
namespace {

typedef unsigned long long u64;
constexpr u64 kEmptyKey = ~0ull;
constexpr int kInfIdx = 0x7F7F7F7F;  // hipMemsetAsync byte pattern 0x7F

// libstdc++ prime bucket counts reached by doubling from 1 (probed with g++ 11.4, see oracle/front_end.c)
__constant__ u64 kGrow[23] = {1ull,       13ull,      29ull,      59ull,       127ull,      257ull,
                              541ull,     1109ull,    2357ull,    5087ull,     10273ull,    20753ull,
                              42043ull,   85229ull,   172933ull,  351061ull,   712697ull,   1447153ull,
                              2938679ull, 5967347ull, 12117689ull, 24607243ull, 49969847ull};
constexpr int kNGrow = 23;

__device__ __forceinline__ unsigned enc_f32(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}
// IEEE-754 correctly rounded fp32 quotient irrespective of compiler flags: a double quotient of two
// floats rounds to the same float as the exact quotient (53 >= 2*24+2).
__device__ __forceinline__ float div_rn(float a, float b) { return (float)((double)a / (double)b); }
__device__ __forceinline__ u64 to_size_t(float v) { return (u64)(long long)v; }

__device__ __forceinline__ int cloud_of(const int* __restrict__ coff, int nb, int i) {
    int lo = 0, hi = nb - 1;  // largest b with coff[b] <= i
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (coff[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// cloud offsets, min/max seeds and every table reset in one launch (instead of a kernel and 5 memsets)
__global__ void __launch_bounds__(256) k_init(const int* __restrict__ len, int nb, int* __restrict__ coff,
                                               unsigned* __restrict__ mm, u64* __restrict__ tkey, int* __restrict__ tfirst,
                                               int* __restrict__ tcnt, int* __restrict__ ccnt, int* __restrict__ cfill,
                                               int* __restrict__ mtot, long n1) {
    const long t0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    if (t0 == 0) {
        int s = 0;
        for (int b = 0; b < nb; ++b) { coff[b] = s; s += len[b]; }
        coff[nb] = s;
        *mtot = 0;
    }
    for (long i = t0; i < nb * 6; i += stride) mm[i] = (i % 6) < 3 ? 0xFFFFFFFFu : 0u;
    for (long i = t0; i < 2 * n1; i += stride) { tkey[i] = kEmptyKey; tfirst[i] = kInfIdx; tcnt[i] = 0; }
    for (long i = t0; i < n1; i += stride) { ccnt[i] = 0; cfill[i] = 0; }
}

// Bounding box per cloud.  Grid-stride over the points; a wavefront reduces what it saw of one cloud with
// shuffles and issues one global atomic per component when its slice of that cloud ends (the first version launched
// 235 workgroups whose every wavefront did so: ~5 600 atomics on six words serialised in one L2 channel, 33 us for
// 60 000 points; 64 workgroups issue ~1 500).
constexpr int kMinmaxBlocks = 64;
// (Every kernel below takes the number of points from coff[nb] -- the sum of the cloud lengths, which live on the
// device -- and its launch grid from a host-side BOUND: the pyramid builder sizes a level from the previous level's
// bound, not from a row count read back from the GPU.)
__global__ void __launch_bounds__(256) k_minmax(const float* __restrict__ pts, const int* __restrict__ coff,
                                                 int nb, unsigned* __restrict__ mm) {
    const int n = coff[nb];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // a workgroup walks a CONTIGUOUS slice of the points, so it sees at most a couple of clouds: the running
    // reduction is flushed whenever the cloud changes
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int begin = blockIdx.x * per, end = min(n, begin + per);
    int cur = -1;
    unsigned lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    auto flush = [&](int cloud) {
        // all lanes of the wavefront hold partial results of `cloud` (or the neutral element)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            unsigned l = lo[d], h = hi[d];
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) {
                l = min(l, (unsigned)__shfl_xor((int)l, s, 64));
                h = max(h, (unsigned)__shfl_xor((int)h, s, 64));
            }
            if (lane == 0 && cloud >= 0 && h >= l) {
                atomicMin(&mm[cloud * 6 + d], l);
                atomicMax(&mm[cloud * 6 + 3 + d], h);
            }
            lo[d] = 0xFFFFFFFFu;
            hi[d] = 0u;
        }
    };
    for (int base = begin + wave * 64; base < end; base += 256) {
        const int i = base + lane;
        const bool valid = i < end;
        const int b = valid ? cloud_of(coff, nb, i) : -1;
        // wave-uniform cloud for all valid lanes?  (clouds are contiguous, so almost always)
        const int b0 = __shfl(b, 0, 64);
        const bool uniform = __all(!valid || b == b0);
        if (uniform) {
            if (b0 != cur) { flush(cur); cur = b0; }
            if (valid) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const unsigned e = enc_f32(pts[3 * (long)i + d]);
                    lo[d] = min(lo[d], e);
                    hi[d] = max(hi[d], e);
                }
            }
        } else if (valid) {       // the rare wavefront that straddles two clouds: per-lane atomics
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const unsigned e = enc_f32(pts[3 * (long)i + d]);
                atomicMin(&mm[b * 6 + d], e);
                atomicMax(&mm[b * 6 + 3 + d], e);
            }
        }
    }
    flush(cur);
}

__device__ __forceinline__ unsigned mix32(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (unsigned)x;
}

__global__ void __launch_bounds__(256) k_cell_insert(const float* __restrict__ pts, const int* __restrict__ coff,
                                                      int nb, const unsigned* __restrict__ mm, float dl, float inv_dl,
                                                      u64* __restrict__ tkey, int* __restrict__ tfirst,
                                                      int* __restrict__ tcnt, int* __restrict__ slot_of,
                                                      u64* __restrict__ pkey) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= coff[nb]) return;
    const int b = cloud_of(coff, nb, i);
    const float mnx = dec_f32(mm[b * 6 + 0]), mny = dec_f32(mm[b * 6 + 1]), mnz = dec_f32(mm[b * 6 + 2]);
    const float mxx = dec_f32(mm[b * 6 + 3]), mxy = dec_f32(mm[b * 6 + 4]);
    // originCorner = floor(minCorner * (1/sampleDl)) * sampleDl            (:27)
    const float ox = floorf(mnx * inv_dl) * dl, oy = floorf(mny * inv_dl) * dl, oz = floorf(mnz * inv_dl) * dl;
    // sampleNX / sampleNY                                                  (:30-31)
    const u64 nX = to_size_t(floorf(div_rn(mxx - ox, dl))) + 1;
    const u64 nY = to_size_t(floorf(div_rn(mxy - oy, dl))) + 1;
    const float x = pts[3 * (long)i], y = pts[3 * (long)i + 1], z = pts[3 * (long)i + 2];
    const u64 iX = to_size_t(floorf(div_rn(x - ox, dl)));                  // :53-55
    const u64 iY = to_size_t(floorf(div_rn(y - oy, dl)));
    const u64 iZ = to_size_t(floorf(div_rn(z - oz, dl)));
    const u64 key = iX + nX * iY + nX * nY * iZ;                            // :56
    pkey[i] = key;
    // per-cloud table region [2*coff[b], 2*coff[b+1])
    const unsigned tsize = 2u * (unsigned)(coff[b + 1] - coff[b]);
    const long tbase = 2l * coff[b];
    unsigned s = __umulhi(mix32(key), tsize);
    for (;;) {
        u64 prev = atomicCAS(&tkey[tbase + s], kEmptyKey, key);
        if (prev == kEmptyKey || prev == key) break;
        s = s + 1 == tsize ? 0 : s + 1;
    }
    const int slot = (int)(tbase + s);
    slot_of[i] = slot;
    atomicMin(&tfirst[slot], i);
    atomicAdd(&tcnt[slot], 1);
}

// (positions between the point count and the bound get 0: the scan runs over the bound)
__global__ void __launch_bounds__(256) k_flag(const int* __restrict__ n_dev, int n_bound, const int* __restrict__ slot_of,
                                               const int* __restrict__ tfirst, int* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < *n_dev) flag[i] = aload(&tfirst[slot_of[i]]) == i ? 1 : 0;
    else if (i < n_bound) flag[i] = 0;
}

// rank = exclusive scan of flag.  A point with rank[i+1] != rank[i] (or the last one with total)
// is the first occurrence of cell rank[i].
__global__ void __launch_bounds__(256) k_cells(const int* __restrict__ n_dev, const int* __restrict__ slot_of, const int* __restrict__ tfirst,
                                                const int* __restrict__ tcnt, const int* __restrict__ rank,
                                                const u64* __restrict__ pkey, u64* __restrict__ ckey,
                                                int* __restrict__ ccnt, int* __restrict__ trank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_dev) return;
    const int s = slot_of[i];
    if (aload(&tfirst[s]) == i) {
        const int c = rank[i];
        ckey[c] = pkey[i];
        ccnt[c] = aload(&tcnt[s]);
        trank[s] = c;
    }
}

__global__ void __launch_bounds__(256) k_fill(const int* __restrict__ n_dev, const int* __restrict__ slot_of, const int* __restrict__ trank,
                                               const int* __restrict__ cstart, int* __restrict__ cfill,
                                               int* __restrict__ cidx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_dev) return;
    const int c = trank[slot_of[i]];
    const int pos = cstart[c] + atomicAdd(&cfill[c], 1);
    cidx[pos] = i;
}

// One thread per cell: order the cell's point ids ascending (= input order) and accumulate.
__global__ void __launch_bounds__(256) k_barycentres(const float* __restrict__ pts, const int* __restrict__ mtot,
                                                      const int* __restrict__ cstart, const int* __restrict__ ccnt,
                                                      int* __restrict__ cidx, float* __restrict__ cbary) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= *mtot) return;
    int* seg = cidx + cstart[c];
    const int cnt = ccnt[c];
    for (int a = 1; a < cnt; ++a) {  // insertion sort; cells hold a handful of points
        int v = seg[a], j = a - 1;
        while (j >= 0 && seg[j] > v) { seg[j + 1] = seg[j]; --j; }
        seg[j + 1] = v;
    }
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;  // SampledData(): point = PointXYZ() = 0      (grid_subsampling.h:24-28)
    for (int a = 0; a < cnt; ++a) {
        const long i = seg[a];
        sx += pts[3 * i];                    // point += p                                   (:74-79)
        sy += pts[3 * i + 1];
        sz += pts[3 * i + 2];
    }
    const float w = (float)(1.0 / (double)cnt);  // operator*(PointXYZ, float) narrows 1.0/count   (:87, cloud.h:120)
    cbary[3 * (long)c] = sx * w;
    cbary[3 * (long)c + 1] = sy * w;
    cbary[3 * (long)c + 2] = sz * w;
}

// ------------------------------------------------------------------------------------------------
// libstdc++ unordered_map iteration order, data-parallel.
//
// libstdc++ keeps ONE singly linked list of nodes; a bucket remembers the node before its first
// node.  Insert (hashtable.h _M_insert_bucket_begin): non-empty bucket -> splice at the FRONT of the
// bucket's run; empty bucket -> splice at the list HEAD.  Rehash (_M_rehash_aux, unique keys) walks
// the old list from the head and re-inserts with the same two rules; it happens right before the
// insert that would make size() exceed the bucket count, and the new count is the next entry of
// kGrow.  Consequence: after a rehash to B buckets, give every element a time stamp t (position in
// the walk for old elements, then insertion rank for new ones -- the two coincide in numbering
// because the walk has exactly as many elements as were inserted before).  Buckets appear in the
// list in DESCENDING order of the first time stamp that hit them, and inside a bucket elements
// appear in DESCENDING time stamp.  So each growth stage is a counting sort by
// (first_time[bucket] desc, t desc), done here with atomics + two block scans.
// ------------------------------------------------------------------------------------------------
struct UmapWs {
    int *ord_a, *ord_b;      // [m] list order ping-pong
    int *bkt, *frank, *cbr, *start, *member;  // [m]
    int *first, *cnt, *fill, *brank;          // [maxB]
};

constexpr int kUmapThreads = 1024;

constexpr int kUmapLds = 1109;   // growth stages up to this bucket count (13 ... 1109: seven stages) run out of LDS

// `wg` = global scratch (any size); the first growth stages are a few hundred elements each and pure latency, so
// their scratch lives in LDS (`s_lds`, 11 * kUmapLds ints) -- same code, the pointers are generic.
__device__ void umap_order_block(const u64* __restrict__ key, int m, UmapWs wg, int* __restrict__ smem, int* s_lds) {
    const int tid = threadIdx.x;
    UmapWs wl;
    wl.ord_a = s_lds + 0 * kUmapLds; wl.ord_b = s_lds + 1 * kUmapLds; wl.bkt = s_lds + 2 * kUmapLds;
    wl.frank = s_lds + 3 * kUmapLds; wl.cbr = s_lds + 4 * kUmapLds; wl.start = s_lds + 5 * kUmapLds;
    wl.member = s_lds + 6 * kUmapLds; wl.first = s_lds + 7 * kUmapLds; wl.cnt = s_lds + 8 * kUmapLds;
    wl.fill = s_lds + 9 * kUmapLds; wl.brank = s_lds + 10 * kUmapLds;
    int cur = 0;
    int* ord = wl.ord_a;
    for (int s = 1; cur < m && s < kNGrow; ++s) {
        const u64 B = kGrow[s];
        const int cur2 = (u64)m < B ? m : (int)B;
        const UmapWs w = B <= (u64)kUmapLds ? wl : wg;
        int* nord = ord == w.ord_a ? w.ord_b : w.ord_a;
        for (u64 b = tid; b < B; b += kUmapThreads) { w.first[b] = kInfIdx; w.cnt[b] = 0; w.fill[b] = 0; }
        __syncthreads();
        for (int t = tid; t < cur2; t += kUmapThreads) {
            const int e = t < cur ? ord[t] : t;
            const int b = (int)(key[e] % B);
            w.bkt[t] = b;
            atomicMin(&w.first[b], t);
            atomicAdd(&w.cnt[b], 1);
        }
        __syncthreads();
        // rank the non-empty buckets by first time stamp (ascending)
        int carry = 0;
        for (int base = 0; base < cur2; base += kUmapThreads) {
            const int t = base + tid;
            int f = 0, b = 0;
            if (t < cur2) { b = w.bkt[t]; f = aload(&w.first[b]) == t ? 1 : 0; }
            int tot;
            const int ex = block_excl_scan_i32<kUmapThreads>(f, &tot, smem);
            if (f) {
                const int r = carry + ex;
                w.frank[t] = r;
                w.brank[b] = r;
                w.cbr[r] = aload(&w.cnt[b]);
            }
            carry += tot;
        }
        const int nbk = carry;
        __syncthreads();
        // start[r] = number of elements in buckets with LARGER rank (they come earlier in the list)
        carry = 0;
        for (int base = 0; base < nbk; base += kUmapThreads) {
            const int i = base + tid;
            const int r = nbk - 1 - i;
            const int v = i < nbk ? w.cbr[r] : 0;
            int tot;
            const int ex = block_excl_scan_i32<kUmapThreads>(v, &tot, smem);
            if (i < nbk) w.start[r] = carry + ex;
            carry += tot;
        }
        __syncthreads();
        for (int t = tid; t < cur2; t += kUmapThreads) {
            const int b = w.bkt[t];
            const int pos = w.start[w.brank[b]] + atomicAdd(&w.fill[b], 1);
            w.member[pos] = t;
        }
        __syncthreads();
        // inside a bucket: descending time stamp
        for (int t = tid; t < cur2; t += kUmapThreads) {
            const int b = w.bkt[t];
            if (aload(&w.first[b]) != t) continue;
            const int r = w.frank[t];
            const int c = w.cbr[r];
            int* seg = w.member + w.start[r];
            for (int a = 1; a < c; ++a) {
                int v = seg[a], j = a - 1;
                while (j >= 0 && seg[j] < v) { seg[j + 1] = seg[j]; --j; }
                seg[j + 1] = v;
            }
            for (int a = 0; a < c; ++a) {
                const int tt = seg[a];
                nord[w.start[r] + a] = tt < cur ? ord[tt] : tt;
            }
        }
        __syncthreads();
        ord = nord;
        cur = cur2;
    }
    // leave the final order in the global ord_a
    if (ord != wg.ord_a)
        for (int t = tid; t < m; t += kUmapThreads) wg.ord_a[t] = ord[t];
    __syncthreads();
}

__device__ __forceinline__ UmapWs umap_carve(int* ebase, int* bbase, long estride, long bstride, long eoff, long boff) {
    UmapWs w;
    w.ord_a = ebase + 0 * estride + eoff;
    w.ord_b = ebase + 1 * estride + eoff;
    w.bkt = ebase + 2 * estride + eoff;
    w.frank = ebase + 3 * estride + eoff;
    w.cbr = ebase + 4 * estride + eoff;
    w.start = ebase + 5 * estride + eoff;
    w.member = ebase + 6 * estride + eoff;
    w.first = bbase + 0 * bstride + boff;
    w.cnt = bbase + 1 * bstride + boff;
    w.fill = bbase + 2 * bstride + boff;
    w.brank = bbase + 3 * bstride + boff;
    return w;
}

// One workgroup per cloud: order the cloud's cells, cap at max_p, write rows and lengths.
// out_cap rows fit behind out_pts: a cloud whose rows would not is cut there and *overflow is raised (the caller
// reports PCRCG_EWORKSPACE; with out_cap >= the number of input points that cannot happen).
__global__ void __launch_bounds__(kUmapThreads) k_order_emit(int nb, const int* __restrict__ coff,
                                                              const int* __restrict__ rank, const int* __restrict__ mtot,
                                                              const u64* __restrict__ ckey, const float* __restrict__ cbary,
                                                              int* __restrict__ ebase, int* __restrict__ bbase,
                                                              long estride, long bstride, int max_p,
                                                              float* __restrict__ out_pts, int* __restrict__ out_len,
                                                              int* __restrict__ out_m, int out_cap, int* __restrict__ overflow) {
    __shared__ int smem[kUmapThreads / 64];
    __shared__ int s_lds[11 * kUmapLds];
    const int b = blockIdx.x;
    const int mt = *mtot;
    const int n = coff[nb];
    // cells of clouds < b precede this cloud's cells (first-occurrence order is input order)
    int out_off = 0, cell_lo = 0, cell_hi = 0;
    for (int bb = 0; bb <= b; ++bb) {
        const int lo = coff[bb] < n ? rank[coff[bb]] : mt;
        const int hi = coff[bb + 1] < n ? rank[coff[bb + 1]] : mt;
        if (bb < b) out_off = min(out_off + min(hi - lo, max_p), out_cap);
        else { cell_lo = lo; cell_hi = hi; }
    }
    const int m = cell_hi - cell_lo;
    UmapWs w = umap_carve(ebase, bbase, estride, bstride, cell_lo, 3l * cell_lo + 16l * b);
    umap_order_block(ckey + cell_lo, m, w, smem, s_lds);
    int keep = min(m, max_p);                                               // :185-205
    if (out_off + keep > out_cap) {
        keep = max(0, out_cap - out_off);
        if (threadIdx.x == 0 && overflow) *overflow = 1;
    }
    for (int j = threadIdx.x; j < keep; j += kUmapThreads) {
        const long c = cell_lo + w.ord_a[j];
        out_pts[3 * (long)(out_off + j)] = cbary[3 * c];
        out_pts[3 * (long)(out_off + j) + 1] = cbary[3 * c + 1];
        out_pts[3 * (long)(out_off + j) + 2] = cbary[3 * c + 2];
    }
    if (threadIdx.x == 0) {
        out_len[b] = keep;
        if (b == nb - 1) *out_m = out_off + keep;
    }
}

__global__ void __launch_bounds__(kUmapThreads) k_umap_only(const u64* __restrict__ key, int m, int* __restrict__ ebase,
                                                             int* __restrict__ bbase, long estride, long bstride,
                                                             int* __restrict__ order) {
    __shared__ int smem[kUmapThreads / 64];
    __shared__ int s_lds[11 * kUmapLds];
    UmapWs w = umap_carve(ebase, bbase, estride, bstride, 0, 0);
    umap_order_block(key, m, w, smem, s_lds);
    for (int j = threadIdx.x; j < m; j += kUmapThreads) order[j] = w.ord_a[j];
}

inline size_t umap_bucket_cap(int n, int nb) { return (size_t)3 * (size_t)n + (size_t)16 * (size_t)nb + 64; }

}  // namespace
}  // namespace pcrcg

using namespace pcrcg;

extern "C" {

size_t pcrcg_umap_order_ws_bytes(int m) {
    if (m < 0) m = 0;
    return 7 * carve_bytes((size_t)m + 1, sizeof(int)) + 4 * carve_bytes(umap_bucket_cap(m, 1), sizeof(int));
}

int pcrcg_umap_order(const uint64_t* keys, int m, int* order, void* ws, size_t ws_bytes, void* stream) {
    PCRCG_CHECK_ARG(m >= 0);
    if (m == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(keys && order && ws);
    PCRCG_CHECK_ARG((unsigned long long)m <= 49969847ull);
    Carver cv(ws, ws_bytes);
    const size_t es = carve_bytes((size_t)m + 1, sizeof(int)) / sizeof(int);
    const size_t bs = carve_bytes(umap_bucket_cap(m, 1), sizeof(int)) / sizeof(int);
    int* ebase = cv.take<int>(7 * es);
    int* bbase = cv.take<int>(4 * bs);
    PCRCG_CHECK_WS(cv);
    hipLaunchKernelGGL(k_umap_only, dim3(1), dim3(kUmapThreads), 0, as_stream(stream),
                       reinterpret_cast<const u64*>(keys), m, ebase, bbase, (long)es, (long)bs, order);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

size_t pcrcg_grid_subsample_ws_bytes(int n, int nb) {
    if (n < 0) n = 0;
    if (nb < 0) nb = 0;
    const size_t N = (size_t)n + 1;
    size_t b = 0;
    b += carve_bytes((size_t)nb + 1, sizeof(int));      // coff
    b += carve_bytes((size_t)nb * 6, sizeof(unsigned)); // mm
    b += carve_bytes(2 * N, sizeof(u64));               // tkey
    b += 3 * carve_bytes(2 * N, sizeof(int));           // tfirst, tcnt, trank
    b += carve_bytes(N, sizeof(u64)) * 2;               // pkey, ckey
    b += 6 * carve_bytes(N, sizeof(int));               // slot, rank, ccnt, cstart, cfill, cidx
    b += carve_bytes(3 * N, sizeof(float));             // cbary
    b += carve_bytes(1, sizeof(int));                   // mtot
    b += 7 * carve_bytes(N, sizeof(int));               // umap element arrays
    b += 4 * carve_bytes(umap_bucket_cap(n, nb), sizeof(int));
    b += scan_ws_bytes(n + 1);
    return b;
}

int pcrcg_grid_subsample_batch(const float* pts, int n, const int* len, int nb, float dl, int max_p,
                               float* out_pts, int* out_len, int* out_m, void* ws, size_t ws_bytes,
                               void* stream) {
    return pcrcg::grid_subsample_bound(pts, n, len, nb, dl, max_p, out_pts, out_len, out_m, n, nullptr, ws, ws_bytes,
                                       as_stream(stream));
}
}

// n = a BOUND on the number of points (the clouds' lengths, on the device, say how many there are); out_cap rows fit
// behind out_pts (fewer than the result: *overflow = 1, the output is cut there).
int pcrcg::grid_subsample_bound(const float* pts, int n, const int* len, int nb, float dl, int max_p, float* out_pts,
                                int* out_len, int* out_m, int out_cap, int* overflow, void* ws, size_t ws_bytes,
                                hipStream_t stream) {
    PCRCG_CHECK_ARG(n >= 0 && nb >= 1 && out_cap >= 0);
    PCRCG_CHECK_ARG(pts && len && out_pts && out_len && out_m && ws);
    PCRCG_CHECK_ARG(dl > 0.0f);
    hipStream_t st = stream;
    if (max_p < 1) max_p = n;  // :134-135
    const size_t N = (size_t)n + 1;
    Carver cv(ws, ws_bytes);
    int* coff = cv.take<int>((size_t)nb + 1);
    unsigned* mm = cv.take<unsigned>((size_t)nb * 6);
    u64* tkey = cv.take<u64>(2 * N);
    int* tfirst = cv.take<int>(2 * N);
    int* tcnt = cv.take<int>(2 * N);
    int* trank = cv.take<int>(2 * N);
    u64* pkey = cv.take<u64>(N);
    u64* ckey = cv.take<u64>(N);
    int* slot_of = cv.take<int>(N);
    int* rank = cv.take<int>(N);
    int* ccnt = cv.take<int>(N);
    int* cstart = cv.take<int>(N);
    int* cfill = cv.take<int>(N);
    int* cidx = cv.take<int>(N);
    float* cbary = cv.take<float>(3 * N);
    int* mtot = cv.take<int>(1);
    const size_t es = carve_bytes(N, sizeof(int)) / sizeof(int);
    const size_t bs = carve_bytes(umap_bucket_cap(n, nb), sizeof(int)) / sizeof(int);
    int* ebase = cv.take<int>(7 * es);
    int* bbase = cv.take<int>(4 * bs);
    void* scan_ws = cv.take<char>(scan_ws_bytes(n + 1));
    PCRCG_CHECK_WS(cv);

    const int blocks = n > 0 ? (n + 255) / 256 : 1;
    const int init_blocks = (int)((2 * N + 255) / 256 < 1024 ? (2 * N + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_init, dim3(init_blocks), dim3(256), 0, st, len, nb, coff, mm, tkey, tfirst, tcnt, ccnt, cfill,
                       mtot, (long)N);
    if (n > 0) {
        const float inv_dl = 1 / dl;  // (1/sampleDl): int/float -> fp32 division on the host (:27)
        const int* n_dev = coff + nb;
        hipLaunchKernelGGL(k_minmax, dim3(blocks < kMinmaxBlocks ? blocks : kMinmaxBlocks), dim3(256), 0, st, pts, coff, nb, mm);
        hipLaunchKernelGGL(k_cell_insert, dim3(blocks), dim3(256), 0, st, pts, coff, nb, mm, dl, inv_dl, tkey,
                           tfirst, tcnt, slot_of, pkey);
        hipLaunchKernelGGL(k_flag, dim3(blocks), dim3(256), 0, st, n_dev, n, slot_of, tfirst, rank);
        PCRCG_PROPAGATE(exclusive_scan_i32(rank, rank, n, mtot, scan_ws, st));
        hipLaunchKernelGGL(k_cells, dim3(blocks), dim3(256), 0, st, n_dev, slot_of, tfirst, tcnt, rank, pkey, ckey, ccnt,
                           trank);
        PCRCG_PROPAGATE(exclusive_scan_i32(ccnt, cstart, n, nullptr, scan_ws, st));
        hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, st, n_dev, slot_of, trank, cstart, cfill, cidx);
        hipLaunchKernelGGL(k_barycentres, dim3(blocks), dim3(256), 0, st, pts, mtot, cstart, ccnt, cidx, cbary);
    }
    hipLaunchKernelGGL(k_order_emit, dim3(nb), dim3(kUmapThreads), 0, st, nb, coff, rank, mtot, ckey, cbary, ebase,
                       bbase, (long)es, (long)bs, max_p, out_pts, out_len, out_m, out_cap, overflow);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
