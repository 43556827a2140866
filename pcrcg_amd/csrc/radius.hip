// radius.hip -- batched radius-neighbour search on gfx950.
//
// Replaces batch_nanoflann_neighbors (ref:cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-333)
// plus the column truncation / int64 cast of batch_neighbors_kpconv (ref:datasets/dataloader.py:54-69,
// 347-349).  The reference builds a KD-tree per cloud and per call; here the supports of a level are
// binned ONCE into a hashed uniform cell grid (edge = radius) that serves all query sets of that
// radius, and each query is handled by one 64-lane wavefront:
//
//   build : per support -> 63-bit cell key -> open-addressing insert into the cloud's table region
//           (one 64-bit CAS) + in-cell position; per occupied slot a start offset from one global
//           cursor; scatter supports as float4 (x,y,z,index) so that a cell is one contiguous,
//           16-byte-aligned run (coalesced reads)
//   query : lanes 0..26 probe the 27 neighbouring cells; the candidate runs are concatenated with a
//           wave prefix sum and swept 64 candidates at a time; hits (d2 < r2, the reference's exact
//           fp32 arithmetic) are compacted into LDS with ballot/popcount; a rank sort over the
//           (d2, index) keys in LDS writes the row in ascending order, truncated to `cols` and padded
//           with ns (:319-325).
//
// Squared distances follow nanoflann's L2_Simple_Adaptor (zip:cpp_utils/nanoflann/nanoflann.hpp:
// 432-440): ((0 + dx*dx) + dy*dy) + dz*dz with every product and sum rounded to fp32, strict
// d2 < r*r (:249-253, neighbors.cpp:226).  Compiled with -ffp-contract=off.
#include "block_scan.h"
#include <cstdlib>

#include "common.h"

namespace pcrcg {
namespace {

typedef unsigned long long u64;
constexpr u64 kEmptyKey = ~0ull;
constexpr int kCoordBias = 1 << 20;   // cell coordinates are stored biased, 21 bits each
constexpr int kListCapFast = 256;     // staged hits per query in the first pass (8 KiB of LDS per workgroup)
constexpr int kListCapFull = 1024;    // second pass for the rare longer lists (reference bound: hist_n = 905,
                                      // ref:datasets/dataloader.py:407)
constexpr long long kRedoMark = -2;   // row[0] marker: list did not fit the first pass
constexpr int kQueryWaves = 4;        // waves (= queries in flight) per workgroup
constexpr int kTieCap = 256;          // tie rows staged per workgroup before they go to the global list

struct GridHeader {   // first 256 bytes of the grid workspace
    double inv_cell;  // 1 / (radius * (1 + 1e-5)): cells are a hair wider than the radius
    int ns, nb;
    int cursor;       // bump allocator for cell runs
    int overflow;     // coordinate range exceeded
};

struct GridView {
    GridHeader* hdr;
    int* soff;     // [nb+1]
    u64* tkey;     // [2*ns + 2]
    int* tcnt;     // [2*ns + 2]
    int* tstart;   // [2*ns + 2]
    int* slot_of;  // [ns]
    int* pos_in;   // [ns]
    float4* spts;  // [ns]
};

inline size_t grid_bytes(int ns, int nb) {
    const size_t N = (size_t)(ns > 0 ? ns : 0) + 1;
    return carve_bytes(1, 256) + carve_bytes((size_t)nb + 1, sizeof(int)) + carve_bytes(2 * N, sizeof(u64)) +
           2 * carve_bytes(2 * N, sizeof(int)) + 2 * carve_bytes(N, sizeof(int)) + carve_bytes(N, sizeof(float4));
}

inline GridView grid_view(void* ws, size_t bytes, int ns, int nb, bool* ok) {
    const size_t N = (size_t)(ns > 0 ? ns : 0) + 1;
    Carver cv(ws, bytes);
    GridView g;
    g.hdr = reinterpret_cast<GridHeader*>(cv.take<char>(256));
    g.soff = cv.take<int>((size_t)nb + 1);
    g.tkey = cv.take<u64>(2 * N);
    g.tcnt = cv.take<int>(2 * N);
    g.tstart = cv.take<int>(2 * N);
    g.slot_of = cv.take<int>(N);
    g.pos_in = cv.take<int>(N);
    g.spts = cv.take<float4>(N);
    *ok = cv.ok();
    return g;
}

__device__ __forceinline__ unsigned mix32(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (unsigned)x;
}

__device__ __forceinline__ int cloud_of(const int* __restrict__ off, int nb, int i) {
    int lo = 0, hi = nb - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ bool cell_coords(float x, float y, float z, double inv_cell, int* cx, int* cy, int* cz) {
    const double fx = floor((double)x * inv_cell), fy = floor((double)y * inv_cell), fz = floor((double)z * inv_cell);
    const double lim = (double)(kCoordBias - 2);
    const bool ok = fx > -lim && fx < lim && fy > -lim && fy < lim && fz > -lim && fz < lim;
    *cx = ok ? (int)fx + kCoordBias : 0;
    *cy = ok ? (int)fy + kCoordBias : 0;
    *cz = ok ? (int)fz + kCoordBias : 0;
    return ok;
}
__device__ __forceinline__ u64 cell_key(int cx, int cy, int cz) {
    return (u64)(unsigned)cx | ((u64)(unsigned)cy << 21) | ((u64)(unsigned)cz << 42);
}

// header + cloud offsets + table reset in one launch (instead of a kernel and two memsets)
__global__ void __launch_bounds__(256) k_grid_init(GridView g, const int* __restrict__ slen, int ns, int nb,
                                                    double inv_cell, long nslots) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int s = 0;
        for (int b = 0; b < nb; ++b) { g.soff[b] = s; s += slen[b]; }
        g.soff[nb] = s;
        g.hdr->inv_cell = inv_cell;
        g.hdr->ns = ns;
        g.hdr->nb = nb;
        g.hdr->cursor = 0;
        g.hdr->overflow = 0;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (long)gridDim.x * blockDim.x) {
        g.tkey[i] = kEmptyKey;
        g.tcnt[i] = 0;
    }
}

__global__ void __launch_bounds__(256) k_grid_insert(const float* __restrict__ sup, int ns, int nb, GridView g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const int b = cloud_of(g.soff, nb, i);
    int cx, cy, cz;
    if (!cell_coords(sup[3 * (long)i], sup[3 * (long)i + 1], sup[3 * (long)i + 2], g.hdr->inv_cell, &cx, &cy, &cz))
        g.hdr->overflow = 1;
    const u64 key = cell_key(cx, cy, cz);
    const unsigned tsize = 2u * (unsigned)(g.soff[b + 1] - g.soff[b]);
    const long tbase = 2l * g.soff[b];
    unsigned s = __umulhi(mix32(key), tsize);
    for (;;) {
        u64 prev = atomicCAS(&g.tkey[tbase + s], kEmptyKey, key);
        if (prev == kEmptyKey || prev == key) break;
        s = s + 1 == tsize ? 0 : s + 1;
    }
    const int slot = (int)(tbase + s);
    g.slot_of[i] = slot;
    g.pos_in[i] = atomicAdd(&g.tcnt[slot], 1);
}

__global__ void __launch_bounds__(256) k_grid_starts(int nslots, GridView g) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    const int c = g.tcnt[s];
    if (c > 0) g.tstart[s] = atomicAdd(&g.hdr->cursor, c);
}

__global__ void __launch_bounds__(256) k_grid_scatter(const float* __restrict__ sup, int ns, GridView g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const int dst = g.tstart[g.slot_of[i]] + g.pos_in[i];
    g.spts[dst] = make_float4(sup[3 * (long)i], sup[3 * (long)i + 1], sup[3 * (long)i + 2], __int_as_float(i));
}

// One wavefront per query.  Pass 1 (REDO = false) handles every query with a small LDS list; a query
// whose list does not fit marks its row, and pass 2 (REDO = true, large list) redoes only marked rows.
template <int CAP, bool REDO, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_radius_query(
    const float* __restrict__ q, int nq, const int* __restrict__ qlen, int nb, float r2, GridView g, int cols,
    long long* __restrict__ out_idx, int* __restrict__ out_count, int* __restrict__ out_max, int* __restrict__ status,
    int* __restrict__ tie_rows, int* __restrict__ tie_count, int group) {
    __shared__ u64 s_list[WAVES][CAP];
    __shared__ u64 s_sorted[WAVES][CAP];
    __shared__ int s_excl[WAVES][32];
    __shared__ int s_start[WAVES][32];
    // rows that hold a tie are collected per workgroup and appended to the global list with ONE atomic per workgroup:
    // on voxelised scans most rows hold one, and one atomic per row on a single word cost 1 ms per 60k-row query
    __shared__ int s_tie[kTieCap];
    __shared__ int s_ntie, s_tie_base;
    if (threadIdx.x == 0) s_ntie = 0;
    __syncthreads();
    // (the wavefront index is uniform: readfirstlane lets the compiler keep the query, its cell and every
    // per-query address in SGPRs)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gw = blockIdx.x * WAVES + wave, nw = gridDim.x * WAVES;
    const int ns = g.hdr->ns;
    const double inv_cell = g.hdr->inv_cell;
    u64* list = s_list[wave];
    u64* sorted = s_sorted[wave];
    // `group` > 0: the clouds form independent groups of `group` clouds each (several fragment pairs stacked into one
    // call): indices are written relative to the first support of the query's group, rows are padded with the
    // group's support count and the longest list is tracked per group (out_max[g]) -- i.e. the table comes out as the
    // groups' own tables stacked on top of each other.  group == 0: one group (the reference's contract).
    int wave_max = 0, wave_grp = 0;
    auto flush_max = [&]() {   // one contended word per group: only the few waves that actually raise the maximum issue an atomic
        if (lane == 0 && wave_max > aload(out_max + wave_grp)) atomicMax(out_max + wave_grp, wave_max);
    };
    if (REDO && blockIdx.x == 0 && threadIdx.x == 0 && g.hdr->overflow && status) *status = 2;
    auto one_query = [&](const int qi) {
        // cloud of this query: walk the (few) query lengths
        int b = 0, qacc = 0;
        while (b < nb - 1 && qi >= qacc + qlen[b]) { qacc += qlen[b]; ++b; }
        const float qx = q[3 * (long)qi], qy = q[3 * (long)qi + 1], qz = q[3 * (long)qi + 2];
        int cx, cy, cz;
        const bool inrange = cell_coords(qx, qy, qz, inv_cell, &cx, &cy, &cz);
        const int nsb = g.soff[b + 1] - g.soff[b];
        int base = 0, ns_out = ns;
        if (group > 0) {
            const int grp = b / group, last = min(grp * group + group, nb);
            base = g.soff[grp * group];
            ns_out = g.soff[last] - base;
            if (grp != wave_grp) { flush_max(); wave_grp = grp; wave_max = 0; }
        }
        // lanes 0..26: look up one neighbouring cell each
        int ccount = 0, cstart = 0;
        if (lane < 27 && inrange && nsb > 0) {
            const int dx = lane % 3 - 1, dy = (lane / 3) % 3 - 1, dz = lane / 9 - 1;
            const u64 key = cell_key(cx + dx, cy + dy, cz + dz);
            const unsigned tsize = 2u * (unsigned)nsb;
            const long tbase = 2l * g.soff[b];
            unsigned s = __umulhi(mix32(key), tsize);
            for (unsigned probe = 0; probe < tsize; ++probe) {
                // key, count and start of the slot in ONE round trip (count / start of a foreign or empty slot are
                // loaded and dropped): the lookup is a chain of dependent global loads, and that latency -- not
                // bandwidth -- is what a query costs
                const u64 k = g.tkey[tbase + s];
                const int kc = g.tcnt[tbase + s], ks = g.tstart[tbase + s];
                if (k == key) { ccount = kc; cstart = ks; break; }
                if (k == kEmptyKey) break;
                s = s + 1 == tsize ? 0 : s + 1;
            }
        }
        const int incl = wave_incl_scan_i32(ccount, lane);
        const int total = __shfl(incl, 26, 64);
        if (lane < 32) { s_excl[wave][lane] = lane < 27 ? incl - ccount : 0x7FFFFFFF; s_start[wave][lane] = cstart; }
        __builtin_amdgcn_wave_barrier();
        int nhit = 0, fill = 0;     // hits found / hits currently staged (they differ only after a compaction)
        // Rows with more hits than the staging list holds (second pass only; the reference has no such bound): the
        // table keeps the `cols` nearest anyway, so whenever the list is about to overflow it is cut down to its
        // cols + 1 smallest keys (one more than is kept: the tie test at the cut needs the first dropped entry) -- a
        // streaming selection; the true count still goes to out_count / out_max.
        const bool can_compact = REDO && cols + 1 + 64 <= CAP;
        auto compact = [&]() {
            const int keep = cols + 1 < fill ? cols + 1 : fill;
            for (int e = lane; e < fill; e += 64) {
                const u64 mine = list[e];
                int rank = 0, j = 0;
                for (; j + 4 <= fill; j += 4) {
                    const u64 a = list[j], b = list[j + 1], c = list[j + 2], d = list[j + 3];
                    rank += (a < mine ? 1 : 0) + (b < mine ? 1 : 0) + (c < mine ? 1 : 0) + (d < mine ? 1 : 0);
                }
                for (; j < fill; ++j) rank += list[j] < mine ? 1 : 0;
                if (rank < keep) sorted[rank] = mine;
            }
            __builtin_amdgcn_wave_barrier();
            for (int e = lane; e < keep; e += 64) list[e] = sorted[e];
            __builtin_amdgcn_wave_barrier();
            fill = keep;
        };
        // candidates 256 at a time: the four gathers of a round are issued before the first is used (they are
        // independent; one after the other each would cost a full memory round trip)
        constexpr int SW = 4;
        for (int base = 0; base < total; base += 64 * SW) {
            float4 p[SW];
#pragma unroll
            for (int u = 0; u < SW; ++u) {
                const int t = base + 64 * u + lane;
                const int tc = t < total ? t : total - 1;   // clamped: loads stay branch-free
                int lo = 0;   // largest j in [0,27) with excl[j] <= tc  (5-step binary search over 32 entries)
#pragma unroll
                for (int step = 16; step >= 1; step >>= 1)
                    if (s_excl[wave][lo + step] <= tc) lo += step;
                p[u] = g.spts[s_start[wave][lo] + (tc - s_excl[wave][lo])];
            }
#pragma unroll
            for (int u = 0; u < SW; ++u) {
                if (base + 64 * u >= total) break;          // wave-uniform
                const int t = base + 64 * u + lane;
                const float d0 = qx - p[u].x, d1 = qy - p[u].y, d2c = qz - p[u].z;
                float d2 = 0.0f;
                d2 += d0 * d0;
                d2 += d1 * d1;
                d2 += d2c * d2c;
                const bool hit = t < total && d2 < r2;
                const u64 packed = ((u64)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(p[u].w);
                const u64 mask = __ballot(hit);
                if (can_compact && fill + 64 > CAP) compact();
                const int pos = fill + __popcll(mask & ((1ull << lane) - 1ull));
                if (hit && pos < CAP) list[pos] = packed;
                nhit += __popcll(mask);
                fill += __popcll(mask);
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int nl = fill < CAP ? fill : CAP;
        long long* row = out_idx + (long)qi * cols;
        if (!REDO && nhit > CAP) {   // leave the row to pass 2
            if (lane == 0) row[0] = kRedoMark;
            wave_max = nhit > wave_max ? nhit : wave_max;
            return;
        }
        // rank sort by (d2, index); keys are distinct.  Lists of up to 64 hits (nearly all) stay in registers: lane e
        // holds entry e and entry j is broadcast through an SGPR (v_readlane), no memory operation in the loop -- the
        // LDS version below waits for one ds_read per comparison (hipcc does not pipeline the loop: 43..77 dependent
        // LDS round trips per query were 45 % of this kernel's time)
        if (nl <= 64) {
            const u64 mine = lane < nl ? list[lane] : ~0ull;
            const unsigned mlo = (unsigned)mine, mhi = (unsigned)(mine >> 32);
            int rank = 0;
            for (int j = 0; j < nl; ++j) {
                const unsigned olo = __builtin_amdgcn_readlane(mlo, j), ohi = __builtin_amdgcn_readlane(mhi, j);
                rank += ((((u64)ohi) << 32) | olo) < mine ? 1 : 0;
            }
            if (lane < nl) sorted[rank] = mine;
        } else {
            for (int e = lane; e < nl; e += 64) {
                const u64 mine = list[e];
                int rank = 0, j = 0;
                for (; j + 4 <= nl; j += 4) {          // four independent LDS reads per step
                    const u64 a = list[j], b = list[j + 1], c = list[j + 2], d = list[j + 3];
                    rank += (a < mine ? 1 : 0) + (b < mine ? 1 : 0) + (c < mine ? 1 : 0) + (d < mine ? 1 : 0);
                }
                for (; j < nl; ++j) rank += list[j] < mine ? 1 : 0;
                sorted[rank] = mine;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // the row in ascending (d2, index) order, padded with the shadow index (:324); a pair of neighbours with
        // EXACTLY equal d2 whose first member lies inside the kept columns makes the row's reference order
        // depend on the reference's traversal (tieorder.hip): report the row
        bool tie = false;
        for (int e = lane; e < cols; e += 64) {
            long long vout = (long long)ns_out;
            if (e < nl) {
                const u64 mine = sorted[e];
                vout = (long long)((int)(unsigned)(mine & 0xFFFFFFFFull) - base);
                tie |= e + 1 < nl && (unsigned)(sorted[e + 1] >> 32) == (unsigned)(mine >> 32);
            }
            __builtin_nontemporal_store(vout, &row[e]);
        }
        if (tie_rows && __ballot(tie) != 0ull && lane == 0) {
            const int slot = atomicAdd(&s_ntie, 1);
            if (slot < kTieCap) s_tie[slot] = qi;
            else tie_rows[atomicAdd(tie_count, 1)] = qi;          // list full: straight to the global list
        }
        if (lane == 0) {
            if (out_count) out_count[qi] = nhit;
            if (((fill > CAP) || !inrange) && status) *status = 1;   // (fill > CAP: more columns asked for than the list can select)
        }
        wave_max = nhit > wave_max ? nhit : wave_max;
        __builtin_amdgcn_wave_barrier();
    };
    if (!REDO) {
        for (int qi = gw; qi < nq; qi += nw) one_query(qi);
    } else {
        // rare second pass (a row held more hits than the first pass stages): a small grid; each lane looks at one
        // row's marker, the wavefront then redoes the marked rows one by one
        for (int base = gw * 64; base < nq; base += nw * 64) {
            const int r = base + lane;
            u64 todo = __ballot(r < nq && out_idx[(long)r * cols] == kRedoMark);
            while (todo) {
                const int bit = __builtin_ctzll(todo);
                todo &= todo - 1;
                one_query(base + bit);
            }
        }
    }
    flush_max();
    if (tie_rows) {
        __syncthreads();
        const int n = s_ntie < kTieCap ? s_ntie : kTieCap;
        if (threadIdx.x == 0 && n > 0) s_tie_base = atomicAdd(tie_count, n);
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += WAVES * 64) tie_rows[s_tie_base + j] = s_tie[j];
    }
}

__global__ void k_zero2(int* a, int* b) {
    if (a) *a = 0;
    if (b) *b = 0;
}


// ---- ground-truth correspondences (ref:lib/benchmark_utils.py:121-134) ---------------------
// The reference moves the source cloud by a 4x4 transform and asks an open3d KD-tree (float64 points) for the targets
// within `radius` of every source point, nearest first.  Here: one wavefront per source point, candidates from the
// fp32 cell grid of the targets (built with a radius a hair larger, so that nothing inside in float64 is lost), the
// distance of every candidate re-measured in float64 from the float64-moved source point, hits ranked by
// (distance, target index) in LDS and written to a staging row; a second kernel turns the rows into the [K, 2] pair
// list at offsets the caller has scanned.  No sort over the whole table, no index tensors: 527 ms of torch glue for one
// 30 000-point pair became two launches.
constexpr int kCorrCap = 1024;       // staged hits per source point (48 KB of LDS per workgroup)
struct CorrXf { double r[9], t[3]; };

__global__ void __launch_bounds__(256) k_correspond_rows(const float* __restrict__ src, int n, CorrXf xf, double radius,
                                                          GridView g, int keep, int cols, int* __restrict__ stage,
                                                          int* __restrict__ counts, int* __restrict__ max_count) {
    __shared__ double s_d[4][kCorrCap];
    __shared__ int s_i[4][kCorrCap];
    __shared__ int s_excl[4][32];
    __shared__ int s_start[4][32];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = blockIdx.x * 4 + wave;
    if (i >= n) return;
    const double sx = src[3 * (long)i], sy = src[3 * (long)i + 1], sz = src[3 * (long)i + 2];
    const double px = xf.r[0] * sx + xf.r[1] * sy + xf.r[2] * sz + xf.t[0];
    const double py = xf.r[3] * sx + xf.r[4] * sy + xf.r[5] * sz + xf.t[1];
    const double pz = xf.r[6] * sx + xf.r[7] * sy + xf.r[8] * sz + xf.t[2];
    int cx, cy, cz;
    const bool inrange = cell_coords((float)px, (float)py, (float)pz, g.hdr->inv_cell, &cx, &cy, &cz);
    const int ns = g.hdr->ns;
    int ccount = 0, cstart = 0;
    if (lane < 27 && inrange && ns > 0) {
        const int dx = lane % 3 - 1, dy = (lane / 3) % 3 - 1, dz = lane / 9 - 1;
        const u64 key = cell_key(cx + dx, cy + dy, cz + dz);
        const unsigned tsize = 2u * (unsigned)ns;
        unsigned s = __umulhi(mix32(key), tsize);
        for (unsigned probe = 0; probe < tsize; ++probe) {
            const u64 k = g.tkey[s];
            const int kc = g.tcnt[s], ks = g.tstart[s];
            if (k == key) { ccount = kc; cstart = ks; break; }
            if (k == kEmptyKey) break;
            s = s + 1 == tsize ? 0 : s + 1;
        }
    }
    const int incl = wave_incl_scan_i32(ccount, lane);
    const int total = __shfl(incl, 26, 64);
    if (lane < 32) { s_excl[wave][lane] = lane < 27 ? incl - ccount : 0x7FFFFFFF; s_start[wave][lane] = cstart; }
    __builtin_amdgcn_wave_barrier();
    int nhit = 0;
    for (int base = 0; base < total; base += 64) {
        const int t = base + lane;
        const int tc = t < total ? t : total - 1;
        int lo = 0;
#pragma unroll
        for (int step = 16; step >= 1; step >>= 1)
            if (s_excl[wave][lo + step] <= tc) lo += step;
        const float4 p = g.spts[s_start[wave][lo] + (tc - s_excl[wave][lo])];
        const double ex = (double)p.x - px, ey = (double)p.y - py, ez = (double)p.z - pz;
        const double d = sqrt(ex * ex + ey * ey + ez * ez);
        const bool hit = t < total && d < radius;
        const u64 mask = __ballot(hit);
        const int pos = nhit + __popcll(mask & ((1ull << lane) - 1ull));
        if (hit && pos < kCorrCap) { s_d[wave][pos] = d; s_i[wave][pos] = __float_as_int(p.w); }
        nhit += __popcll(mask);
    }
    __builtin_amdgcn_wave_barrier();
    const int nl = nhit < kCorrCap ? nhit : kCorrCap;
    const int kept = keep > 0 && keep < nl ? keep : nl;
    if (lane == 0) {
        counts[i] = nhit > kCorrCap ? nhit : kept;            // > kCorrCap: the caller sees it through max_count and fails
        if (nhit > aload(max_count)) atomicMax(max_count, nhit);
    }
    // rank by (float64 distance, target index); ranks below the cut go to the staging row (only if it is wide enough:
    // the caller re-runs with cols >= max_count otherwise)
    for (int e = lane; e < nl; e += 64) {
        const double md = s_d[wave][e];
        const int mi = s_i[wave][e];
        int rank = 0;
        for (int j = 0; j < nl; ++j) {
            const double od = s_d[wave][j];
            rank += (od < md || (od == md && s_i[wave][j] < mi)) ? 1 : 0;
        }
        if (rank < kept && rank < cols) stage[(long)i * cols + rank] = mi;
    }
}

__global__ void __launch_bounds__(256) k_correspond_emit(const int* __restrict__ stage, int cols, const int* __restrict__ counts,
                                                          const long long* __restrict__ offsets, int n,
                                                          long long* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int c = counts[i];
    const long long o = offsets[i];
    for (int e = lane; e < c; e += 64) {
        out[2 * (o + e)] = i;
        out[2 * (o + e) + 1] = stage[(long)i * cols + e];
    }
}
}  // namespace
}  // namespace pcrcg

using namespace pcrcg;

extern "C" {

size_t pcrcg_cellgrid_ws_bytes(int ns, int nb) { return grid_bytes(ns, nb < 1 ? 1 : nb); }

int pcrcg_cellgrid_build(const float* sup, int ns, const int* slen, int nb, float radius, void* grid,
                         size_t grid_bytes_, void* stream) {
    PCRCG_CHECK_ARG(ns >= 0 && nb >= 1 && slen && grid);
    PCRCG_CHECK_ARG(ns == 0 || sup);
    PCRCG_CHECK_ARG(radius > 0.0f);
    hipStream_t st = as_stream(stream);
    bool ok;
    GridView g = grid_view(grid, grid_bytes_, ns, nb, &ok);
    if (!ok) {
        set_error("pcrcg_cellgrid_build: workspace too small (%zu needed, %zu given)", grid_bytes(ns, nb), grid_bytes_);
        return PCRCG_EWORKSPACE;
    }
    const size_t N = (size_t)ns + 1;
    const double inv_cell = 1.0 / ((double)radius * (1.0 + 1e-5));
    const int init_blocks = (int)((2 * N + 255) / 256 < 1024 ? (2 * N + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_grid_init, dim3(init_blocks), dim3(256), 0, st, g, slen, ns, nb, inv_cell, (long)(2 * N));
    if (ns > 0) {
        const int blocks = (ns + 255) / 256;
        hipLaunchKernelGGL(k_grid_insert, dim3(blocks), dim3(256), 0, st, sup, ns, nb, g);
        hipLaunchKernelGGL(k_grid_starts, dim3((2 * ns + 255) / 256), dim3(256), 0, st, 2 * ns, g);
        hipLaunchKernelGGL(k_grid_scatter, dim3(blocks), dim3(256), 0, st, sup, ns, g);
    }
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

int pcrcg_correspondences_rows(const float* src, int n, const double* trans, double radius, int keep, int m,
                               const void* grid, int cols, int* stage, int* counts, int* max_count, void* stream) {
    PCRCG_CHECK_ARG(n >= 0 && m >= 0 && cols >= 1 && keep >= 0 && radius > 0.0 && trans && grid && counts && max_count);
    if (n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(src && stage);
    bool ok;
    GridView g = grid_view(const_cast<void*>(grid), grid_bytes(m, 1), m, 1, &ok);
    CorrXf xf;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) xf.r[3 * r + c] = trans[4 * r + c];
        xf.t[r] = trans[4 * r + 3];
    }
    hipLaunchKernelGGL(k_correspond_rows, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), src, n, xf, radius, g, keep, cols,
                       stage, counts, max_count);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

int pcrcg_correspondences_emit(const int* stage, int cols, const int* counts, const int64_t* offsets, int n, int64_t* out,
                               void* stream) {
    PCRCG_CHECK_ARG(n >= 0 && cols >= 1);
    if (n == 0) return PCRCG_OK;
    PCRCG_CHECK_ARG(stage && counts && offsets && out);
    hipLaunchKernelGGL(k_correspond_emit, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), stage, cols, counts,
                       reinterpret_cast<const long long*>(offsets), n, reinterpret_cast<long long*>(out));
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

int pcrcg_radius_query(const float* q, int nq, const int* qlen, int ns, const int* slen, int nb,
                       float radius, const void* grid, int cols, int64_t* out_idx, int* out_count,
                       int* out_max_count, int* status, void* stream) {
    return pcrcg_radius_query_ex(q, nq, qlen, ns, slen, nb, radius, grid, cols, out_idx, out_count, out_max_count,
                                 status, nullptr, nullptr, stream);
}

int pcrcg_radius_query_ex(const float* q, int nq, const int* qlen, int ns, const int* slen, int nb,
                          float radius, const void* grid, int cols, int64_t* out_idx, int* out_count,
                          int* out_max_count, int* status, int* out_tie_rows, int* out_tie_count, void* stream) {
    return pcrcg_radius_query_groups(q, nq, qlen, ns, slen, nb, 0, radius, grid, cols, out_idx, out_count, out_max_count,
                                     status, out_tie_rows, out_tie_count, stream);
}

int pcrcg_radius_query_groups(const float* q, int nq, const int* qlen, int ns, const int* slen, int nb, int group,
                              float radius, const void* grid, int cols, int64_t* out_idx, int* out_count,
                              int* out_max_count, int* status, int* out_tie_rows, int* out_tie_count, void* stream) {
    return pcrcg::radius_query_pass(q, nq, qlen, ns, slen, nb, group, radius, grid, cols, out_idx, out_count, out_max_count,
                                    status, out_tie_rows, out_tie_count, as_stream(stream), 0);
}
}

namespace pcrcg {
int radius_fast_cap() { return kListCapFast; }

// pass 0: both kernels (the public entry point).  pass 1: the first kernel only -- rows whose list does not fit its
// 256-entry staging are marked, and out_max_count receives their true length, so a caller that reads out_max_count
// anyway (the pyramid builder) launches pass 2 only for tables that need it: normally none.  pass 2: the redo kernel.
int radius_query_pass(const float* q, int nq, const int* qlen, int ns, const int* slen, int nb, int group, float radius,
                      const void* grid, int cols, int64_t* out_idx, int* out_count, int* out_max_count, int* status,
                      int* out_tie_rows, int* out_tie_count, hipStream_t st, int pass) {
    PCRCG_CHECK_ARG(nq >= 0 && ns >= 0 && nb >= 1 && cols >= 1 && group >= 0);
    PCRCG_CHECK_ARG((out_tie_rows == nullptr) == (out_tie_count == nullptr));
    PCRCG_CHECK_ARG(qlen && slen && grid && out_idx && out_max_count);
    PCRCG_CHECK_ARG(nq == 0 || q);
    (void)slen;
    if (nq == 0) return PCRCG_OK;
    bool ok;
    GridView g = grid_view(const_cast<void*>(grid), grid_bytes(ns, nb), ns, nb, &ok);
    const float r2 = radius * radius;  // neighbors.cpp:226
    int blocks = (nq + kQueryWaves - 1) / kQueryWaves;
    const int max_blocks_env = debug_opts().radius_blocks;
    // 2 workgroups (8 wavefronts) per CU, wavefronts loop over the queries: inside the pipeline a smaller grid takes less
    // from the model streams (1024 workgroups: 455 pairs/s, 512: 464, 384: 453, 256: 398), and on voxelised data every
    // workgroup appends its tie rows with one atomic on one word (4096 workgroups: 264 us per 60k-row table)
    const int max_blocks = max_blocks_env > 0 ? max_blocks_env : 256 * 2;
    if (blocks > max_blocks) blocks = max_blocks;
    if (pass != 2)
        hipLaunchKernelGGL((k_radius_query<kListCapFast, false, kQueryWaves>), dim3(blocks), dim3(kQueryWaves * 64), 0, st, q, nq,
                           qlen, nb, r2, g, cols, reinterpret_cast<long long*>(out_idx), out_count, out_max_count, status,
                           out_tie_rows, out_tie_count, group);
    // second pass: one wavefront per workgroup, 16 KB of LDS -- it finds a free slot at once on a busy GPU and
    // normally has nothing to do
    const int redo_blocks = blocks < 64 ? blocks : 64;
    if (pass != 1)
        hipLaunchKernelGGL((k_radius_query<kListCapFull, true, 1>), dim3(redo_blocks), dim3(64), 0, st, q, nq, qlen, nb, r2, g,
                           cols, reinterpret_cast<long long*>(out_idx), out_count, out_max_count, status, out_tie_rows,
                           out_tie_count, group);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}
}  // namespace pcrcg

extern "C" {

size_t pcrcg_radius_neighbors_ws_bytes(int ns, int nb) { return pcrcg_cellgrid_ws_bytes(ns, nb); }

int pcrcg_radius_neighbors_batch(const float* q, int nq, const float* sup, int ns, const int* qlen,
                                 const int* slen, int nb, float radius, int cols, int64_t* out_idx,
                                 int* out_count, int* out_max_count, int* status, void* ws,
                                 size_t ws_bytes, void* stream) {
    PCRCG_CHECK_ARG(out_max_count != nullptr);
    hipLaunchKernelGGL(k_zero2, dim3(1), dim3(1), 0, as_stream(stream), out_max_count, status);
    PCRCG_PROPAGATE(pcrcg_cellgrid_build(sup, ns, slen, nb, radius, ws, ws_bytes, stream));
    return pcrcg_radius_query(q, nq, qlen, ns, slen, nb, radius, ws, cols, out_idx, out_count, out_max_count,
                              status, stream);
}
}
