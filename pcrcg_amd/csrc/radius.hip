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
//   query : CELL-COOPERATIVE (k_radius_cells, round 4): the queries are taken cell by cell from a grid of their own
//           (in the pyramid every query set already has one); one workgroup per occupied query cell resolves the
//           hash probes of the support cells within reach ONCE (27 for a conv / upsample table, 64 for a pool table),
//           stages their candidate runs and the cell's queries in LDS, and every wavefront then runs queries of the
//           cell against LDS: hits (d2 < r2, the reference's exact fp32 arithmetic) are compacted with
//           ballot/popcount, a rank sort over the (d2, index) keys writes the row in ascending order, truncated to
//           `cols` and padded with ns (:319-325).  ~14 queries share a cell at S30k: the probe -> slot -> run ->
//           gather chain of dependent global loads is paid once per cell instead of once per query.
//           PER-QUERY (k_radius_query, rounds 1-3): one wavefront per query probes its own 27 cells and gathers its
//           candidates from global memory; it serves query sets that have no grid, and as the second pass
//           (REDO) the rows the cell kernel hands over (more than kCellListCap = 128 hits, or a neighbourhood that does not fit LDS).
//
// Squared distances follow nanoflann's L2_Simple_Adaptor (zip:cpp_utils/nanoflann/nanoflann.hpp:
// 432-440): ((0 + dx*dx) + dy*dy) + dz*dz with every product and sum rounded to fp32, strict
// d2 < r*r (:249-253, neighbors.cpp:226).  Compiled with -ffp-contract=off.
#include <hip/hip_ext.h>

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

constexpr int kRedoStatus = 4;        // status bit: the cell kernel left rows to the per-query second pass
constexpr int kCellCand = 1024;       // candidates staged per query cell (16 KiB; a level subsampled at dl = r / 2.5 holds
                                      // at most 15.6 points per cell: 27 cells 421, the 64 cells of a pool table 1000 in a
                                      // solid volume, ~225 on scanned surfaces)
constexpr int kCellListCap = 128;     // hits staged per query by the cell kernel (longer rows: second pass)
constexpr unsigned kPadKey = 0x7FFFFFFFu;   // padding "d2" of the cell kernel's hit lists: above every real d2 word, INT_MAX as an int
constexpr int kCellTieCap = 512;      // tie rows staged per workgroup of the cell kernel (beyond: one global atomic per row -- 1 ms per
                                      // 60k-row table on voxelised data; two stacked T30k pairs on 512 workgroups give ~235 per workgroup)
constexpr int kCellMaxCells = 256;    // support cells within reach of one query cell
constexpr int kCellQ = 64;            // queries of a cell staged per batch

struct GridHeader {   // first 256 bytes of the grid workspace
    double inv_cell;  // 1 / (radius * (1 + 1e-5)): cells are a hair wider than the radius
    int ns, nb;
    u64 cursor;       // low word: bump allocator for cell runs; high word: occupied cells listed so far (ONE atomic)
    int overflow;     // coordinate range exceeded
};
constexpr int kTickStride = 64;       // ints between two ticket words: every counter in a 256-byte block of its own (eight
                                      // counters in ONE cache line were served one after the other, ~50 atomics per microsecond
                                      // for the whole chip: 84 of the 94 us of the 60 000-row search)

struct Slot {         // one 16-byte record per hash slot: a probe is ONE load
    u64 key;
    int cnt, start;
};

struct GridView {
    GridHeader* hdr;
    int* soff;     // [nb+1]
    Slot* tab;     // [2*ns + 2]
    int* slot_of;  // [ns]
    int* pos_in;   // [ns]
    float4* spts;  // [ns]  supports cell by cell as (x, y, z, index)
    u64* ckey;     // [ns]  occupied cells, compact (any order): key ...
    int4* cinfo;   // [ns]  ... and (count, start of the run in spts, cloud, slot)
    int* qtick;    // [16 * kTickStride]  k_radius_cells walking THIS grid as its query grid: ticket counter of shard k at
                   // [k * kTickStride], workgroups of shard k that have left at [(8 + k) * kTickStride]
};

inline size_t grid_bytes(int ns, int nb) {
    const size_t N = (size_t)(ns > 0 ? ns : 0) + 1;
    return carve_bytes(1, 256) + carve_bytes((size_t)nb + 1, sizeof(int)) + carve_bytes(2 * N, sizeof(Slot)) +
           2 * carve_bytes(N, sizeof(int)) + carve_bytes(N, sizeof(float4)) + carve_bytes(N, sizeof(u64)) +
           carve_bytes(N, sizeof(int4)) + carve_bytes(16 * kTickStride, sizeof(int));
}

inline GridView grid_view(void* ws, size_t bytes, int ns, int nb, bool* ok) {
    const size_t N = (size_t)(ns > 0 ? ns : 0) + 1;
    Carver cv(ws, bytes);
    GridView g;
    g.hdr = reinterpret_cast<GridHeader*>(cv.take<char>(256));
    g.soff = cv.take<int>((size_t)nb + 1);
    g.tab = cv.take<Slot>(2 * N);
    g.slot_of = cv.take<int>(N);
    g.pos_in = cv.take<int>(N);
    g.spts = cv.take<float4>(N);
    g.ckey = cv.take<u64>(N);
    g.cinfo = cv.take<int4>(N);
    g.qtick = cv.take<int>(16 * kTickStride);
    *ok = cv.ok();
    return g;
}

__device__ __forceinline__ Slot load_slot(const Slot* p) {     // one global_load_dwordx4
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    Slot s;
    s.key = (u64)v.x | ((u64)v.y << 32);
    s.cnt = (int)v.z;
    s.start = (int)v.w;
    return s;
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
        g.hdr->ns = s;          // (the kernel argument is the caller's bound; the lengths say how many supports there are)
        g.hdr->nb = nb;
        g.hdr->cursor = 0ull;
        g.hdr->overflow = 0;
        for (int k = 0; k < 16; ++k) g.qtick[k * kTickStride] = 0;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (long)gridDim.x * blockDim.x)
        *reinterpret_cast<uint4*>(&g.tab[i]) = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
}

__global__ void __launch_bounds__(256) k_grid_insert(const float* __restrict__ sup, int nb, GridView g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.soff[nb]) return;
    const int b = cloud_of(g.soff, nb, i);
    int cx, cy, cz;
    if (!cell_coords(sup[3 * (long)i], sup[3 * (long)i + 1], sup[3 * (long)i + 2], g.hdr->inv_cell, &cx, &cy, &cz))
        g.hdr->overflow = 1;
    const u64 key = cell_key(cx, cy, cz);
    const unsigned tsize = 2u * (unsigned)(g.soff[b + 1] - g.soff[b]);
    const long tbase = 2l * g.soff[b];
    unsigned s = __umulhi(mix32(key), tsize);
    for (;;) {
        u64 prev = atomicCAS(&g.tab[tbase + s].key, kEmptyKey, key);
        if (prev == kEmptyKey || prev == key) break;
        s = s + 1 == tsize ? 0 : s + 1;
    }
    const int slot = (int)(tbase + s);
    g.slot_of[i] = slot;
    g.pos_in[i] = atomicAdd(&g.tab[slot].cnt, 1);
}

// run start of every occupied slot + the compact list of occupied cells (what the cell-cooperative search walks): one
// 64-bit atomic hands out both the run's offset (low word) and the cell's list position (high word)
__global__ void __launch_bounds__(256) k_grid_starts(int nslots, int nb, GridView g) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    const int c = g.tab[s].cnt;
    if (c > 0) {
        const u64 got = atomicAdd(&g.hdr->cursor, (1ull << 32) | (u64)(unsigned)c);
        const int start = (int)(unsigned)got, j = (int)(got >> 32);
        g.tab[s].start = start;
        g.ckey[j] = g.tab[s].key;
        g.cinfo[j] = make_int4(c, start, cloud_of(g.soff, nb, s >> 1), s);
    }
}

__global__ void __launch_bounds__(256) k_grid_scatter(const float* __restrict__ sup, int nb, GridView g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.soff[nb]) return;
    const int dst = g.tab[g.slot_of[i]].start + g.pos_in[i];
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
    if (REDO && blockIdx.x == 0 && threadIdx.x == 0 && status) {
        atomicAnd(status, ~kRedoStatus);        // this pass takes every row the cell kernel handed over
        if (g.hdr->overflow) atomicOr(status, 2);
    }
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
                const Slot sl = load_slot(&g.tab[tbase + s]);
                if (sl.key == key) { ccount = sl.cnt; cstart = sl.start; break; }
                if (sl.key == kEmptyKey) break;
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
            if (((fill > CAP) || !inrange) && status) atomicOr(status, 1);   // (fill > CAP: more columns asked for than the list can select)
        }
        wave_max = nhit > wave_max ? nhit : wave_max;
        __builtin_amdgcn_wave_barrier();
    };
    // nq is the caller's BOUND on the rows (what its launch and its buffers were sized for); the query lengths, on the
    // device, say how many rows there are
    {
        int have = 0;
        for (int b = 0; b < nb; ++b) have += qlen[b];
        nq = nq < have ? nq : have;
    }
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


// ---- cell-cooperative search --------------------------------------------------------------------------------------
// One workgroup per occupied cell of the QUERY grid gq (any cell size; in the pyramid the query set's own conv grid).
// Every support within the radius of a query of that cell lies in the cells of the support grid gs that the query cell,
// grown by `reach` = radius * (1 + 1e-6), overlaps (fp32 rounding lets d2 < r*r pass for a true distance of at most
// radius * (1 + 2e-7)): 3x3x3 cells when both grids have the search radius as cell edge (conv), 4x4x4 when the query
// cells are twice as large (pool), 3x3x3 when they are half as large (upsample).
//   wave 0      : the cell range, one hash probe per support cell (a lane each), wave prefix sum of the run lengths;
//                 then the NEXT cell's record and ticket are requested (they arrive during the query phase)
//   waves 1..   : meanwhile the cell's queries -> LDS
//   all threads : the candidate runs -> LDS (float4 records; P threads side by side take the P cells, rows of threads walk
//                 the runs: no search for "which cell does candidate t belong to")
//   every wave  : queries of the cell one after the other, entirely out of LDS / registers: sweep (64 candidates per
//                 step, hits compacted with ballot / popcount), rank sort, one coalesced row store
// The kernel is bound by instruction issue (6 wavefronts per SIMD, ~40 M wavefront instructions for the 60 000-row
// table in its first form): everything wave-uniform is computed by wave 0 only and handed over through LDS, and the
// three barriers of a cell wait for LDS only (s_waitcnt lgkmcnt(0) + s_barrier: __syncthreads() also drains every
// outstanding global load and store, which serialised the prefetch behind the probes).
// Rank sort: every lane holds one hit (two from 65 hits on) and counts the hits with a smaller d2 while the d2 words
// are broadcast from LDS, four per ds_read_b128 -- 32-bit compares only; ranks of hits with EXACTLY equal d2 collide,
// which the scatter itself detects (a lane reads back somebody else's index); only then the (d2, index) order is
// computed with both words.  (The first version broadcast 64-bit keys through SGPRs, v_readlane + v_cmp_lt_u64 + s_nop
// per hit: 60 of its 105 us on the 60 000-row table.)
// Cells are handed out dynamically (cells differ 1..40 queries: a static split leaves the slowest workgroup with twice
// the mean): a workgroup's first two cells are fixed by its index, further ones come from a ticket counter in the query
// grid's header -- one counter per shard (workgroup index mod 8 = its XCD; cell c belongs to shard c mod 8), because a
// single word serves ~88 atomics per microsecond and 1536 workgroups asking at once cost more than the search.  The
// last workgroup of a shard to leave resets its counters for the next search over this grid.
// A query cell whose neighbourhood does not fit (more than SC candidates or kCellMaxCells cells) and rows with more
// than CAP hits are marked for the per-query second pass (k_radius_query<.., REDO = true>) and announced through
// status bit kRedoStatus (rows of more than CAP hits also through out_max).
struct CellArgs {
    const GridHeader* qhdr;       // query grid: header, ticket counters, cell runs, compact cell list
    int* qtick;
    const float4* qspts;
    const u64* ckey;
    const int4* cinfo;
    const GridHeader* shdr;       // support grid: header, cloud offsets, hash table, cell runs
    const int* soff;
    const Slot* tab;
    const float4* spts;
    long long* out_idx;
    int* out_count;
    int* out_max;
    int* status;
    int* tie_rows;
    int* tie_count;
    double reach;
    float r2;
    int nb, cols, group;
    long long* prof;              // measurement aid (PROF kernels only)
};

struct CellState {                // what wave 0 hands to the other waves of the workgroup, per cell
    int more, nqc, qstart, base, ns_out, grp;
    int total, log2p, fits;
};

// workgroup barrier that waits for this wave's LDS operations only (global loads / stores stay in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int SC, int CAP, int WAVES, bool PROF = false>
__global__ void __launch_bounds__(WAVES * 64) k_radius_cells(const CellArgs a) {
    __shared__ float4 s_cand[SC];                    // (SC is a multiple of 64: the padded tail of the last step fits)
    __shared__ float4 s_q[kCellQ];
    __shared__ unsigned s_key[WAVES][CAP + 8];       // d2 bits of a query's hits ...
    __shared__ int s_idx[WAVES][CAP + 8];            // ... and their support indices
    __shared__ int s_excl[kCellMaxCells];
    __shared__ int s_start[kCellMaxCells];
    __shared__ int s_cnt[kCellMaxCells];
    __shared__ int s_tie[kCellTieCap];
    __shared__ CellState s_state;
    __shared__ int s_ntie, s_tie_base;
    // PROF (measurement aid, DebugOpts::radius_prof): shader-clock cycles per phase, summed over wavefronts
    int pcells = 0, pqueries = 0;
    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long pt_last = PROF ? (long long)__builtin_readcyclecounter() : 0;
    auto stamp = [&](int phase) {
        if (PROF) {
            const long long now = (long long)__builtin_readcyclecounter();
            pt[phase] += now - pt_last;
            pt_last = now;
        }
    };
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cols = a.cols;
    const float r2 = a.r2;
    unsigned* key = s_key[wave];
    int* idx = s_idx[wave];
    constexpr int kTrash = CAP + 7;                  // last slot of both lists: where lanes without a hit write
    int wave_max = 0, wave_grp = 0;
    auto flush_max = [&]() {
        if (lane == 0 && wave_max > aload(a.out_max + wave_grp)) atomicMax(a.out_max + wave_grp, wave_max);
    };
    if (threadIdx.x == 0) s_ntie = 0;

    // ---- wave 0's bookkeeping: c = the cell in hand, c1 = the next one, its record `rec1` (lanes 0..5) on its way ----
    // Shard k owns the cells c = 8 j + k; its Gk workgroups start with j = w and j = w + Gk, tickets continue from 2 Gk.
    const int nshard = gridDim.x < 8 ? (int)gridDim.x : 8;
    const int shard = blockIdx.x % nshard, Gk = ((int)gridDim.x - shard + nshard - 1) / nshard;
    int ncells_q = 0, ns = 0, c = 0, c1 = 0, ticket = 0;
    double inv_s = 0.0, cell_q = 0.0;
    unsigned rec = 0, rec1 = 0;                      // lanes 0,1: the cell's key; lanes 2..5: (count, start, cloud, slot)
    int soff_reg = 0;                                // wave 0, lane i: soff[i] -- the clouds of a launch are few: no load (and
                                                     // no dependent round trip in front of the probes) per cell
    auto soff_at = [&](int i) -> int { return a.nb < 63 ? __builtin_amdgcn_readlane(soff_reg, i) : a.soff[i]; };
    auto load_rec = [&](int cell) -> unsigned {      // one dword per lane, a VECTOR load: it stays in flight across LDS waits
        unsigned v = 0;
        if (cell < ncells_q && lane < 6)
            v = lane < 2 ? reinterpret_cast<const unsigned*>(a.ckey + cell)[lane]
                         : reinterpret_cast<const unsigned*>(a.cinfo + cell)[lane - 2];
        return v;
    };
    if (wave == 0) {
        ncells_q = (int)(a.qhdr->cursor >> 32);
        ns = a.shdr->ns;
        inv_s = a.shdr->inv_cell;
        cell_q = 1.0 / a.qhdr->inv_cell;
        if (a.nb < 63) soff_reg = lane <= a.nb ? a.soff[lane] : 0;
        c = blockIdx.x;
        c1 = c + Gk * nshard;
        rec = load_rec(c);
        if (threadIdx.x == 0 && a.status && (a.qhdr->overflow || a.shdr->overflow)) atomicOr(a.status, 2);
    }

    for (;;) {
        int nqc = 0, qstart = 0, b = 0;
        if (wave == 0) {
            const bool more = c < ncells_q;
            nqc = __builtin_amdgcn_readlane((int)rec, 2);
            qstart = __builtin_amdgcn_readlane((int)rec, 3);
            b = more ? __builtin_amdgcn_readlane((int)rec, 4) : 0;
            int base = 0, ns_out = ns, grp = 0;
            if (a.group > 0 && more) {
                grp = b / a.group;
                const int last = min(grp * a.group + a.group, a.nb);
                base = soff_at(grp * a.group);
                ns_out = soff_at(last) - base;
            }
            if (lane == 0) {
                s_state.more = more ? 1 : 0;
                s_state.nqc = nqc; s_state.qstart = qstart; s_state.base = base; s_state.ns_out = ns_out; s_state.grp = grp;
            }
        }
        lds_barrier();                               // A: the cell's state is out; the previous cell's readers are done
        stamp(0);
        if (!s_state.more) break;
        nqc = s_state.nqc;
        qstart = s_state.qstart;
        const int base = s_state.base, ns_out = s_state.ns_out;
        if (s_state.grp != wave_grp) { flush_max(); wave_grp = s_state.grp; wave_max = 0; }

        if (wave == 0) {
            // support cells the grown query cell overlaps (unbiased cell coordinates; conservative by construction)
            const u64 qkey = (u64)(unsigned)__builtin_amdgcn_readlane((int)rec, 0) |
                             ((u64)(unsigned)__builtin_amdgcn_readlane((int)rec, 1) << 32);
            const int nsb = soff_at(b + 1) - soff_at(b);
            const long tbase = 2l * soff_at(b);
            const int ux = (int)(qkey & 0x1FFFFF) - kCoordBias, uy = (int)((qkey >> 21) & 0x1FFFFF) - kCoordBias,
                      uz = (int)((qkey >> 42) & 0x1FFFFF) - kCoordBias;
            const int lox = (int)floor(((double)ux * cell_q - a.reach) * inv_s), hix = (int)floor(((double)(ux + 1) * cell_q + a.reach) * inv_s);
            const int loy = (int)floor(((double)uy * cell_q - a.reach) * inv_s), hiy = (int)floor(((double)(uy + 1) * cell_q + a.reach) * inv_s);
            const int loz = (int)floor(((double)uz * cell_q - a.reach) * inv_s), hiz = (int)floor(((double)(uz + 1) * cell_q + a.reach) * inv_s);
            const int nx = hix - lox + 1, ny = hiy - loy + 1, nz = hiz - loz + 1;
            const long ncell_l = (long)nx * ny * nz;
            const bool cells_fit = ncell_l <= kCellMaxCells;
            const int ncell = cells_fit ? (int)ncell_l : 0;
            int log2p = 5;                           // the cell table is padded to P = 2^log2p entries
            while ((1 << log2p) < ncell) ++log2p;
            const int P = 1 << log2p;
            int carry = 0;
            for (int c0 = 0; c0 < P; c0 += 64) {
                const int t = c0 + lane;
                int ccount = 0, cstart = 0;
                if (t < ncell && nsb > 0) {
                    const int sx = lox + t % nx + kCoordBias, sy = loy + (t / nx) % ny + kCoordBias,
                              sz = loz + t / (nx * ny) + kCoordBias;
                    if ((unsigned)sx < (1u << 21) && (unsigned)sy < (1u << 21) && (unsigned)sz < (1u << 21)) {
                        const u64 ckey = cell_key(sx, sy, sz);
                        const unsigned tsize = 2u * (unsigned)nsb;
                        unsigned s = __umulhi(mix32(ckey), tsize);
                        for (unsigned probe = 0; probe < tsize; ++probe) {
                            const Slot sl = load_slot(&a.tab[tbase + s]);
                            if (sl.key == ckey) { ccount = sl.cnt; cstart = sl.start; break; }
                            if (sl.key == kEmptyKey) break;
                            s = s + 1 == tsize ? 0 : s + 1;
                        }
                    }
                }
                const int incl = wave_incl_scan_i32(ccount, lane);
                if (t < P) { s_excl[t] = carry + incl - ccount; s_start[t] = cstart; s_cnt[t] = ccount; }
                carry += __shfl(incl, 63, 64);
            }
            if (lane == 0) { s_state.total = carry; s_state.log2p = log2p; s_state.fits = cells_fit && carry <= SC ? 1 : 0; }
            // the probes are in: now ask for the next cell's record and for the ticket after it
            rec1 = load_rec(c1);
            ticket = 0x40000000;                     // (no cell behind c1: tickets grow monotonically)
            if (lane == 0 && c1 < ncells_q) ticket = (2 * Gk + atomicAdd(&a.qtick[shard * kTickStride], 1)) * nshard + shard;
        } else {                                     // the first batch of the cell's queries, beside the probes
            const int t = (int)threadIdx.x - 64;
            if (t < kCellQ && t < nqc) s_q[t] = a.qspts[qstart + t];
        }
        if (WAVES == 1 && lane < kCellQ && lane < nqc) s_q[lane] = a.qspts[qstart + lane];
        lds_barrier();                               // B
        stamp(1);
        const int total = s_state.total, log2p = s_state.log2p;
        const bool fits = s_state.fits != 0;
        if (fits) {
            const int cell = threadIdx.x & ((1 << log2p) - 1), stride = (WAVES * 64) >> log2p;
            const int cnt = s_cnt[cell], src = s_start[cell], dst = s_excl[cell];
            for (int j = threadIdx.x >> log2p; j < cnt; j += stride) s_cand[dst + j] = a.spts[src + j];
            if ((int)threadIdx.x < 64 && total + (int)threadIdx.x < ((total + 63) & ~63))      // far-away tail: never a hit
                s_cand[total + threadIdx.x] = make_float4(3.0e18f, 3.0e18f, 3.0e18f, 0.0f);
        }
        stamp(2);
        for (int q0 = 0; q0 < nqc; q0 += kCellQ) {
            if (q0 > 0) {
                lds_barrier();
                if ((int)threadIdx.x < kCellQ && q0 + (int)threadIdx.x < nqc) s_q[threadIdx.x] = a.qspts[qstart + q0 + threadIdx.x];
            }
            lds_barrier();                           // C
            stamp(3);
            const int nbatch = min(kCellQ, nqc - q0);
            for (int k = wave; k < nbatch; k += WAVES) {
                const float4 qr = s_q[k];
                const float qx = qr.x, qy = qr.y, qz = qr.z;
                const int qi = __builtin_amdgcn_readfirstlane(__float_as_int(qr.w));
                long long* row = a.out_idx + (long)qi * cols;
                if (!fits) {                         // neighbourhood too large for LDS: the per-query pass takes the row
                    if (lane == 0) { row[0] = kRedoMark; if (a.status) atomicOr(a.status, kRedoStatus); }
                    continue;
                }
                // sweep: 64 staged candidates per step (the tail of the last step is padded with far-away points: no bounds
                // test); hits are compacted with ballot / mbcnt.  The kernel is bound by its SCALAR instructions (round 4:
                // SQ_ACTIVE_INST_SCA + _MISC = 85 % of the CU cycles), so per-lane conditions are VALU selects with
                // unconditional LDS accesses -- a non-hit writes into a trash slot -- instead of exec-mask branches.
                int nhit = 0;
                for (int t0 = 0; t0 < total; t0 += 64) {
                    const float4 p = s_cand[t0 + lane];
                    const float d0 = qx - p.x, d1 = qy - p.y, d2c = qz - p.z;
                    float d2 = 0.0f;
                    d2 += d0 * d0;
                    d2 += d1 * d1;
                    d2 += d2c * d2c;
                    const bool hit = d2 < r2;
                    const u64 mask = __builtin_amdgcn_ballot_w64(hit);
                    int pos = nhit + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    pos = hit ? pos : kTrash;
                    pos = pos < kTrash ? pos : kTrash;           // (more than CAP hits: the row goes to pass 2 anyway)
                    key[pos] = __float_as_uint(d2);
                    idx[pos] = __float_as_int(p.w);
                    nhit += __builtin_popcountll(mask);
                }
                stamp(4);
                if (PROF) ++pqueries;
                wave_max = nhit > wave_max ? nhit : wave_max;
                if (nhit > CAP) {                    // leave the row to pass 2
                    if (lane == 0) { row[0] = kRedoMark; if (a.status) atomicOr(a.status, kRedoStatus); }
                    continue;
                }
                const int nl = nhit;
                key[nl + (lane & 7)] = kPadKey;                 // the broadcast loop reads eight keys per step (all lanes
                __builtin_amdgcn_wave_barrier();                // write: eight addresses, one value -- no exec mask)
                if (nl <= 64 && cols <= 64) {
                    // ---- the common row, scalar-lean: one hit per lane, rank = number of hits with a smaller d2 ----
                    // (d2 words are positive floats, i.e. below 2^31, and the padding is INT_MAX: "k < mine" is the sign
                    // of the signed difference -- pure VGPR arithmetic, no v_cmp -> VCC -> v_addc chain and its wait states)
                    const bool have = lane < nl;
                    const int mkey = have ? (int)key[lane] : (int)kPadKey;
                    const int midx = idx[lane];
                    int acc = 0;
#pragma unroll
                    for (int j = 0; j < 64; j += 8) {
                        if (j >= nl) break;                     // wave-uniform
                        const int4 k0 = *reinterpret_cast<const int4*>(&key[j]);
                        const int4 k1 = *reinterpret_cast<const int4*>(&key[j + 4]);
                        acc += ((k0.x - mkey) >> 31) + ((k0.y - mkey) >> 31);
                        acc += ((k0.z - mkey) >> 31) + ((k0.w - mkey) >> 31);
                        acc += ((k1.x - mkey) >> 31) + ((k1.y - mkey) >> 31);
                        acc += ((k1.z - mkey) >> 31) + ((k1.w - mkey) >> 31);
                    }
                    int dst = have ? -acc : kTrash;
                    __builtin_amdgcn_wave_barrier();
                    key[dst] = (unsigned)mkey;
                    idx[dst] = midx;
                    __builtin_amdgcn_wave_barrier();
                    const bool collide = have && idx[dst] != midx;
                    if (__builtin_amdgcn_ballot_w64(collide) != 0ull) {
                        // exactly equal distances in this row: order by (d2, index) with both words
                        __builtin_amdgcn_wave_barrier();
                        const int home = have ? lane : kTrash;
                        key[home] = (unsigned)mkey;
                        idx[home] = midx;
                        key[nl + (lane & 7)] = kPadKey;
                        __builtin_amdgcn_wave_barrier();
                        int rk = 0;
                        for (int j = 0; j < nl; j += 4) {
                            const int4 k0 = *reinterpret_cast<const int4*>(&key[j]);
                            const int4 i0 = *reinterpret_cast<const int4*>(&idx[j]);
                            rk += ((k0.x < mkey || (k0.x == mkey && i0.x < midx)) ? 1 : 0) + ((k0.y < mkey || (k0.y == mkey && i0.y < midx)) ? 1 : 0) +
                                  ((k0.z < mkey || (k0.z == mkey && i0.z < midx)) ? 1 : 0) + ((k0.w < mkey || (k0.w == mkey && i0.w < midx)) ? 1 : 0);
                        }
                        dst = have ? rk : kTrash;
                        __builtin_amdgcn_wave_barrier();
                        key[dst] = (unsigned)mkey;
                        idx[dst] = midx;
                    }
                    __builtin_amdgcn_wave_barrier();
                    stamp(5);
                    // the row: lane e writes column e
                    const int e = lane < CAP ? lane : 0;
                    const unsigned ke = key[e], kn = key[e + 1];
                    const int ie = idx[e];
                    const bool valid = lane < nl;
                    const long long vout = valid ? (long long)(ie - base) : (long long)ns_out;
                    const bool tie = valid && lane < cols && lane + 1 < nl && kn == ke;
                    if (lane < cols) __builtin_nontemporal_store(vout, &row[lane]);
                    if (__builtin_amdgcn_ballot_w64(tie) != 0ull && a.tie_rows) {
                        if (lane == 0) {
                            const int slot = atomicAdd(&s_ntie, 1);
                            if (slot < kCellTieCap) s_tie[slot] = qi;
                            else a.tie_rows[atomicAdd(a.tie_count, 1)] = qi;
                        }
                    }
                    if (a.out_count) a.out_count[qi] = nhit;    // (every lane, one address, one value)
                    __builtin_amdgcn_wave_barrier();
                    stamp(6);
                    continue;
                }
                // ---- rank sort (rows of 65..CAP hits, tables of more than 64 columns); the sorted hits go back in place ----
                constexpr int R = CAP / 64;                     // hits per lane
                unsigned mk[R];
                int mi[R], rank[R];
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    const bool have = lane + 64 * u < nl;
                    mk[u] = have ? key[lane + 64 * u] : kPadKey;
                    mi[u] = have ? idx[lane + 64 * u] : 0x7FFFFFFF;
                    rank[u] = 0;
                }
                if (nl <= 64) {                                 // (nearly always) one hit per lane
                    int rk = 0;
                    for (int j = 0; j < nl; j += 8) {
                        const uint4 k0 = *reinterpret_cast<const uint4*>(&key[j]);
                        const uint4 k1 = *reinterpret_cast<const uint4*>(&key[j + 4]);
                        rk += (k0.x < mk[0] ? 1 : 0) + (k0.y < mk[0] ? 1 : 0) + (k0.z < mk[0] ? 1 : 0) + (k0.w < mk[0] ? 1 : 0) +
                              (k1.x < mk[0] ? 1 : 0) + (k1.y < mk[0] ? 1 : 0) + (k1.z < mk[0] ? 1 : 0) + (k1.w < mk[0] ? 1 : 0);
                    }
                    rank[0] = rk;
                } else {
                    for (int j = 0; j < nl; j += 4) {
                        const uint4 k0 = *reinterpret_cast<const uint4*>(&key[j]);
#pragma unroll
                        for (int u = 0; u < R; ++u)
                            rank[u] += (k0.x < mk[u] ? 1 : 0) + (k0.y < mk[u] ? 1 : 0) + (k0.z < mk[u] ? 1 : 0) + (k0.w < mk[u] ? 1 : 0);
                    }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < R; ++u)
                    if (lane + 64 * u < nl) { key[rank[u]] = mk[u]; idx[rank[u]] = mi[u]; }
                __builtin_amdgcn_wave_barrier();
                bool collide = false;
#pragma unroll
                for (int u = 0; u < R; ++u)
                    if (lane + 64 * u < nl) collide |= idx[rank[u]] != mi[u];
                if (__ballot(collide) != 0ull) {
                    // exactly equal distances in this row: order by (d2, index) with both words
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        if (lane + 64 * u < nl) { key[lane + 64 * u] = mk[u]; idx[lane + 64 * u] = mi[u]; }
                        rank[u] = 0;
                    }
                    __builtin_amdgcn_wave_barrier();
                    for (int j = 0; j < nl; j += 4) {            // four hits per step (voxelised scans: most rows come here)
                        const uint4 k0 = *reinterpret_cast<const uint4*>(&key[j]);
                        const int4 i0 = *reinterpret_cast<const int4*>(&idx[j]);
#pragma unroll
                        for (int u = 0; u < R; ++u)
                            rank[u] += ((k0.x < mk[u] || (k0.x == mk[u] && i0.x < mi[u])) ? 1 : 0) +
                                       ((k0.y < mk[u] || (k0.y == mk[u] && i0.y < mi[u])) ? 1 : 0) +
                                       ((k0.z < mk[u] || (k0.z == mk[u] && i0.z < mi[u])) ? 1 : 0) +
                                       ((k0.w < mk[u] || (k0.w == mk[u] && i0.w < mi[u])) ? 1 : 0);
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int u = 0; u < R; ++u)
                        if (lane + 64 * u < nl) { key[rank[u]] = mk[u]; idx[rank[u]] = mi[u]; }
                }
                __builtin_amdgcn_wave_barrier();
                stamp(5);
                bool tie = false;
                for (int e = lane; e < cols; e += 64) {
                    long long vout = (long long)ns_out;
                    if (e < nl) {
                        vout = (long long)(idx[e] - base);
                        tie |= e + 1 < nl && key[e + 1] == key[e];
                    }
                    __builtin_nontemporal_store(vout, &row[e]);
                }
                if (a.tie_rows && __ballot(tie) != 0ull && lane == 0) {
                    const int slot = atomicAdd(&s_ntie, 1);
                    if (slot < kCellTieCap) s_tie[slot] = qi;
                    else a.tie_rows[atomicAdd(a.tie_count, 1)] = qi;
                }
                if (lane == 0 && a.out_count) a.out_count[qi] = nhit;
                __builtin_amdgcn_wave_barrier();
                stamp(6);
            }
        }
        if (wave == 0) {                             // next cell: its record and the ticket behind it have arrived meanwhile
            c = c1;
            rec = rec1;
            c1 = __builtin_amdgcn_readfirstlane(ticket);
        }
        stamp(7);
        if (PROF) ++pcells;
    }
    if (PROF && lane == 0 && a.prof) {
        for (int k = 0; k < 8; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(a.prof) + k, (unsigned long long)pt[k]);
        atomicAdd(reinterpret_cast<unsigned long long*>(a.prof) + 8, (unsigned long long)pcells);
        atomicAdd(reinterpret_cast<unsigned long long*>(a.prof) + 9, (unsigned long long)pqueries);
    }
    flush_max();
    __syncthreads();
    if (a.tie_rows) {
        const int n = s_ntie < kCellTieCap ? s_ntie : kCellTieCap;
        if (threadIdx.x == 0 && n > 0) s_tie_base = atomicAdd(a.tie_count, n);
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += WAVES * 64) a.tie_rows[s_tie_base + j] = s_tie[j];
    }
    // the last workgroup of a shard to leave resets the shard's counters: the next search over this query grid starts
    // from zero (every workgroup has received its last ticket before it counts itself out)
    if (threadIdx.x == 0) {
        const int done = atomicAdd(&a.qtick[(8 + shard) * kTickStride], 1);
        if (done == Gk - 1) {
            __hip_atomic_store(&a.qtick[shard * kTickStride], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.qtick[(8 + shard) * kTickStride], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
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
            const Slot sl = load_slot(&g.tab[s]);
            if (sl.key == key) { ccount = sl.cnt; cstart = sl.start; break; }
            if (sl.key == kEmptyKey) break;
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
        hipLaunchKernelGGL(k_grid_insert, dim3(blocks), dim3(256), 0, st, sup, nb, g);
        hipLaunchKernelGGL(k_grid_starts, dim3((2 * ns + 255) / 256), dim3(256), 0, st, 2 * ns, nb, g);
        hipLaunchKernelGGL(k_grid_scatter, dim3(blocks), dim3(256), 0, st, sup, nb, g);
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
    if (pass != 2) {
        KpProfScope ev(st, nq, cols, ns, 0, 4);
        hipExtLaunchKernelGGL((k_radius_query<kListCapFast, false, kQueryWaves>), dim3(blocks), dim3(kQueryWaves * 64), 0, st, ev.a,
                              ev.b, 0, q, nq, qlen, nb, r2, g, cols, reinterpret_cast<long long*>(out_idx), out_count,
                              out_max_count, status, out_tie_rows, out_tie_count, group);
    }
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

namespace pcrcg {
// The cell-cooperative search (k_radius_cells) over a query grid; pass 0: followed by the per-query second pass for the
// rows it hands over (the public entry point), pass 1: the cell kernel only (the pyramid builder reads status / out_max
// anyway and launches radius_query_pass(.., 2) for the tables that need it: normally none).
int radius_cells_pass(const void* qgrid, const float* q, int nq, const int* qlen, const void* sgrid, int ns, const int* slen,
                      int nb, int group, float radius, int cols, int64_t* out_idx, int* out_count, int* out_max_count,
                      int* status, int* out_tie_rows, int* out_tie_count, hipStream_t st, int pass) {
    PCRCG_CHECK_ARG(nq >= 0 && ns >= 0 && nb >= 1 && cols >= 1 && group >= 0 && radius > 0.0f);
    PCRCG_CHECK_ARG((out_tie_rows == nullptr) == (out_tie_count == nullptr));
    PCRCG_CHECK_ARG(qgrid && sgrid && out_idx && out_max_count);
    PCRCG_CHECK_ARG(pass == 1 || (q && qlen && slen));
    if (nq == 0) return PCRCG_OK;
    bool ok;
    GridView gq = grid_view(const_cast<void*>(qgrid), grid_bytes(nq, nb), nq, nb, &ok);
    GridView gs = grid_view(const_cast<void*>(sgrid), grid_bytes(ns, nb), ns, nb, &ok);
    const float r2 = radius * radius;  // neighbors.cpp:226
    const double reach = (double)radius * (1.0 + 1e-6);
    const int max_blocks_env = debug_opts().radius_blocks;
    // Alone on the GPU the search is fastest with the CUs full (86 VGPRs: five workgroups per CU; 1536 workgroups -> 56 us
    // for the 60 000-row table, 1280 -> 59).  Inside the pair engine (pass 1: the pyramid builder) the front-end stream
    // shares the CUs with three model streams and the grid size hardly matters: 256 / 512 / 768 / 1024 / 1536 workgroups
    // -> 457 / 463 / 461 / 460 / 463 pairs/s (same box); 512 as in rounds 1-3.
    const int max_blocks = max_blocks_env > 0 ? max_blocks_env : (pass == 1 ? 512 : 1536);
    int blocks = (nq + 7) / 8;                     // never more workgroups than there can be cells worth having one
    if (blocks > max_blocks) blocks = max_blocks;
    CellArgs ca;
    ca.qhdr = gq.hdr; ca.qtick = gq.qtick; ca.qspts = gq.spts; ca.ckey = gq.ckey; ca.cinfo = gq.cinfo;
    ca.shdr = gs.hdr; ca.soff = gs.soff; ca.tab = gs.tab; ca.spts = gs.spts;
    ca.out_idx = reinterpret_cast<long long*>(out_idx); ca.out_count = out_count; ca.out_max = out_max_count;
    ca.status = status; ca.tie_rows = out_tie_rows; ca.tie_count = out_tie_count;
    ca.reach = reach; ca.r2 = r2; ca.nb = nb; ca.cols = cols; ca.group = group;
    ca.prof = nullptr;
    // The ticket block lives in the QUERY grid and cleans itself (the last workgroup of a shard to leave zeroes it), which
    // is enough for the pyramid builder: it owns its grids, rebuilds them per call and walks them on one stream.  A caller
    // of the public entry point may have aborted an earlier walk or may hand over a grid some other walk left mid-way, so
    // here the block is reset on the launch stream first (4 KB; walks of ONE query grid must still not overlap in time:
    // include/pcrcg.h, and pcrcg_amd/ops.py orders them by event).
    if (pass != 1) PCRCG_CHECK_HIP(hipMemsetAsync(gq.qtick, 0, 16 * kTickStride * sizeof(int), st));
    if (debug_opts().radius_prof) {       // measurement aid: per-phase shader cycles, printed when the process exits
        static long long* prof = nullptr;
        if (!prof) {
            (void)hipMalloc(&prof, 16 * sizeof(long long));
            (void)hipMemset(prof, 0, 16 * sizeof(long long));
            static struct Dump {
                ~Dump() {
                    long long h[16];
                    if (hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
                    static const char* nm[8] = {"barrier A (state out, previous readers done)", "probes | query staging, barrier B",
                                                "candidate staging", "barrier C", "sweep", "rank sort", "row store", "hand-over"};
                    fprintf(stderr, "k_radius_cells: %lld wave-cells, %lld queries\n", h[8], h[9]);
                    double tot = 0;
                    for (int k = 0; k < 8; ++k) tot += (double)h[k];
                    for (int k = 0; k < 8; ++k)
                        fprintf(stderr, "k_radius_cells phase %d %-46s %6.2f %%  %9.0f cycles per %s\n", k, nm[k], 100.0 * h[k] / (tot > 0 ? tot : 1),
                                (double)h[k] / (double)((k >= 4 && k <= 6) ? (h[9] ? h[9] : 1) : (h[8] ? h[8] : 1)), (k >= 4 && k <= 6) ? "query" : "wave-cell");
                }
            } dump;
        }
        ca.prof = prof;
        hipLaunchKernelGGL((k_radius_cells<kCellCand, kCellListCap, kQueryWaves, true>), dim3(blocks), dim3(kQueryWaves * 64), 0, st, ca);
    } else {
        KpProfScope ev(st, nq, cols, ns, 1, 4);        // bench.py's radius roofline: the kernel's own start / stop events
        hipExtLaunchKernelGGL((k_radius_cells<kCellCand, kCellListCap, kQueryWaves>), dim3(blocks), dim3(kQueryWaves * 64), 0, st,
                              ev.a, ev.b, 0, ca);
    }
    PCRCG_CHECK_LAUNCH();
    if (pass == 1) return PCRCG_OK;
    return radius_query_pass(q, nq, qlen, ns, slen, nb, group, radius, sgrid, cols, out_idx, out_count, out_max_count, status,
                             out_tie_rows, out_tie_count, st, 2);
}
}  // namespace pcrcg

extern "C" {

int pcrcg_radius_query_cells(const void* qgrid, const float* q, int nq, const int* qlen, const void* sgrid, int ns,
                             const int* slen, int nb, int group, float radius, int cols, int64_t* out_idx, int* out_count,
                             int* out_max_count, int* status, int* out_tie_rows, int* out_tie_count, void* stream) {
    return pcrcg::radius_cells_pass(qgrid, q, nq, qlen, sgrid, ns, slen, nb, group, radius, cols, out_idx, out_count,
                                    out_max_count, status, out_tie_rows, out_tie_count, as_stream(stream), 0);
}

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
