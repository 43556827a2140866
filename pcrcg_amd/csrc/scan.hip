// scan.hip -- library plumbing: error string, status check and a device-wide int32 exclusive scan.
#include <cstring>

#include <atomic>
#include <mutex>

#include "common.h"

namespace pcrcg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;                            // consecutive items per thread
constexpr int kScanTile = kScanThreads * kScanItems;     // items per block

__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive scan of one value per thread across a block of kScanThreads; returns the exclusive
// prefix and stores the block total in *total (valid in all threads).
__device__ __forceinline__ int block_exclusive_scan(int v, int* total, int* smem /* [kScanThreads/64 + 1] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = wave_inclusive_scan(v, lane);
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    int wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) {
        int s = smem[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return wave_off + inc - v;
}

__global__ void __launch_bounds__(kScanThreads) scan_tile_sums(const int* __restrict__ in, int n,
                                                                int* __restrict__ partial) {
    __shared__ int smem[kScanThreads / 64 + 1];
    const long base = (long)blockIdx.x * kScanTile + (long)threadIdx.x * kScanItems;
    int s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i)
        if (base + i < n) s += in[base + i];
    int tot;
    block_exclusive_scan(s, &tot, smem);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// One block scans the tile sums in place (looping with a carry) and publishes the grand total.
__global__ void __launch_bounds__(kScanThreads) scan_partials(int* __restrict__ partial, int nt,
                                                               int* __restrict__ total) {
    __shared__ int smem[kScanThreads / 64 + 1];
    int carry = 0;
    for (int base = 0; base < nt; base += kScanThreads) {
        int i = base + threadIdx.x;
        int v = i < nt ? partial[i] : 0;
        int tot;
        int ex = block_exclusive_scan(v, &tot, smem);
        if (i < nt) partial[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

__global__ void __launch_bounds__(kScanThreads) scan_apply(const int* in, int* out,
                                                            int n, const int* __restrict__ partial,
                                                            int* __restrict__ total_single) {
    __shared__ int smem[kScanThreads / 64 + 1];
    const long base = (long)blockIdx.x * kScanTile + (long)threadIdx.x * kScanItems;
    int v[kScanItems];
    int s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        v[i] = base + i < n ? in[base + i] : 0;
        s += v[i];
    }
    int tot;
    int ex = block_exclusive_scan(s, &tot, smem) + (partial ? partial[blockIdx.x] : 0);
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        if (base + i < n) out[base + i] = ex;
        ex += v[i];
    }
    if (total_single && threadIdx.x == 0) *total_single = tot;  // single-tile case only
}

// The whole scan in ONE workgroup of 1024 threads walking 8192-item tiles with a running carry.  The front end's
// scans cover at most a few hundred thousand items and sit on a serial chain of small dependent kernels, where every
// launch costs its dispatch gap on top of its run time (under load ~10-15 us per kernel): one ~12 us kernel for
// 60 000 items instead of three kernels of 5-9 us each.
constexpr int kOneThreads = 1024;
constexpr int kOneTiles = 8;                                  // up to 8 x 8192 = 65 536 items
__global__ void __launch_bounds__(kOneThreads) scan_one_block(const int* in, int* out, int n, int* __restrict__ total) {
    __shared__ int smem[kOneThreads / 64];
    __shared__ int s_tot;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every tile's loads are issued before the first use: one memory round trip for the whole scan
    int v[kOneTiles][kScanItems];
#pragma unroll
    for (int t = 0; t < kOneTiles; ++t) {
        const long base = (long)t * kOneThreads * kScanItems + (long)threadIdx.x * kScanItems;
        if (base + kScanItems <= n) {                       // two 16-byte loads (in is 16-byte aligned, base % 8 == 0)
            const int4 a = *reinterpret_cast<const int4*>(in + base), b = *reinterpret_cast<const int4*>(in + base + 4);
            v[t][0] = a.x; v[t][1] = a.y; v[t][2] = a.z; v[t][3] = a.w;
            v[t][4] = b.x; v[t][5] = b.y; v[t][6] = b.z; v[t][7] = b.w;
        } else {
#pragma unroll
            for (int i = 0; i < kScanItems; ++i) v[t][i] = base + i < n ? in[base + i] : 0;
        }
    }
    int carry = 0;
#pragma unroll
    for (int t = 0; t < kOneTiles; ++t) {
        const long base = (long)t * kOneThreads * kScanItems + (long)threadIdx.x * kScanItems;
        if ((long)t * kOneThreads * kScanItems >= n) break;                             // block-uniform
        int s = 0;
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) {
            v[t][i] = base + i < n ? v[t][i] : 0;
            s += v[t][i];
        }
        const int inc = wave_inclusive_scan(s, lane);
        if (lane == 63) smem[wave] = inc;
        __syncthreads();
        if (wave == 0) {
            const int w = lane < kOneThreads / 64 ? smem[lane] : 0;
            const int winc = wave_inclusive_scan(w, lane);
            if (lane < kOneThreads / 64) smem[lane] = winc - w;      // exclusive offset of each wavefront
            if (lane == kOneThreads / 64 - 1) s_tot = winc;
        }
        __syncthreads();
        int ex = carry + smem[wave] + inc - s;
        carry += s_tot;
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) {
            const int x = v[t][i];
            v[t][i] = ex;
            ex += x;
        }
        if (base + kScanItems <= n) {
            *reinterpret_cast<int4*>(out + base) = make_int4(v[t][0], v[t][1], v[t][2], v[t][3]);
            *reinterpret_cast<int4*>(out + base + 4) = make_int4(v[t][4], v[t][5], v[t][6], v[t][7]);
        } else {
#pragma unroll
            for (int i = 0; i < kScanItems; ++i)
                if (base + i < n) out[base + i] = v[t][i];
        }
        __syncthreads();                                                  // smem / s_tot reused by the next tile
    }
    if (total && threadIdx.x == 0) *total = carry;
}

__global__ void zero_int(int* p) { *p = 0; }

}  // namespace

size_t scan_ws_bytes(int n) {
    int nt = n > 0 ? (n + kScanTile - 1) / kScanTile : 1;
    return carve_bytes((size_t)nt, sizeof(int));
}

int exclusive_scan_i32(const int* in, int* out, int n, int* total, void* ws, hipStream_t stream) {
    if (n <= 0) {
        if (total) hipLaunchKernelGGL(zero_int, dim3(1), dim3(1), 0, stream, total);
        PCRCG_CHECK_LAUNCH();
        return PCRCG_OK;
    }
    const int nt = (n + kScanTile - 1) / kScanTile;
    if (nt == 1) {
        hipLaunchKernelGGL(scan_apply, dim3(1), dim3(kScanThreads), 0, stream, in, out, n,
                           (const int*)nullptr, total);
        PCRCG_CHECK_LAUNCH();
        return PCRCG_OK;
    }
    if (n <= kOneTiles * kOneThreads * kScanItems && (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        hipLaunchKernelGGL(scan_one_block, dim3(1), dim3(kOneThreads), 0, stream, in, out, n, total);
        PCRCG_CHECK_LAUNCH();
        return PCRCG_OK;
    }
    int* partial = static_cast<int*>(ws);
    hipLaunchKernelGGL(scan_tile_sums, dim3(nt), dim3(kScanThreads), 0, stream, in, n, partial);
    hipLaunchKernelGGL(scan_partials, dim3(1), dim3(kScanThreads), 0, stream, partial, nt, total);
    hipLaunchKernelGGL(scan_apply, dim3(nt), dim3(kScanThreads), 0, stream, in, out, n,
                       (const int*)partial, (int*)nullptr);
    PCRCG_CHECK_LAUNCH();
    return PCRCG_OK;
}

// ---- the library's tuning / A-B switches (common.h: DebugOpts) ------------------------------------------------
namespace {
struct DebugName { const char* name; int DebugOpts::*field; };
const DebugName kDebugNames[] = {
    {"zero_arena", &DebugOpts::zero_arena}, {"stat_sums", &DebugOpts::stat_sums}, {"stat_sums_rows", &DebugOpts::stat_sums_rows},
    {"fuse_norm", &DebugOpts::fuse_norm}, {"fuse_pack", &DebugOpts::fuse_pack}, {"fuse_upsample", &DebugOpts::fuse_upsample}, {"gnn_merge", &DebugOpts::gnn_merge}, {"edge_rows", &DebugOpts::edge_rows}, {"att_mfma", &DebugOpts::att_mfma},
    {"radius_blocks", &DebugOpts::radius_blocks}, {"radius_eager_redo", &DebugOpts::radius_eager_redo},
    {"radius_cells", &DebugOpts::radius_cells}, {"kd_blocks", &DebugOpts::kd_blocks}, {"radius_prof", &DebugOpts::radius_prof},
    {"pyr_wait", &DebugOpts::pyr_wait}, {"pyr_trace", &DebugOpts::pyr_trace}, {"pyr_morton", &DebugOpts::pyr_morton}, {"c1_rows16", &DebugOpts::c1_rows16}, {"knock_tail", &DebugOpts::knock_tail}, {"att_tq", &DebugOpts::att_tq},
    {"kd_spin_limit", &DebugOpts::kd_spin_limit}, {"gemm_log", &DebugOpts::gemm_log}, {"x6_tile", &DebugOpts::x6_tile}, {"x6_order", &DebugOpts::x6_order}, {"x6_big", &DebugOpts::x6_big}, {"x6_h2", &DebugOpts::x6_h2}, {"train_side_stream", &DebugOpts::train_side_stream},
    {"x6_splitk", &DebugOpts::x6_splitk}, {"x6_t1", &DebugOpts::x6_t1}, {"x6_t2", &DebugOpts::x6_t2},
    {"gemm_tile", &DebugOpts::gemm_tile}, {"gemm_splitk", &DebugOpts::gemm_splitk},
    {"gemm_split_target", &DebugOpts::gemm_split_target}, {"deterministic", &DebugOpts::deterministic},
    {"bwd_mfma", &DebugOpts::bwd_mfma}};
// The options are IMMUTABLE snapshots behind one atomic pointer: the engine's front and model threads read them
// concurrently (their library calls hold no lock), pcrcg_debug_set publishes a new snapshot and never frees an old one
// (a handful of 100-byte objects per process), so a reader's reference stays valid and no field is ever seen half
// written.  The environment is read exactly once (std::call_once).  A multi-kernel call (a forward's sizing pass and
// its live pass) reads the options several times: pcrcg_debug_set must not be called while calls are in flight.
std::atomic<const DebugOpts*> g_debug{nullptr};
std::atomic<const DebugOpts*> g_debug_env{nullptr};     // what PCRCG_DEBUG set at first use: pcrcg_debug_set(NULL) goes back to it
std::once_flag g_debug_once;
std::mutex g_debug_set_lock;
const DebugOpts kDebugDefaults{};

// "name=value,name=value" on top of *base -> a new published snapshot; false (and the error text) on an unknown name or a
// malformed item
bool debug_parse(const char* spec, const DebugOpts* base) {
    DebugOpts d = *base;
    const char* p = spec;
    while (p && *p) {
        const char* end = strchr(p, ',');
        const size_t len = end ? (size_t)(end - p) : strlen(p);
        const char* eq = static_cast<const char*>(memchr(p, '=', len));
        bool ok = false;
        if (eq) {
            for (const DebugName& n : kDebugNames)
                if (strlen(n.name) == (size_t)(eq - p) && strncmp(n.name, p, (size_t)(eq - p)) == 0) {
                    d.*(n.field) = atoi(eq + 1);
                    ok = true;
                }
        }
        if (!ok && len > 0) {
            set_error("pcrcg_debug_set: unknown or malformed item '%.*s'", (int)len, p);
            return false;
        }
        p = end ? end + 1 : nullptr;
    }
    if (d.deterministic) {           // what the switch implies (common.h)
        d.stat_sums = 0;
        d.gemm_splitk = 1;       // (the split-bf16 / fp16 family keeps its split-K plans and reduces them in two passes)
    }
    g_debug.store(new DebugOpts(d), std::memory_order_release);
    return true;
}
}  // namespace

const DebugOpts& debug_opts() {
    std::call_once(g_debug_once, [] {
        g_debug.store(&kDebugDefaults, std::memory_order_release);
        if (const char* e = getenv("PCRCG_DEBUG"))
            if (!debug_parse(e, &kDebugDefaults)) fprintf(stderr, "libpcrcg_hip: PCRCG_DEBUG ignored: %s\n", g_err);
        g_debug_env.store(g_debug.load(std::memory_order_acquire), std::memory_order_release);
    });
    return *g_debug.load(std::memory_order_acquire);
}

}  // namespace pcrcg

extern "C" {

const char* pcrcg_last_error(void) { return pcrcg::g_err; }

int pcrcg_abi_version(void) { return PCRCG_ABI_VERSION; }   // include/pcrcg.h lists what each version changed

int pcrcg_debug_set(const char* spec) {
    const pcrcg::DebugOpts* cur = &pcrcg::debug_opts();   // the environment first, then this call on top of it
    std::lock_guard<std::mutex> hold(pcrcg::g_debug_set_lock);   // setters serialise; readers never block
    cur = pcrcg::g_debug.load(std::memory_order_acquire);
    if (!spec) {      // back to what the process started with: the defaults, or what PCRCG_DEBUG made of them
        pcrcg::g_debug.store(pcrcg::g_debug_env.load(std::memory_order_acquire), std::memory_order_release);
        return PCRCG_OK;
    }
    return pcrcg::debug_parse(spec, cur) ? PCRCG_OK : PCRCG_EBADARG;
}

// The deterministic debug mode (deterministic=1) allocates its scratch itself, one buffer per stream it has been used on,
// keyed by the stream handle and kept until the process ends.  A host that destroys streams (a recycled handle would inherit
// a stale entry) or wants the memory back calls this with those streams DRAINED; the next deterministic call allocates anew.
int pcrcg_debug_release(void) {
    PCRCG_CHECK_HIP(hipDeviceSynchronize());
    pcrcg::gemm_x6_release_det();
    pcrcg::trainops_release_det();
    return PCRCG_OK;
}

int pcrcg_check_status(const int* status, void* stream) {
    PCRCG_CHECK_ARG(status != nullptr);
    int h = 0;
    PCRCG_CHECK_HIP(hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, pcrcg::as_stream(stream)));
    PCRCG_CHECK_HIP(hipStreamSynchronize(pcrcg::as_stream(stream)));
    if (h != 0) {
        pcrcg::set_error("device status word = %d (capacity overflow)", h);
        return PCRCG_ECAPACITY;
    }
    return PCRCG_OK;
}
}
