"""Tensor-level wrappers over the C ABI (include/pcrcg.h).  PyTorch is used only for device memory
and streams; every computation below runs in the hand-written HIP kernels of libpcrcg_hip.so.
All tensors must live on a HIP device -- there is no CPU path."""
import ctypes

import torch

from . import _lib

_F32, _I32, _I64 = torch.float32, torch.int32, torch.int64


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"pcrcg_amd: `{name}` must be a tensor on a HIP device (no CPU fallback exists)")
    if t.dtype != dtype:
        raise RuntimeError(f"pcrcg_amd: `{name}` must have dtype {dtype}, got {t.dtype}")
    return t


def _rows(t, dtype, name):
    """2-D tensor with unit inner stride; returns (tensor, leading dimension in elements)."""
    _dev(t, dtype, name)
    if t.dim() != 2:
        raise RuntimeError(f"pcrcg_amd: `{name}` must be 2-D")
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    ld = t.stride(0) if t.shape[0] > 1 else t.shape[1]
    if ld < t.shape[1]:  # e.g. an expanded (stride-0) tensor
        t = t.contiguous()
        ld = t.shape[1]
    return t, max(ld, 1)


def _ptr(t):
    return None if t is None else t.data_ptr()


class _Workspaces:
    """Grow-only scratch buffers keyed by (device, stream, tag): stream-ordered reuse is safe."""

    def __init__(self):
        self._bufs = {}

    def get(self, tag, nbytes, device):
        key = (device.index, _stream(), tag)
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            self._bufs[key] = buf
        return buf


_ws = _Workspaces()

def kpconv_profile_start(gemm=False, radius=False, kpconv=True):
    """Bracket every KPConv gather/aggregate launch -- with gemm=True also every GEMM of the split-bf16 family, with
    radius=True every radius search of the front end -- with HIP start / stop events (on the launching stream)."""
    _lib.lib().pcrcg_profile_kpconv((1 if kpconv else 0) | (2 if gemm else 0) | (4 if radius else 0))


def kpconv_profile_stop(cap=1 << 16):
    """-> list of (milliseconds, nq, h, cin, cout, kind) per KPConv kernel launch since
    kpconv_profile_start(); kind 0 = gather/aggregate kernel (cout unknown: 0), 1 = fused kernel, 2 = bf16-storage
    gather kernel, 3 = GEMM (then the fields are M, N, K, bf16 products per element), 4 = radius search (queries, columns,
    supports, 1 = cell-cooperative kernel)."""
    import ctypes
    L = _lib.lib()
    ms = (ctypes.c_float * cap)()
    arr = [(ctypes.c_int * cap)() for _ in range(5)]
    n = L.pcrcg_profile_kpconv_read(ms, *arr, cap)
    L.pcrcg_profile_kpconv(0)
    if n < 0:
        raise RuntimeError("pcrcg_profile_kpconv_read failed")
    return [(ms[i],) + tuple(a[i] for a in arr) for i in range(n)]


# ------------------------------------------------------------------------------------------------
# front end
# ------------------------------------------------------------------------------------------------
def grid_subsample_launch(points, lengths, dl, max_p=0):
    """batch_grid_subsampling on device WITHOUT the host round trip: -> (rows [max(N,1),3] f32 of which the first M
    are valid, sub_lengths [B] i32, M [1] i32 on the device)."""
    L = _lib.lib()
    points = _dev(points, _F32, "points").contiguous()
    lengths = _dev(lengths, _I32, "lengths").contiguous()
    n, nb = points.shape[0], lengths.shape[0]
    out = torch.empty((max(n, 1), 3), dtype=_F32, device=points.device)
    out_len = torch.empty(nb, dtype=_I32, device=points.device)
    out_m = torch.zeros(1, dtype=_I32, device=points.device)
    nbytes = L.pcrcg_grid_subsample_ws_bytes(n, nb)
    ws = _ws.get("subsample", nbytes, points.device)
    _lib.check(L.pcrcg_grid_subsample_batch(points.data_ptr(), n, lengths.data_ptr(), nb, float(dl), int(max_p),
                                            out.data_ptr(), out_len.data_ptr(), out_m.data_ptr(), ws.data_ptr(),
                                            nbytes, _stream()), "pcrcg_grid_subsample_batch")
    return out, out_len, out_m


def grid_subsample(points, lengths, dl, max_p=0):
    """batch_grid_subsampling on device.  points [N,3] f32, lengths [B] i32 ->
    (sub_points [M,3] f32, sub_lengths [B] i32).  One host sync (to learn M)."""
    out, out_len, out_m = grid_subsample_launch(points, lengths, dl, max_p)
    return out[:int(out_m.item())], out_len


def host_copy(t):
    """Asynchronous device -> pinned-host copy of a small int tensor on the current stream:
    -> (pinned tensor, event recorded after the copy); read the tensor after event.synchronize()."""
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return host, ev


def umap_order(keys):
    """libstdc++ unordered_map iteration order of distinct uint64 keys (given as int64 bit patterns)."""
    L = _lib.lib()
    keys = _dev(keys, _I64, "keys").contiguous()
    m = keys.shape[0]
    order = torch.empty(m, dtype=_I32, device=keys.device)
    nbytes = L.pcrcg_umap_order_ws_bytes(m)
    ws = _ws.get("umap", nbytes, keys.device)
    _lib.check(L.pcrcg_umap_order(keys.data_ptr(), m, order.data_ptr(), ws.data_ptr(), nbytes, _stream()),
               "pcrcg_umap_order")
    return order


class CellGrid:
    """Hashed uniform cell grid over one level's supports; serves every query set of that radius."""

    def __init__(self, supports, lengths, radius):
        L = _lib.lib()
        self.supports = _dev(supports, _F32, "supports").contiguous()
        self.lengths = _dev(lengths, _I32, "s_batches").contiguous()
        self.radius = float(radius)
        self.ns, self.nb = self.supports.shape[0], self.lengths.shape[0]
        self.nbytes = L.pcrcg_cellgrid_ws_bytes(self.ns, self.nb)
        self.grid = torch.empty(self.nbytes, dtype=torch.uint8, device=self.supports.device)
        _lib.check(L.pcrcg_cellgrid_build(self.supports.data_ptr(), self.ns, self.lengths.data_ptr(), self.nb,
                                          self.radius, self.grid.data_ptr(), self.nbytes, _stream()),
                   "pcrcg_cellgrid_build")

    def query(self, queries, q_lengths, cols, want_counts=False, want_ties=False, query_grid=None):
        """-> (idx [Nq, cols] i64, meta [3] i32 device = (max_count, status, tie_rows)[, counts [Nq] i32]
        [, tie_rows [Nq] i32: the first meta[2] entries are the rows holding a group of exactly equal distances
        inside the kept columns, whose reference order KdForest.reorder restores]).
        query_grid: a CellGrid over `queries` (any radius; `self` for a self query) -> the cell-cooperative LDS-staged
        search (pcrcg_radius_query_cells); None -> the per-query kernel."""
        L = _lib.lib()
        queries = _dev(queries, _F32, "queries").contiguous()
        q_lengths = _dev(q_lengths, _I32, "q_batches").contiguous()
        if q_lengths.shape[0] != self.nb:
            raise RuntimeError("Wrong number of batch elements: different for queries and supports ")
        nq = queries.shape[0]
        idx = torch.empty((nq, int(cols)), dtype=_I64, device=queries.device)
        meta = torch.zeros(3, dtype=_I32, device=queries.device)  # [max_count, status, tie_rows]
        counts = torch.empty(nq, dtype=_I32, device=queries.device) if (want_counts or want_ties) else None
        ties = torch.empty(max(nq, 1), dtype=_I32, device=queries.device) if want_ties else None
        if query_grid is not None:
            if query_grid.ns != nq or query_grid.nb != self.nb:
                raise RuntimeError("query_grid was not built over these queries")
            # the cell search keeps its work-distribution tickets inside the query grid: two walks of one grid must not
            # overlap in time.  Same stream: ordered anyway; another stream: it waits for the previous walk's event.
            cur = torch.cuda.current_stream(queries.device)
            last = getattr(query_grid, "_walk", None)
            if last is not None and last[0] != cur.cuda_stream:
                cur.wait_event(last[1])
            _lib.check(L.pcrcg_radius_query_cells(query_grid.grid.data_ptr(), queries.data_ptr(), nq, q_lengths.data_ptr(),
                                                  self.grid.data_ptr(), self.ns, self.lengths.data_ptr(), self.nb, 0,
                                                  self.radius, int(cols), idx.data_ptr(), _ptr(counts), meta[0:1].data_ptr(),
                                                  meta[1:2].data_ptr(), _ptr(ties),
                                                  meta[2:3].data_ptr() if want_ties else None, _stream()),
                       "pcrcg_radius_query_cells")
            ev = torch.cuda.Event()
            ev.record(cur)
            query_grid._walk = (cur.cuda_stream, ev)
            out = (idx, meta)
            if want_counts or want_ties:
                out += (counts,)
            if want_ties:
                out += (ties,)
            return out
        _lib.check(L.pcrcg_radius_query_ex(queries.data_ptr(), nq, q_lengths.data_ptr(), self.ns,
                                           self.lengths.data_ptr(), self.nb, self.radius, self.grid.data_ptr(),
                                           int(cols), idx.data_ptr(), _ptr(counts), meta[0:1].data_ptr(),
                                           meta[1:2].data_ptr(), _ptr(ties), meta[2:3].data_ptr() if want_ties else None,
                                           _stream()), "pcrcg_radius_query_ex")
        out = (idx, meta)
        if want_counts or want_ties:
            out += (counts,)
        if want_ties:
            out += (ties,)
        return out


def stream_pipe_classes(streams):
    """Dispatcher class of every torch stream in `streams` (pcrcg_stream_pipe_classes: measured, ~1.5 ms per test, on an
    idle GPU): streams of one class share one of gfx950's four compute dispatchers and take turns kernel by kernel."""
    L = _lib.lib()
    n = len(streams)
    ptrs = (ctypes.c_void_p * n)(*[s.cuda_stream for s in streams])
    cls = (ctypes.c_int * n)()
    scratch = torch.zeros(16, dtype=torch.int32, device=streams[0].device)
    torch.cuda.synchronize(streams[0].device)
    with torch.cuda.device(streams[0].device):
        _lib.check(L.pcrcg_stream_pipe_classes(ptrs, n, cls, scratch.data_ptr()), "pcrcg_stream_pipe_classes")
    return [int(c) for c in cls]


def streams_on_other_dispatchers(n=2, reference=None, candidates=12):
    """-> n torch streams that sit on n DIFFERENT hardware dispatchers, none of them the dispatcher of `reference` (default:
    the current stream): what build_pyramid(side_streams=...) / NativePyramid.set_side_streams want on a GPU that is
    otherwise idle.  Classified by measurement (stream_pipe_classes): call it once, at start-up, with the GPU idle; fewer
    than n such streams among the candidates raises RuntimeError."""
    ref = reference if reference is not None else torch.cuda.current_stream()
    cands = [torch.cuda.Stream(device=ref.device) for _ in range(candidates)]
    cls = stream_pipe_classes([ref] + cands)
    out, seen = [], {cls[0]}
    for s, c in zip(cands, cls[1:]):
        if c not in seen:
            out.append(s)
            seen.add(c)
        if len(out) == n:
            return out
    raise RuntimeError("pcrcg_amd.ops.streams_on_other_dispatchers: only %d dispatcher classes beside the reference stream's "
                       "among %d candidates" % (len(out), candidates))


MAX_REORDER_JOBS = 12     # PCRCG_MAX_REORDER_JOBS


class ReorderJob(ctypes.Structure):
    """pcrcg_reorder_job (include/pcrcg.h)."""
    _fields_ = [("q", ctypes.c_void_p), ("qlen", ctypes.c_void_p), ("rows", ctypes.c_void_p), ("count", ctypes.c_void_p),
                ("idx", ctypes.c_void_p), ("nq", ctypes.c_int), ("nbq", ctypes.c_int), ("cloud0", ctypes.c_int),
                ("nrows", ctypes.c_int), ("max_count", ctypes.c_int), ("cols", ctypes.c_int), ("radius", ctypes.c_float),
                ("group", ctypes.c_int), ("sup", ctypes.c_void_p), ("forest", ctypes.c_void_p), ("forest_ns", ctypes.c_int),
                ("forest_nb", ctypes.c_int)]


class KdForest:
    """nanoflann-1.3.0-identical KD-trees (leaf size 10) over stacked clouds -- the clouds of every pyramid level in
    one forest -- used to give rows with exactly equal distances the reference's own order (csrc/tieorder.hip)."""

    def __init__(self, supports, lengths):
        L = _lib.lib()
        self.supports = _dev(supports, _F32, "supports").contiguous()
        self.lengths = _dev(lengths, _I32, "s_batches").contiguous()
        self.ns, self.nb = self.supports.shape[0], self.lengths.shape[0]
        self.nbytes = L.pcrcg_kdforest_ws_bytes(self.ns, self.nb)
        self.ws = torch.empty(self.nbytes, dtype=torch.uint8, device=self.supports.device)
        _lib.check(L.pcrcg_kdforest_build(self.supports.data_ptr(), self.ns, self.lengths.data_ptr(), self.nb,
                                          self.ws.data_ptr(), self.nbytes, _stream()), "pcrcg_kdforest_build")

    def reorder(self, idx, queries, q_lengths, cloud0, radius, max_count, rows=None, nrows=None, counts=None,
                status=None):
        """Rewrite rows of `idx` [Nq, cols] (in place) in the reference's order.  The query clouds search the
        forest's clouds cloud0 .. cloud0 + len(q_lengths) - 1.  rows [>= nrows] i32 = rows to redo (None: all)."""
        return self.reorder_tables([dict(idx=idx, q=queries, qlen=q_lengths, cloud0=cloud0, radius=radius,
                                         max_count=max_count, rows=rows, nrows=nrows, counts=counts)], status)

    def reorder_tables(self, tables, status=None):
        """The same for several tables (dicts with the arguments of `reorder`) in one launch."""
        L = _lib.lib()
        if len(tables) > MAX_REORDER_JOBS:
            for i in range(0, len(tables), MAX_REORDER_JOBS):
                status = self.reorder_tables(tables[i:i + MAX_REORDER_JOBS], status)
            return status
        jobs = (ReorderJob * max(len(tables), 1))()
        keep = []
        for j, t in zip(jobs, tables):
            idx = _dev(t["idx"], _I64, "idx")
            if not idx.is_contiguous():
                raise RuntimeError("pcrcg_amd.KdForest.reorder: idx must be contiguous")
            q = _dev(t["q"], _F32, "queries").contiguous()
            qlen = _dev(t["qlen"], _I32, "q_batches").contiguous()
            keep += [q, qlen]
            rows, counts = t.get("rows"), t.get("counts")
            j.q, j.qlen, j.rows, j.count, j.idx = q.data_ptr(), qlen.data_ptr(), _ptr(rows), _ptr(counts), idx.data_ptr()
            j.nq, j.nbq, j.cloud0 = idx.shape[0], qlen.shape[0], int(t["cloud0"])
            j.nrows = idx.shape[0] if rows is None else int(t["nrows"])
            j.max_count, j.cols, j.radius = int(t["max_count"]), idx.shape[1], float(t["radius"])
            j.group = int(t.get("group", 0))
            if status is None:
                status = torch.zeros(1, dtype=_I32, device=idx.device)
        if status is None:
            return None
        _lib.check(L.pcrcg_radius_reorder_jobs(ctypes.cast(jobs, ctypes.c_void_p), len(tables), self.supports.data_ptr(),
                                               self.ns, self.nb, self.ws.data_ptr(), status.data_ptr(), _stream()),
                   "pcrcg_radius_reorder_jobs")
        return status


# ------------------------------------------------------------------------------------------------
# model kernels
# ------------------------------------------------------------------------------------------------
def _operand_b(b):
    """[k, n] operand -> (tensor, ld, trans_b).  A transposed view of a row-major [n, k] matrix (e.g.
    `linear.weight.t()`) is passed as it is with trans_b = 1: no copy."""
    _dev(b, _F32, "b")
    if b.dim() != 2:
        raise RuntimeError("pcrcg_amd: `b` must be 2-D")
    k, n = b.shape
    if b.stride(1) == 1 and (k == 1 or b.stride(0) >= n):
        return b, (b.stride(0) if k > 1 else n), 0
    if b.stride(0) == 1 and (n == 1 or b.stride(1) >= k):
        return b, (b.stride(1) if n > 1 else k), 1
    b = b.contiguous()
    return b, n, 0


def _operand_a(a):
    """[m, k] operand -> (tensor, ld, trans_a): a transposed view of a row-major [k, m] matrix (`x.t()`) is
    passed as it is with trans_a = 1 (the weight-gradient products X^T dY)."""
    _dev(a, _F32, "a")
    if a.dim() != 2:
        raise RuntimeError("pcrcg_amd: `a` must be 2-D")
    m, k = a.shape
    if a.stride(1) == 1 and (m == 1 or a.stride(0) >= k):
        return a, (a.stride(0) if m > 1 else k), 0
    if a.stride(0) == 1 and (k == 1 or a.stride(1) >= m):
        return a, (a.stride(1) if k > 1 else m), 1
    return a.contiguous(), k, 0


def gemm(a, b, row_scale=None, bias=None, out=None, grad_operand=0):
    """out[m,n] = (a[m,k] @ b[k,n]) * row_scale[m] + bias[n] on the fp32 matrix cores.  Transposed views
    of row-major matrices are consumed in place (no copies) for either operand.  grad_operand (backward products):
    1 = `a` holds gradients, 2 = `b` does -- include/pcrcg_train.h pcrcg_gemm_f32_grad."""
    L = _lib.lib()
    a, lda, trans_a = _operand_a(a)
    b, ldb, trans_b = _operand_b(b)
    m, k = a.shape
    k2, n = b.shape
    if k != k2:
        raise RuntimeError(f"pcrcg_amd.gemm: inner dimensions differ ({k} vs {k2})")
    if out is None:
        out = torch.empty((m, n), dtype=_F32, device=a.device)
    out_, ldc = _rows(out, _F32, "out")
    if out_.data_ptr() != out.data_ptr():
        raise RuntimeError("pcrcg_amd.gemm: `out` must have unit inner stride")
    if row_scale is not None:
        row_scale = _dev(row_scale, _F32, "row_scale").contiguous()
    if bias is not None:
        bias = _dev(bias, _F32, "bias").contiguous()
    if grad_operand:
        _lib.check(L.pcrcg_gemm_f32_grad(a.data_ptr(), lda, int(trans_a), b.data_ptr(), ldb, trans_b, out.data_ptr(), ldc, m, n, k,
                                         _ptr(row_scale), _ptr(bias), int(grad_operand), _stream()), "pcrcg_gemm_f32_grad")
        return out
    if trans_a:
        _lib.check(L.pcrcg_gemm_f32_ex(a.data_ptr(), lda, 1, b.data_ptr(), ldb, trans_b, out.data_ptr(), ldc, m, n, k,
                                       _ptr(row_scale), _ptr(bias), _stream()), "pcrcg_gemm_f32_ex")
        return out
    _lib.check(L.pcrcg_gemm_f32(a.data_ptr(), lda, b.data_ptr(), ldb, trans_b, out.data_ptr(), ldc, m, n, k,
                                _ptr(row_scale), _ptr(bias), _stream()), "pcrcg_gemm_f32")
    return out


def gemm_fused(a, w, idx=None, out=None, accumulate=False, sums=None, slope=1.0, bias=None, eps=1e-5):
    """out (+)= f(a)[idx[:, 0]] @ w^T + bias for a [ns, k], w [n, k] (both k-contiguous).  idx: an int64 table whose first
    column selects the row of `a` for every output row (an index outside [0, ns) selects zeros: the shadow neighbour).
    sums: float64 [2, k] column sums of `a` -- then f(a) = lrelu(IN(a), slope) is applied on load (pcrcg_gemm_f32_fused)."""
    L = _lib.lib()
    a, lda = _rows(a, _F32, "a")
    w, ldb = _rows(w, _F32, "w")
    k, n = a.shape[1], w.shape[0]
    if idx is not None:
        idx, ld_idx = _rows(idx, _I64, "idx")
        m = idx.shape[0]
    else:
        ld_idx, m = 0, a.shape[0]
    if out is None:
        if accumulate:
            raise RuntimeError("pcrcg_amd.gemm_fused: accumulate needs `out`")
        out = torch.empty((m, n), dtype=_F32, device=a.device)
    if sums is not None and (sums.dtype != torch.float64 or tuple(sums.shape) != (2, k) or not sums.is_contiguous()):
        raise RuntimeError("pcrcg_amd.gemm_fused: sums must be a contiguous float64 [2, k] tensor")
    if bias is not None:
        bias = _dev(bias, _F32, "bias").contiguous()
    zero = torch.zeros(k + 8, dtype=_F32, device=a.device)
    _lib.check(L.pcrcg_gemm_f32_fused(a.data_ptr(), lda, _ptr(idx), ld_idx, a.shape[0], zero.data_ptr(), _ptr(sums),
                                      float(a.shape[0]), float(eps), float(slope), w.data_ptr(), ldb, _ptr(bias),
                                      out.data_ptr(), out.stride(0) if m > 1 else max(n, out.stride(0)), m, n, k,
                                      int(accumulate), _stream()), "pcrcg_gemm_f32_fused")
    return out


def kpconv(q_pts, s_pts, idx, x, kernel_points, weights, extent):
    """KPConv.forward (rigid / linear / sum): aggregate kernel + MFMA contraction with 1/n row scale.
    weights: [15, cin, cout] as stored in the reference state_dict."""
    L = _lib.lib()
    q_pts = _dev(q_pts, _F32, "q_pts").contiguous()
    s_pts = _dev(s_pts, _F32, "s_pts").contiguous()
    idx, ld_idx = _rows(idx, _I64, "neighb_inds")
    x = _dev(x, _F32, "x").contiguous()
    kp = _dev(kernel_points, _F32, "kernel_points").contiguous()
    nq, h = idx.shape
    ns, cin = x.shape
    kdim = kp.shape[0]
    if kdim != 15:
        raise RuntimeError("pcrcg_amd.kpconv: only 15 kernel points are supported")
    if cin % 4 != 0 and cin > 4:
        # e.g. PCR-CG's 129-channel first layer: zero-pad the channels so that the MFMA gather kernel applies
        # (zeros change neither the sums nor the neighbour count of the normaliser)
        pad = (-cin) % 4
        x = torch.nn.functional.pad(x, (0, pad))
        weights = torch.nn.functional.pad(weights, (0, 0, 0, pad))
        cin += pad
    wf = torch.empty((nq, kdim * cin), dtype=_F32, device=x.device)
    inv_n = torch.empty(nq, dtype=_F32, device=x.device)
    nbytes = L.pcrcg_kpconv_ws_bytes(ns)
    ws = _ws.get("kpconv", nbytes, x.device)
    w2 = _dev(weights, _F32, "weights").reshape(kdim * cin, -1)
    _lib.check(L.pcrcg_kpconv_aggregate(q_pts.data_ptr(), nq, s_pts.data_ptr(), ns, idx.data_ptr(), h, ld_idx,
                                        x.data_ptr(), cin, kp.data_ptr(), float(extent), wf.data_ptr(),
                                        inv_n.data_ptr(), ws.data_ptr(), nbytes, _stream()),
               "pcrcg_kpconv_aggregate")
    return gemm(wf, w2, row_scale=inv_n)


def inject_image_features(n_points, len_src, images, channels=128, width=None):
    """PCR-CG's image-feature injection (ref:models/architectures.py:195-514): -> x [n_points, channels + 1] f32 = ones
    with the 2-D features of the projected points written in.  `images`: list, IN THE REFERENCE'S WRITE ORDER (the last
    entry wins where projections overlap), of dicts with fmap [C,H,W] f32, inds2d [n,2] i64 (column, row), inds3d [n]
    i64, target (bool: indices are relative to the target cloud) and optionally valid [W,H] f32.
    width > channels + 1: the rows are that wide, the extra columns zero -- the form the network runners take the
    129-channel input in (132: the first KPConv's gather kernel wants rows of whole float4s; zero columns against
    zero-padded weights change nothing)."""
    L = _lib.lib()
    dev = images[0]["fmap"].device
    width = channels + 1 if width is None else int(width)
    if width < channels + 1:
        raise RuntimeError("pcrcg_amd.inject_image_features: width must be at least channels + 1")
    x = (torch.zeros if width > channels + 1 else torch.empty)((n_points, width), dtype=_F32, device=dev)
    _lib.check(L.pcrcg_fill2d(x.data_ptr(), width, n_points, channels + 1, 1.0, _stream()), "pcrcg_fill2d")
    for im in images:
        fmap = _dev(im["fmap"], _F32, "fmap").contiguous()
        if fmap.dim() != 3 or fmap.shape[0] != channels:
            raise RuntimeError("pcrcg_amd.inject_image_features: fmap must be [channels, H, W]")
        valid = im.get("valid")
        if valid is not None:
            valid = _dev(valid.to(_F32), _F32, "valid").contiguous()
            if tuple(valid.shape) != (fmap.shape[2], fmap.shape[1]):
                raise RuntimeError("pcrcg_amd.inject_image_features: valid must be [W, H] (the reference's layout)")
        i2 = _dev(im["inds2d"].to(_I64), _I64, "inds2d").contiguous()
        i3 = _dev(im["inds3d"].to(_I64), _I64, "inds3d").contiguous()
        _lib.check(L.pcrcg_inject_image_features(fmap.data_ptr(), channels, fmap.shape[1], fmap.shape[2], _ptr(valid),
                                                 i2.data_ptr(), i3.data_ptr(), i3.shape[0],
                                                 int(len_src) if im.get("target") else 0, n_points, x.data_ptr(),
                                                 width, _stream()), "pcrcg_inject_image_features")
    return x


def kpconv_bf16(q_pts, s_pts, idx, x, kernel_points, weights, extent, intermediates=False):
    """KPConv.forward of the bf16 feature-storage VARIANT (pcrcg_model.feature_bf16): the gathers read a bf16 copy of
    x, the aggregate is bf16 in memory, the contraction takes it as the bf16 operand against the exactly split fp32
    weights.  cin % 32 == 0.  intermediates=True also returns (x_bf16, wf_bf16, inv_n) as int16 / fp32 tensors."""
    L = _lib.lib()
    q_pts, s_pts = _dev(q_pts, _F32, "q_pts").contiguous(), _dev(s_pts, _F32, "s_pts").contiguous()
    idx, ld_idx = _rows(idx, _I64, "idx")
    x = _dev(x, _F32, "x").contiguous()
    kernel_points = _dev(kernel_points, _F32, "kernel_points").contiguous()
    nq, ns, h, cin, cout = q_pts.shape[0], s_pts.shape[0], idx.shape[1], x.shape[1], weights.shape[2]
    if cin % 32:
        raise RuntimeError("pcrcg_amd.kpconv_bf16: cin must be a multiple of 32")
    kk = 15 * cin
    wt = _dev(weights, _F32, "weights").reshape(-1, cout).t().contiguous()           # [cout, 15*cin], K-contiguous
    xb = torch.empty((ns, cin), dtype=torch.int16, device=x.device)
    wfb = torch.empty((nq, kk), dtype=torch.int16, device=x.device)
    inv_n = torch.empty(nq, dtype=_F32, device=x.device)
    out = torch.empty((nq, cout), dtype=_F32, device=x.device)
    nbytes = L.pcrcg_kpconv_ws_bytes(ns)
    ws = _ws.get("kpconv", nbytes, x.device)
    _lib.check(L.pcrcg_kpconv_aggregate_bf16(q_pts.data_ptr(), nq, s_pts.data_ptr(), ns, idx.data_ptr(), h, ld_idx,
                                             x.data_ptr(), cin, kernel_points.data_ptr(), float(extent), xb.data_ptr(),
                                             wfb.data_ptr(), inv_n.data_ptr(), ws.data_ptr(), nbytes, _stream()),
               "pcrcg_kpconv_aggregate_bf16")
    _lib.check(L.pcrcg_gemm_bf16a_f32_colstats(wfb.data_ptr(), kk, wt.data_ptr(), kk, out.data_ptr(), cout, nq, cout, kk,
                                               inv_n.data_ptr(), None, None, 0, None, _stream()),
               "pcrcg_gemm_bf16a_f32_colstats")
    return (out, xb, wfb, inv_n) if intermediates else out


def gemm_bf16a(a_bf16, b, row_scale=None, bias=None):
    """(a @ b^T) * row_scale + bias with a [m, k] torch.bfloat16 (k % 32 == 0, rows 16-byte aligned) and b [n, k]
    fp32 (pcrcg_gemm_bf16a_f32_colstats)."""
    L = _lib.lib()
    if a_bf16.dtype != torch.bfloat16 or not a_bf16.is_cuda:
        raise RuntimeError("pcrcg_amd.gemm_bf16a: `a` must be a bfloat16 tensor on the HIP device")
    a = a_bf16.contiguous()
    b = _dev(b, _F32, "b").contiguous()
    m, k = a.shape
    n = b.shape[0]
    out = torch.empty((m, n), dtype=_F32, device=a.device)
    if row_scale is not None:
        row_scale = _dev(row_scale, _F32, "row_scale").contiguous()
    if bias is not None:
        bias = _dev(bias, _F32, "bias").contiguous()
    _lib.check(L.pcrcg_gemm_bf16a_f32_colstats(a.data_ptr(), k, b.data_ptr(), k, out.data_ptr(), n, m, n, k,
                                               _ptr(row_scale), _ptr(bias), None, 0, None, _stream()),
               "pcrcg_gemm_bf16a_f32_colstats")
    return out


def gather_max(x, idx):
    L = _lib.lib()
    x = _dev(x, _F32, "x").contiguous()
    idx, ld_idx = _rows(idx, _I64, "inds")
    ns, c = x.shape
    nq, h = idx.shape
    out = torch.empty((nq, c), dtype=_F32, device=x.device)
    _lib.check(L.pcrcg_gather_max(x.data_ptr(), ns, c, idx.data_ptr(), nq, h, ld_idx, out.data_ptr(), _stream()),
               "pcrcg_gather_max")
    return out


def gather_first(x, idx, out=None):
    """closest_pool; `out` may be a column slice [nq, c] of a wider row-major buffer."""
    L = _lib.lib()
    x = _dev(x, _F32, "x").contiguous()
    idx, ld_idx = _rows(idx, _I64, "inds")
    ns, c = x.shape
    nq = idx.shape[0]
    if out is None:
        out = torch.empty((nq, c), dtype=_F32, device=x.device)
    ld_out = out.stride(0) if nq > 1 else max(c, out.stride(0))
    if out.stride(1) != 1 and c > 1:
        raise RuntimeError("pcrcg_amd.gather_first: `out` must have unit inner stride")
    _lib.check(L.pcrcg_gather_first(x.data_ptr(), ns, c, idx.data_ptr(), nq, ld_idx, out.data_ptr(), ld_out,
                                    _stream()), "pcrcg_gather_first")
    return out


def instnorm_stats(x, eps=1e-5):
    L = _lib.lib()
    x, ldx = _rows(x, _F32, "x")
    n, c = x.shape
    stats = torch.empty(2 * c, dtype=_F32, device=x.device)
    nbytes = L.pcrcg_instnorm_ws_bytes(c)
    ws = _ws.get("instnorm", nbytes, x.device)
    _lib.check(L.pcrcg_instnorm_stats(x.data_ptr(), n, c, ldx, float(eps), stats.data_ptr(), ws.data_ptr(), nbytes,
                                      _stream()), "pcrcg_instnorm_stats")
    return stats


def instnorm_apply(x, stats, slope=1.0, res=None, res_stats=None, out=None):
    L = _lib.lib()
    x, ldx = _rows(x, _F32, "x")
    n, c = x.shape
    ldr = 0
    if res is not None:
        res, ldr = _rows(res, _F32, "res")
    if out is None:
        out = torch.empty((n, c), dtype=_F32, device=x.device)
    ldy = out.stride(0) if n > 1 else max(c, out.stride(0))
    _lib.check(L.pcrcg_instnorm_apply(x.data_ptr(), n, c, ldx, stats.data_ptr(), _ptr(res), ldr, _ptr(res_stats),
                                      float(slope), out.data_ptr(), ldy, _stream()), "pcrcg_instnorm_apply")
    return out


def instnorm_colsums(x):
    """float64 [2, c]: column sums and column sums of squares of x [n, c] (one launch, fp64 atomics)."""
    L = _lib.lib()
    x, ldx = _rows(x, _F32, "x")
    sums = torch.zeros((2, x.shape[1]), dtype=torch.float64, device=x.device)
    if x.shape[0] > 0:
        _lib.check(L.pcrcg_instnorm_colsums(x.data_ptr(), x.shape[0], x.shape[1], ldx, sums.data_ptr(), _stream()),
                   "pcrcg_instnorm_colsums")
    return sums


def instnorm_apply_sums(x, sums, slope=1.0, res=None, res_sums=None, eps=1e-5, out=None, count=None):
    """lrelu(IN(x) [+ res | + IN(res)], slope) with the statistics given as float64 column sums [2, c] =
    (sum_r x, sum_r x^2) -- the form the runner's GEMM epilogues leave (pcrcg_instnorm_apply_sums)."""
    L = _lib.lib()
    x, ldx = _rows(x, _F32, "x")
    n, c = x.shape
    if sums.dtype != torch.float64 or tuple(sums.shape) != (2, c) or not sums.is_contiguous():
        raise RuntimeError("pcrcg_amd.instnorm_apply_sums: sums must be a contiguous float64 [2, c] tensor")
    ldr = 0
    if res is not None:
        res, ldr = _rows(res, _F32, "res")
    if out is None:
        out = torch.empty((n, c), dtype=_F32, device=x.device)
    ldy = out.stride(0) if n > 1 else max(c, out.stride(0))
    _lib.check(L.pcrcg_instnorm_apply_sums(x.data_ptr(), n, c, ldx, sums.data_ptr(), float(n if count is None else count),
                                           float(eps), _ptr(res), ldr,
                                           _ptr(res_sums), float(slope), out.data_ptr(), ldy, _stream()),
               "pcrcg_instnorm_apply_sums")
    return out


def instnorm_lrelu(x, slope, eps=1e-5):
    """InstanceNorm over rows followed by LeakyReLU(slope); slope 1.0 = identity."""
    return instnorm_apply(x, instnorm_stats(x, eps), slope)


def knn(coords, k):
    L = _lib.lib()
    coords = _dev(coords, _F32, "coords").contiguous()
    n = coords.shape[0]
    idx = torch.empty((n, k), dtype=_I32, device=coords.device)
    _lib.check(L.pcrcg_knn(coords.data_ptr(), n, int(k), idx.data_ptr(), _stream()), "pcrcg_knn")
    return idx


def edgeconv_reduce(ctr, nbr, idx, eps=1e-5):
    """-> (emax [n,c], stats [2c]) for e[i,j,:] = ctr[i,:] + nbr[idx[i,j],:]."""
    L = _lib.lib()
    ctr, ld_ctr = _rows(ctr, _F32, "ctr")
    nbr, ld_nbr = _rows(nbr, _F32, "nbr")
    idx = _dev(idx, _I32, "idx").contiguous()
    n, c = ctr.shape
    k = idx.shape[1]
    emax = torch.empty((n, c), dtype=_F32, device=ctr.device)
    stats = torch.empty(2 * c, dtype=_F32, device=ctr.device)
    nbytes = L.pcrcg_edgeconv_ws_bytes(c)
    ws = _ws.get("edgeconv", nbytes, ctr.device)
    _lib.check(L.pcrcg_edgeconv_reduce(ctr.data_ptr(), ld_ctr, nbr.data_ptr(), ld_nbr, idx.data_ptr(), n, k, c,
                                       float(eps), emax.data_ptr(), c, stats.data_ptr(), ws.data_ptr(), nbytes,
                                       _stream()), "pcrcg_edgeconv_reduce")
    return emax, stats


def edgeconv_reduce_sums(ctr, nbr, idx):
    """-> (emax [n,c], sums f64 [2,c]) for e[i,j,:] = ctr[i,:] + nbr[idx[i,j],:]: the statistics as sums over all (i,j)
    (finish with instnorm_apply_sums(emax, sums, count = n * k))."""
    L = _lib.lib()
    ctr, ld_ctr = _rows(ctr, _F32, "ctr")
    nbr, ld_nbr = _rows(nbr, _F32, "nbr")
    idx = _dev(idx, _I32, "idx").contiguous()
    n, c = ctr.shape
    emax = torch.empty((n, c), dtype=_F32, device=ctr.device)
    sums = torch.zeros((2, c), dtype=torch.float64, device=ctr.device)
    _lib.check(L.pcrcg_edgeconv_reduce_sums(ctr.data_ptr(), ld_ctr, nbr.data_ptr(), ld_nbr, idx.data_ptr(), n, idx.shape[1],
                                            c, emax.data_ptr(), c, sums.data_ptr(), _stream()), "pcrcg_edgeconv_reduce_sums")
    return emax, sums


def softmax_rows_(x, scale=1.0):
    """In-place softmax(x * scale) over the last dimension of a 2-D tensor."""
    L = _lib.lib()
    x_, ld = _rows(x, _F32, "x")
    if x_.data_ptr() != x.data_ptr():
        raise RuntimeError("pcrcg_amd.softmax_rows_: tensor must have unit inner stride")
    _lib.check(L.pcrcg_softmax_rows(x.data_ptr(), x.shape[0], x.shape[1], ld, float(scale), _stream()),
               "pcrcg_softmax_rows")
    return x


def attention(q, k, v, heads, scale=None):
    """Multi-head attention in one launch: head h = column block [h*d, (h+1)*d) of q [n, heads*d], k / v [ms, heads*d];
    out = cat_h softmax(scale * q_h k_h^T) v_h, scale = 1/sqrt(d) by default.  d in {16, 32, 48, 64, 128}."""
    L = _lib.lib()
    q, ldq = _rows(q, _F32, "q")
    k, ldk = _rows(k, _F32, "k")
    v, ldv = _rows(v, _F32, "v")
    n, ch = q.shape
    ms = k.shape[0]
    d = ch // heads
    if d * heads != ch or k.shape[1] != ch or v.shape != k.shape or not L.pcrcg_attention_supported(d):
        raise RuntimeError(f"pcrcg_amd.attention: unsupported widths ({ch} channels, {heads} heads)")
    out = torch.empty((n, ch), dtype=_F32, device=q.device)
    _lib.check(L.pcrcg_attention(q.data_ptr(), ldq, k.data_ptr(), ldk, v.data_ptr(), ldv, out.data_ptr(), ch, n, ms, heads,
                                 d, float(d ** -0.5 if scale is None else scale), _stream()), "pcrcg_attention")
    return out


def softmax_matvec(x, vec, scale=1.0):
    """softmax(x * scale, dim=1) @ vec for x [rows, cols], vec [cols] (any stride) -> [rows]."""
    L = _lib.lib()
    x, ld = _rows(x, _F32, "x")
    vec = _dev(vec, _F32, "vec")
    if vec.dim() != 1 or vec.shape[0] != x.shape[1]:
        raise RuntimeError("pcrcg_amd.softmax_matvec: vec must be 1-D with x.shape[1] entries")
    y = torch.empty(x.shape[0], dtype=_F32, device=x.device)
    _lib.check(L.pcrcg_softmax_matvec(x.data_ptr(), x.shape[0], x.shape[1], ld, float(scale), vec.data_ptr(),
                                      int(vec.stride(0)) if vec.shape[0] > 1 else 1, y.data_ptr(), 1, _stream()),
               "pcrcg_softmax_matvec")
    return y


# ------------------------------------------------------------------------------------------------
# training-side rows (include/pcrcg_train.h)
# ------------------------------------------------------------------------------------------------
def feature_argmax(a, b, want_best=False):
    """argmax_j <a_i, b_j> for every row of a [n,c] over the rows of b [m,c] (int64 [n]); the n x m score
    matrix of `torch.matmul(a, b.T).max(1)` (ref:lib/loss.py:209-213) is never materialised."""
    L = _lib.lib()
    a, lda = _rows(a, _F32, "a")
    b, ldb = _rows(b, _F32, "b")
    n, c = a.shape
    m = b.shape[0]
    if b.shape[1] != c:
        raise RuntimeError("pcrcg_amd.feature_argmax: feature widths differ")
    if m == 0:
        raise RuntimeError("pcrcg_amd.feature_argmax: `b` has no rows")
    arg = torch.empty(n, dtype=_I64, device=a.device)
    best = torch.empty(n, dtype=_F32, device=a.device) if want_best else None
    nbytes = L.pcrcg_feature_argmax_ws_bytes(n)
    ws = _ws.get("feature_argmax", nbytes, a.device)
    _lib.check(L.pcrcg_feature_argmax(a.data_ptr(), lda, n, b.data_ptr(), ldb, m, c, arg.data_ptr(), _ptr(best),
                                      ws.data_ptr(), nbytes, _stream()), "pcrcg_feature_argmax")
    return (arg, best) if want_best else arg
