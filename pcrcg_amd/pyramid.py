"""Point-pyramid builder on the MI355X: the device-resident counterpart of the reference's
``collate_fn_descriptor`` (ref:datasets/dataloader.py:203-400), ``batch_grid_subsampling_kpconv``
(:14-52), ``batch_neighbors_kpconv`` (:54-69) and ``calibrate_neighbors`` (:402-434).

The reference runs this on the CPU inside DataLoader workers and ships ten index tables to the GPU;
here the stacked cloud is uploaded once and every table is produced in HBM.  Per level the supports
are binned once and the conv / pool / upsample searches reuse that grid (all three use the level's
radius, :273,:298,:301)."""
import os

import numpy as np
import torch

from . import ops
from .config import as_config

_I32 = torch.int32


def _layer_plan(config):
    """Walk config.architecture like ref:datasets/dataloader.py:252-359 and return one entry per
    pyramid level: (radius, has_conv, pooled, dl_next)."""
    arch = config.architecture
    r_normal = config.first_subsampling_dl * config.conv_radius
    levels, layer_blocks = [], []
    for i, block in enumerate(arch):
        if "global" in block or "upsample" in block:
            break
        if not ("pool" in block or "strided" in block):
            layer_blocks.append(block)
            if i < len(arch) - 1 and "upsample" not in arch[i + 1]:
                continue
        deform_conv = any("deformable" in b for b in layer_blocks[:-1])
        r_conv = r_normal * config.deform_radius / config.conv_radius if deform_conv else r_normal
        pooled = "pool" in block or "strided" in block
        r_pool = r_normal * config.deform_radius / config.conv_radius if "deformable" in block else r_normal
        levels.append(dict(r_conv=r_conv, has_conv=bool(layer_blocks), pooled=pooled, r_pool=r_pool,
                           dl=2 * r_normal / config.conv_radius))
        r_normal *= 2
        layer_blocks = []
    return levels


TIE_ORDERS = ("auto", "reference", "index")


TIE_STATUS_TEXT = {1: "the KD-forest kernel gave up waiting for work", 2: "traversal stack overflow",
                   3: "KD-tree and cell-grid searches disagree", 4: "row longer than the staging width"}


def check_tie_status(code):
    if code != 0:
        raise RuntimeError("pcrcg_amd.build_pyramid: restoring the reference's tie order failed (status %d: %s)" % (
            code, TIE_STATUS_TEXT.get(code, "?")))


def build_pyramid(points, lengths, config, neighborhood_limits, want_counts=False, tie_order=None,
                  defer_tie_check=False, mirror=False, side_streams=None):
    """points [N0,3] f32 and lengths [B] i32 on the device -> the reference's batch dict
    (ref:datasets/dataloader.py:363-380) restricted to the keys KPFCNN.forward reads, all on the
    device: points, neighbors, pools, upsamples (int64, shadow = support count), stack_lengths,
    features (ones).  Extra keys: 'stack_lengths_host' (list of python int lists) and, if
    want_counts, 'neighbor_counts' (untruncated list lengths of the conv tables, for calibration).

    tie_order: order of neighbours at EXACTLY equal distance, which decides what `[:, :limit]` keeps and what
    column 0 of an upsample table holds (csrc/tieorder.hip):
      "auto"       the reference's order (nanoflann traversal + std::sort replayed); the KD-forest is built only
                   when some row holds such a tie, and only those rows are redone  [default];
      "reference"  the same through the KD-forest for every row (a cross-check of "auto");
      "index"      ascending index -- a defined order, NOT the reference's; no forest.
    None = the environment variable PCRCG_TIE_ORDER if set, else "auto".
    mirror: build through the op-by-op Python mirror (pyramid_steps) instead of the C++ builder (same tables entry for
    entry; want_counts / defer_tie_check / tie_order="reference" always take the mirror).
    defer_tie_check: the restore step reports "cannot happen on sane clouds" conditions (a KD-tree with more than
    128 pending branches on one query's path, ...) through a device status word.  By default it is read back here
    (one more host sync); with defer_tie_check=True it is returned as out["tie_status"] ([1] i32 device tensor or
    None) and the CALLER must pass its value to check_tie_status() once the stream has finished (a pipelining caller
    does).

    side_streams: (subsampling stream, KD-forest stream) -- two torch streams on hardware dispatchers OTHER than the current
    stream's and otherwise idle (pcrcg_amd.ops.streams_on_other_dispatchers() finds such): the C++ builder then runs the
    chain as a DAG over the three streams, which halves it on an otherwise idle GPU (one 2 x 30 000-point pair: 1.01
    against 1.65 ms; profiles/r06_chain_latency_alone.txt).  None: one line of kernels on the current stream.

    The C++ builder (the default path) sizes its levels from a row bound and waits for the stream once, at the end of the
    chain.  The Python mirror needs four host round trips (three subsampled row counts, one for all table widths);
    pyramid_steps() is that build as a generator that YIELDS at those points."""
    if not points.is_cuda:
        raise RuntimeError("pcrcg_amd.build_pyramid: points must be on a HIP device (no CPU path)")
    mode = tie_order if tie_order is not None else os.environ.get("PCRCG_TIE_ORDER", "auto")
    if not want_counts and not defer_tie_check and mode in ("auto", "index") and not mirror:
        return build_pyramid_native(points, lengths, config, neighborhood_limits, mode, side_streams)
    steps = pyramid_steps(points, lengths, config, neighborhood_limits, want_counts, tie_order, defer_tie_check)
    try:
        while True:
            next(steps).synchronize()
    except StopIteration as done:
        return done.value


def pyramid_steps(points, lengths, config, neighborhood_limits, want_counts=False, tie_order=None,
                  defer_tie_check=False, defer_restore=False):
    """Generator form of build_pyramid: enqueues kernels on the current stream, yields a torch.cuda.Event whenever
    it needs a value from the device (resume it -- under the same current stream -- once the event has passed),
    and returns the batch dict through StopIteration.value.
    defer_restore: do not launch the tie-order restore step (KD-forest + reorder) here; return it as
    out["restore"] = (callable -> status tensor or None, [tensors it touches]) for the consumer to run on ITS stream
    before it reads the tables (a pipelining caller runs it on the pair's model stream, which has slack, instead of the
    front-end stream, which is the pipeline's bottleneck)."""
    config = as_config(config)
    if tie_order is None:
        tie_order = os.environ.get("PCRCG_TIE_ORDER", "auto")
    if tie_order not in TIE_ORDERS:
        raise ValueError(f"pcrcg_amd.build_pyramid: tie_order must be one of {TIE_ORDERS}")
    if not points.is_cuda:
        raise RuntimeError("pcrcg_amd.build_pyramid: points must be on a HIP device (no CPU path)")
    pts = points.to(torch.float32).contiguous()
    lens = lengths.to(_I32).contiguous()
    plan = _layer_plan(config)
    want_ties = tie_order != "index"
    in_points, in_neighbors, in_pools, in_ups, in_lens = [], [], [], [], []
    metas, tables, counts_out = [], [], []
    empty_idx = torch.zeros((0, 1), dtype=torch.int64, device=pts.device)
    carried = None   # grid over the current level's points built for the previous level's upsample search

    def search(grid, key, layer, q_pts, q_lens, sup_level, limit, keep_counts=False, query_grid=None):
        res = grid.query(q_pts, q_lens, limit, want_counts=keep_counts, want_ties=want_ties, query_grid=query_grid)
        idx, meta = res[0], res[1]
        metas.append(meta)
        tables.append(dict(key=key, layer=layer, q=q_pts, qlen=q_lens, sup_level=sup_level, radius=grid.radius,
                           counts=res[2] if len(res) > 2 else None, ties=res[3] if want_ties else None))
        if keep_counts:
            counts_out.append(res[2])
        return idx

    for layer, lv in enumerate(plan):
        limit = int(neighborhood_limits[layer])
        grid = None
        if lv["has_conv"]:
            grid = carried if carried is not None and carried.radius == float(lv["r_conv"]) else \
                ops.CellGrid(pts, lens, lv["r_conv"])
            conv_i = search(grid, "neighbors", layer, pts, lens, layer, limit, keep_counts=want_counts, query_grid=grid)
        else:
            conv_i = empty_idx
        if lv["pooled"]:
            rows, pool_b, m_dev = ops.grid_subsample_launch(pts, lens, lv["dl"])
            m_host, ev = ops.host_copy(m_dev)
            yield ev                                                         # host round trip: the row count
            pool_p = rows[:int(m_host[0])]
            if grid is None or grid.radius != float(lv["r_pool"]):
                grid = carried if carried is not None and carried.radius == float(lv["r_pool"]) else \
                    ops.CellGrid(pts, lens, lv["r_pool"])
            # the coarse level's grid first: support grid of the upsample search, conv grid of the next level and the
            # QUERY grid of the pool search (csrc/pyramid.hip does the same: every query set walks a grid of its own)
            up_grid = ops.CellGrid(pool_p, pool_b, 2 * lv["r_pool"])
            pool_i = search(grid, "pools", layer, pool_p, pool_b, layer, limit, query_grid=up_grid)
            up_i = search(up_grid, "upsamples", layer, pts, lens, layer + 1, limit, query_grid=grid)
            # the next level's conv and pool searches use these supports with this radius: hand it on
            carried = up_grid
        else:
            pool_i, up_i = empty_idx, empty_idx
            pool_p = torch.zeros((0, 3), dtype=torch.float32, device=pts.device)
            pool_b = torch.zeros((0,), dtype=torch.int64, device=pts.device)
            carried = None
        in_points.append(pts)
        in_neighbors.append(conv_i)
        in_pools.append(pool_i)
        in_ups.append(up_i)
        in_lens.append(lens)
        pts, lens = pool_p, pool_b
    out = {"points": in_points, "neighbors": in_neighbors, "pools": in_pools, "upsamples": in_ups,
           "stack_lengths": in_lens}
    # one host round trip for all tables and levels: column counts (neighbors[:, :limit] keeps FEWER columns
    # when the longest list is shorter than the limit, ref:datasets/dataloader.py:65-67), status, the
    # number of rows holding a tie, and the per-level cloud lengths
    nb = in_lens[0].shape[0]
    host_t, ev = ops.host_copy(torch.cat([m for m in metas] + [l.to(_I32) for l in in_lens]))
    yield ev
    host = host_t.tolist()
    meta_h = [host[3 * i:3 * i + 3] for i in range(len(metas))]
    lens_h = [host[3 * len(metas) + nb * i:3 * len(metas) + nb * (i + 1)] for i in range(len(in_lens))]
    redo = []
    for tab, (max_count, status, tie_rows) in zip(tables, meta_h):
        if status != 0:
            raise RuntimeError(f"pcrcg_amd.build_pyramid: radius search capacity exceeded ({tab['key']}[{tab['layer']}])")
        tab["max_count"] = max_count
        if max_count > 0 and (tie_order == "reference" or (tie_order == "auto" and tie_rows > 0)):
            redo.append((tab, tie_rows))
    out["tie_status"] = None
    if redo:
        for tab, _ in redo:
            tab["idx"] = out[tab["key"]][tab["layer"]]        # full-width, contiguous (trimmed views are made below)
        all_rows = tie_order == "reference"
        if defer_restore:
            touched = list(in_points) + list(in_lens)
            for tab, _ in redo:
                touched += [t for t in (tab["idx"], tab["q"], tab["qlen"], tab["ties"], tab["counts"]) if t is not None]
            out["restore"] = (lambda: _restore_reference_order(redo, in_points, in_lens, all_rows), touched)
        else:
            status = _restore_reference_order(redo, in_points, in_lens, all_rows)
            if defer_tie_check:
                out["tie_status"] = status
            else:
                status_h, ev = ops.host_copy(status)
                yield ev
                check_tie_status(int(status_h[0]))
    for tab in tables:
        t = out[tab["key"]][tab["layer"]]
        if tab["max_count"] < t.shape[1]:
            out[tab["key"]][tab["layer"]] = t[:, :max(tab["max_count"], 0)]
    out["features"] = torch.ones((in_points[0].shape[0], 1), dtype=torch.float32, device=points.device)
    out["stack_lengths_host"] = lens_h
    if want_counts:
        out["neighbor_counts"] = counts_out
    return out


def _restore_reference_order(redo, level_points, level_lens, all_rows):
    """Rows with exactly equal distances -> the reference's order (ops.KdForest): one forest over the clouds of
    all levels, then one launch for all tables that reported such rows.  -> status [1] i32 (device)."""
    nb = level_lens[0].shape[0]
    forest = ops.KdForest(torch.cat(level_points, 0), torch.cat([l.to(_I32) for l in level_lens], 0))
    status = torch.zeros(1, dtype=_I32, device=level_points[0].device)
    forest.reorder_tables([dict(idx=tab["idx"], q=tab["q"], qlen=tab["qlen"], cloud0=nb * tab["sup_level"],
                                radius=tab["radius"], max_count=min(tab["max_count"], 8192), counts=tab["counts"],
                                rows=None if all_rows else tab["ties"], nrows=None if all_rows else tie_rows)
                           for tab, tie_rows in redo], status)
    return status


class NativePyramid:
    """Front end of one pair through pcrcg_pyramid_build (csrc/pyramid.hip): ONE call into the library enqueues the
    whole pyramid (levels sized from the bound `shrink`; the call waits for its one host round trip itself, with the GIL
    released).  One instance owns
    an arena, pinned scratch and a ring of status words; it serves one host thread / one stream at a time
    (pcrcg_amd/pairstream.py gives every worker its own).  tie_order "auto" or "index"; "reference" (every row
    through the KD-forest, a cross-check) and want_counts stay with the Python mirror pyramid_steps."""

    STATUS_RING = 64

    def __init__(self, config, neighborhood_limits, tie_order=None, up_nearest=False):
        import ctypes
        from . import _lib
        from .runner import Batch, PyramidCfg
        config = as_config(config)
        if tie_order is None:
            tie_order = os.environ.get("PCRCG_TIE_ORDER", "auto")
        if tie_order not in ("auto", "index"):
            raise ValueError("pcrcg_amd.NativePyramid: tie_order must be 'auto' or 'index'")
        self._ct, self._lib, self._Batch = ctypes, _lib, Batch
        plan = _layer_plan(config)
        c = PyramidCfg()
        c.n_levels = len(plan)
        for l, lv in enumerate(plan):
            c.r_conv[l], c.r_pool[l], c.dl[l] = float(lv["r_conv"]), float(lv["r_pool"]), float(lv["dl"])
            c.has_conv[l], c.pooled[l], c.limit[l] = int(lv["has_conv"]), int(lv["pooled"]), int(neighborhood_limits[l])
        c.tie_order = 0 if tie_order == "index" else 1
        # one-column upsample tables (the nearest coarse point: all the network reads of them) -- for the pair engine,
        # whose batches go to pcrcg_kpfcnn_forward only; the batch-dict builders keep the reference's full tables
        c.up_nearest = int(bool(up_nearest))
        self.cfg, self.levels = c, len(plan)
        self.scratch = torch.empty(512, dtype=_I32, pin_memory=True)
        self.status = torch.zeros(self.STATUS_RING, dtype=_I32, pin_memory=True)
        self.arena, self.shrink, self.calls = None, 0.5, 0
        self.side = None        # set_side_stream(): a second stream for the KD-forests of the restore step

    def set_side_streams(self, sub=None, forest=None):
        """Further torch streams of the caller's for the parts of the chain that need nothing from the searches
        (pcrcg_pyramid_cfg.side_stream / side_stream2): `sub` runs the three subsamplings back to back, `forest` the
        KD-forests of the tie-order restore step (None: behind the subsamplings on `sub`).  Both None: one stream."""
        self.side = (sub, forest)
        self.cfg.side_stream = sub.cuda_stream if sub is not None else None
        self.cfg.side_stream2 = forest.cuda_stream if forest is not None else None

    def restore(self, deferred, slot):
        """Enqueue a deferred tie-order restore step (build(..., defer_restore=True)) on the CURRENT stream."""
        self._lib.check(self._lib.lib().pcrcg_pyramid_restore_run(self._ct.byref(deferred), self.status.data_ptr() + 4 * slot,
                                                                  torch.cuda.current_stream().cuda_stream),
                        "pcrcg_pyramid_restore_run")

    def build(self, points, lengths, fresh_arena=False, defer_restore=False, group=0):
        """points [N0,3] f32, lengths [B] i32 on the device -- or LISTS of such tensors, one entry per part (the pairs of a
        grouped build): the builder copies the parts behind each other into its arena (pcrcg_pyramid_build_parts), no
        torch.cat kernel runs; enqueues on the CURRENT stream.
        -> (pcrcg_batch mirror, arena tensor it points into, per-level cloud lengths (python lists),
            slot of this call's status word in self.status -- valid once the stream has drained).
        defer_restore: the tie-order restore step is not enqueued; a fifth value, its descriptor, is returned for
        restore() to run on the stream that will read the tables.
        group: 0 = the clouds form one batch.  2 = the clouds are len(lengths) / 2 independent PAIRS stacked into one
        call (the front end's kernel chain is latency-bound: two pairs cost little more than one); the first value then
        is a ctypes array of that many pcrcg_batch mirrors, each with its own tables."""
        ct, L = self._ct, self._lib.lib()
        parts = isinstance(points, (list, tuple))
        pts_l = [p.to(torch.float32).contiguous() for p in (points if parts else [points])]
        lens_l = [l.to(_I32).contiguous() for l in (lengths if parts else [lengths])]
        if not all(p.is_cuda for p in pts_l):
            raise RuntimeError("pcrcg_amd.build_pyramid: points must be on a HIP device (no CPU path)")
        pts, lens = pts_l[0], lens_l[0]
        n0, nb = sum(int(p.shape[0]) for p in pts_l), sum(int(l.shape[0]) for l in lens_l)
        stream = torch.cuda.current_stream().cuda_stream
        if group and nb % group:
            raise RuntimeError("pcrcg_amd.NativePyramid: the number of clouds is not a multiple of `group`")
        self.cfg.group = int(group)
        nbatch = nb // group if group else 1
        h_len = (ct.c_int * (self.levels * nb))()
        from .runner import PyramidRestore
        deferred = PyramidRestore() if defer_restore else None
        slot = self.calls % self.STATUS_RING
        self.calls += 1
        while True:
            self.cfg.shrink = float(self.shrink)
            need = L.pcrcg_pyramid_ws_bytes(n0, nb, ct.byref(self.cfg))
            if need == 0:
                raise RuntimeError("pcrcg_pyramid_ws_bytes rejected the configuration")
            arena = self.arena
            if fresh_arena or arena is None or arena.numel() < need or arena.device != pts.device:
                arena = torch.empty(int(need), dtype=torch.uint8, device=pts.device)
                if not fresh_arena:
                    self.arena = arena
            b = (self._Batch * nbatch)() if group else self._Batch()
            if parts:
                k = len(pts_l)
                pp = (ct.c_void_p * k)(*[p.data_ptr() for p in pts_l])
                lp = (ct.c_void_p * k)(*[l.data_ptr() for l in lens_l])
                pn = (ct.c_int * k)(*[int(p.shape[0]) for p in pts_l])
                ln = (ct.c_int * k)(*[int(l.shape[0]) for l in lens_l])
                rc = L.pcrcg_pyramid_build_parts(pp, pn, lp, ln, k, ct.byref(self.cfg), arena.data_ptr(), arena.numel(),
                                                 self.scratch.data_ptr(), ct.byref(b), h_len,
                                                 self.status.data_ptr() + 4 * slot,
                                                 ct.byref(deferred) if defer_restore else None, stream)
            else:
                rc = L.pcrcg_pyramid_build(pts.data_ptr(), n0, lens.data_ptr(), nb, ct.byref(self.cfg), arena.data_ptr(),
                                           arena.numel(), self.scratch.data_ptr(), ct.byref(b), h_len,
                                           self.status.data_ptr() + 4 * slot,
                                           ct.byref(deferred) if defer_restore else None, stream)
            if rc == -2 and self.shrink < 1.0:      # PCRCG_EWORKSPACE: this cloud keeps more rows per level than assumed
                self.shrink = 1.0
                continue
            self._lib.check(rc, "pcrcg_pyramid_build")
            break
        lens_h = [[int(h_len[l * nb + i]) for i in range(nb)] for l in range(self.levels)]
        if defer_restore:
            return b, arena, lens_h, slot, deferred
        return b, arena, lens_h, slot

    def as_dict(self, b, arena, lens_h, part=None):
        """The reference's batch dict as zero-copy views into the arena.  part = (first cloud, clouds) selects the
        lengths of one group when `b` is one batch of a grouped build."""
        base = arena.data_ptr()
        if part is not None:
            lens_h = [row[part[0]:part[0] + part[1]] for row in lens_h]
        nb = len(lens_h[0])

        def view(ptr, nbytes, dtype):
            off = ptr - base
            return arena[off:off + nbytes].view(dtype)

        def table(t):
            if t.idx and t.rows > 0 and t.cols > 0:
                return view(t.idx, t.rows * t.ld * 8, torch.int64).view(t.rows, t.ld)[:, :t.cols]
            if t.idx and t.rows > 0:          # every list empty: the reference's [rows, 0]
                return torch.zeros((t.rows, 0), dtype=torch.int64, device=arena.device)
            return torch.zeros((0, 1), dtype=torch.int64, device=arena.device)   # no table at the last level

        out = {"points": [], "neighbors": [], "pools": [], "upsamples": [], "stack_lengths": []}
        for l in range(b.n_levels):
            n = b.n_points[l]
            out["points"].append(view(b.points[l], n * 12, torch.float32).view(n, 3))
            out["neighbors"].append(table(b.neighbors[l]))
            out["pools"].append(table(b.pools[l]))
            out["upsamples"].append(table(b.upsamples[l]))
            out["stack_lengths"].append(view(b.stack_lengths[l], nb * 4, _I32))
        out["features"] = view(b.features, b.n_points[0] * 4, torch.float32).view(-1, 1)
        out["stack_lengths_host"] = lens_h
        return out


def build_pyramid_native(points, lengths, config, neighborhood_limits, tie_order=None, side_streams=None):
    """build_pyramid through the C++ builder (one FFI call); same dict, same tables entry for entry."""
    nat = NativePyramid(config, neighborhood_limits, tie_order)
    if side_streams is not None:
        nat.set_side_streams(*side_streams)
    b, arena, lens_h, slot = nat.build(points, lengths, fresh_arena=True)
    torch.cuda.current_stream().synchronize()
    check_tie_status(int(nat.status[slot]))
    out = nat.as_dict(b, arena, lens_h)
    out["tie_status"] = None
    return out


def _point2node(nodes, points):
    """ref:datasets/dataloader.py:90-105: nearest node of every point (dense distances)."""
    d = -2.0 * points @ nodes.t()
    d = d + (points ** 2).sum(-1, keepdim=True) + (nodes ** 2).sum(-1)[None, :]
    return torch.clamp(d, min=1e-12).argmin(1)


def _node_visibility(nodes, points, visible_idx):
    """ref:datasets/dataloader.py:107-152: fraction of each node's points that have a correspondence."""
    p2n = _point2node(nodes, points)
    tot = torch.ones(nodes.shape[0], device=nodes.device)
    idx, cts = torch.unique(p2n, return_counts=True)
    tot[idx] = cts.float()
    vis_mask = torch.zeros(points.shape[0], device=nodes.device)
    vis_mask[visible_idx] = 1.0
    vis_pts = vis_mask.nonzero().squeeze(1)
    vis = torch.zeros(nodes.shape[0], device=nodes.device)
    idx_, cts_ = torch.unique(p2n[vis_pts], return_counts=True)
    vis[idx_] = cts_.float()
    return vis / tot, p2n


def collate_fn_descriptor(list_data, config, neighborhood_limits, device=None, side_streams=None):
    """Same contract as ref:datasets/dataloader.py:203-400 (one pair per batch, :207), with the result
    already on the device.  Image-feature keys (:383-398) are passed through untouched.  side_streams: see build_pyramid."""
    assert len(list_data) == 1
    device = torch.device(device if device is not None else "cuda")
    item = list_data[0]
    src = torch.as_tensor(np.asarray(item["src_pcd"]), dtype=torch.float32).to(device)
    tgt = torch.as_tensor(np.asarray(item["tgt_pcd"]), dtype=torch.float32).to(device)
    points = torch.cat([src, tgt], 0)
    lengths = torch.tensor([src.shape[0], tgt.shape[0]], dtype=_I32, device=device)
    out = build_pyramid(points, lengths, config, neighborhood_limits, side_streams=side_streams)
    feats = np.concatenate([np.asarray(item["src_feats"]), np.asarray(item["tgt_feats"])], 0)
    out["features"] = torch.as_tensor(feats, dtype=torch.float32).to(device)
    out["rot"] = torch.as_tensor(np.asarray(item["rot"])).to(device)
    out["trans"] = torch.as_tensor(np.asarray(item["trans"])).to(device)
    corr = item["correspondences"]
    out["correspondences"] = corr
    out["src_pcd_raw"], out["tgt_pcd_raw"] = src, tgt
    out["sample"] = item.get("sample")
    # node-overlap labels of the coarsest level (:309-322): training labels, plain dense torch math
    nodes = out["points"][-1]
    n_src = out["stack_lengths_host"][-1][0]
    corr_t = (corr if isinstance(corr, torch.Tensor) else torch.as_tensor(np.asarray(corr))).to(device).long()
    src_vis, src_p2n = _node_visibility(nodes[:n_src], src, corr_t[:, 0])
    tgt_vis, tgt_p2n = _node_visibility(nodes[n_src:], tgt, corr_t[:, 1])
    out["node_overlap_gt"] = torch.cat([src_vis, tgt_vis])
    out["points2node"] = torch.cat([src_p2n, tgt_p2n])
    for key in item.keys():
        if key not in out and key not in ("src_pcd", "tgt_pcd", "src_feats", "tgt_feats"):
            out[key] = item[key]
    return out


def calibrate_neighbors(pairs, config, keep_ratio=0.8, samples_threshold=2000):
    """ref:datasets/dataloader.py:402-434: limits[l] = #{h : cumhist_l(h) < keep_ratio * total_l} over
    the histograms of untruncated neighbour counts.  `pairs` yields (points [N,3], lengths [B]) device
    tensors."""
    config = as_config(config)
    hist_n = int(np.ceil(4 / 3 * np.pi * (config.deform_radius + 1) ** 3))
    hists = np.zeros((config.num_layers, hist_n), dtype=np.int64)
    for points, lengths in pairs:
        b = build_pyramid(points, lengths, config, [hist_n] * config.num_layers, want_counts=True)
        for l, cnt in enumerate(b["neighbor_counts"]):
            c = torch.clamp(cnt.long(), max=hist_n)   # the reference sees at most hist_n columns
            hists[l] += torch.bincount(c, minlength=hist_n + 1)[:hist_n].cpu().numpy()
        if hists.sum(1).min() > samples_threshold:
            break
    cumsum = np.cumsum(hists.T, axis=0)
    return np.sum(cumsum < (keep_ratio * cumsum[hist_n - 1, :]), axis=0)
