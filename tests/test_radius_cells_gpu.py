"""The cell-cooperative, LDS-staged radius search (csrc/radius.hip: k_radius_cells, C ABI pcrcg_radius_query_cells) on
its OWN edge paths, against the CPU oracle's brute force in (d2, index) order (oracle/front_end.c: the definition of the
result set, ref:cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:268-301,319-325) and against the per-query kernel:

  * query grids whose cells are larger / smaller than the support grid's, by factors that are not powers of two
    (3x3x3 ... 8x8x8 support cells within reach, and more than the kernel stages -> handed to the second pass),
  * cells holding more queries than one staging batch (64) and more candidates than LDS stages (1024),
  * rows of more than 128 hits (second pass), rows of exactly equal distances (the tie-break path and the tie report),
  * ragged batches with an EMPTY query cloud and an empty support cloud, one-point clouds,
  * several groups of clouds stacked into one call (indices relative to the group, per-group padding and maximum).

Bar: bit-exact tables, counts, column maxima, status 0, and the same SET of reported tie rows as the per-query kernel."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import frontend as OF
from pcrcg_amd import _lib, ops

pytestmark = pytest.mark.gpu


def _cells(cuda, q, s, ql, sl, radius, cols, q_radius=None, group=0):
    """-> (idx, max_count per group, status, counts, tie rows) of pcrcg_radius_query_cells; the query grid is built with
    `q_radius` (default: the search radius)."""
    L = _lib.lib()
    tq, ts = torch.from_numpy(q).to(cuda), torch.from_numpy(s).to(cuda)
    tql, tsl = torch.from_numpy(np.asarray(ql, np.int32)).to(cuda), torch.from_numpy(np.asarray(sl, np.int32)).to(cuda)
    sg = ops.CellGrid(ts, tsl, radius)
    qg = ops.CellGrid(tq, tql, radius if q_radius is None else q_radius)
    nq, nb = len(q), len(ql)
    ngrp = nb // group if group else 1
    idx = torch.full((max(nq, 1), cols), -7, dtype=torch.int64, device=cuda)
    meta = torch.zeros(ngrp + 2, dtype=torch.int32, device=cuda)            # max per group, status, tie count
    counts = torch.full((max(nq, 1),), -1, dtype=torch.int32, device=cuda)
    ties = torch.zeros(max(nq, 1), dtype=torch.int32, device=cuda)
    _lib.check(L.pcrcg_radius_query_cells(qg.grid.data_ptr(), tq.data_ptr(), nq, tql.data_ptr(), sg.grid.data_ptr(), len(s),
                                          tsl.data_ptr(), nb, group, ctypes.c_float(radius), cols, idx.data_ptr(),
                                          counts.data_ptr(), meta.data_ptr(), meta[ngrp:].data_ptr(), ties.data_ptr(),
                                          meta[ngrp + 1:].data_ptr(), torch.cuda.current_stream().cuda_stream),
               "pcrcg_radius_query_cells")
    torch.cuda.synchronize()
    m = meta.tolist()
    return idx[:nq].cpu().numpy(), m[:ngrp], m[ngrp], counts[:nq].cpu().numpy(), set(ties[:m[ngrp + 1]].tolist())


def _check(cuda, q, s, ql, sl, radius, cols=None, q_radius=None):
    want = OF.oracle_batch_query(q, s, ql, sl, radius)                      # (d2, index) order, padded with len(s)
    cols = cols or max(want.shape[1], 1)
    idx, mx, status, counts, ties = _cells(cuda, q, s, ql, sl, radius, cols, q_radius)
    assert status == 0
    assert mx == [want.shape[1]]
    exp = np.full((len(q), cols), len(s), np.int64)
    w = min(cols, want.shape[1])
    exp[:, :w] = want[:, :w]
    assert (idx == exp).all(), "rows differ: %d" % int((idx != exp).any(1).sum())
    assert (counts == (want < len(s)).sum(1)).all()
    # the reported tie rows are those the per-query kernel reports
    tq, ts = torch.from_numpy(q).to(cuda), torch.from_numpy(s).to(cuda)
    g = ops.CellGrid(ts, torch.from_numpy(np.asarray(sl, np.int32)).to(cuda), radius)
    res = g.query(tq, torch.from_numpy(np.asarray(ql, np.int32)).to(cuda), cols, want_ties=True)
    n_old = int(res[1][2])
    assert ties == set(res[3][:n_old].tolist())
    assert torch.equal(res[0].cpu(), torch.from_numpy(idx))
    return want


@pytest.mark.parametrize("factor", [1.0, 2.0, 0.5, 3.0, 0.37, 1.7, 5.0])
def test_query_cells_of_any_size(cuda, factor):
    """conv (factor 1), pool (2), upsample (0.5) are the pyramid's cases; the others make 5..8 support cells per axis
    fall within reach of one query cell -- above 256 cells the kernel hands the cell's rows to the second pass."""
    rng = np.random.RandomState(int(factor * 100))
    s = (rng.rand(6000, 3) * [0.9, 0.7, 0.3] - 0.2).astype(np.float32)      # negative coordinates included
    q = (rng.rand(2500, 3) * [0.9, 0.7, 0.3] - 0.2).astype(np.float32)
    _check(cuda, q, s, [1400, 1100], [3500, 2500], 0.06, q_radius=0.06 * factor)


def test_crowded_cells(cuda):
    """One tight cluster: its cell holds > 64 queries (several staging batches), > 1024 candidates (does not fit LDS ->
    second pass) and rows of > 128 hits (second pass); a sparse halo around it takes the LDS path in the same launch."""
    rng = np.random.RandomState(5)
    cluster = (rng.rand(1500, 3) * 0.03 + 0.5).astype(np.float32)
    halo = rng.rand(3000, 3).astype(np.float32)
    pts = np.concatenate([cluster, halo]).astype(np.float32)
    want = _check(cuda, pts, pts, [len(pts)], [len(pts)], 0.05, cols=60)
    assert want.shape[1] > 1024                                               # rows longer than any staging list
    # a medium cluster: cells of > 64 queries whose neighbourhood still fits (the multi-batch LDS path itself)
    medium = np.concatenate([(rng.rand(300, 3) * 0.02 + 0.3).astype(np.float32), rng.rand(500, 3).astype(np.float32)])
    want = _check(cuda, medium, medium, [len(medium)], [len(medium)], 0.05)
    assert 128 < want.shape[1] <= 1024


def test_exactly_equal_distances_take_the_tie_break_path(cuda):
    """Lattice points: nearly every row holds groups of exactly equal d2 (32-bit ranks collide -> (d2, index) order with
    both words), inside and beyond the kept columns; duplicates (d2 = 0 several times) included."""
    rng = np.random.RandomState(9)
    pts = (np.round(rng.rand(4000, 3) * 24) / 64).astype(np.float32)
    pts = np.concatenate([pts, pts[:200]])                                    # exact duplicates
    lens = [2500, 1700]
    for cols in (8, 40, None):
        _check(cuda, pts, pts, lens, lens, 0.0625, cols=cols)


def test_ragged_batches_with_empty_clouds(cuda):
    rng = np.random.RandomState(2)
    s = np.concatenate([rng.rand(700, 3), rng.rand(1, 3) + 3, rng.rand(400, 3) - 2]).astype(np.float32)
    q = np.concatenate([rng.rand(50, 3), rng.rand(90, 3) - 2]).astype(np.float32)
    # query cloud 1 is EMPTY (its support cloud holds one point); then support cloud 2 empty instead
    _check(cuda, q, s, [50, 0, 90], [700, 1, 400], 0.15)
    s2 = np.concatenate([rng.rand(700, 3), rng.rand(400, 3) - 2]).astype(np.float32)
    q2 = np.concatenate([rng.rand(50, 3), rng.rand(7, 3) + 3, rng.rand(90, 3) - 2]).astype(np.float32)
    _check(cuda, q2, s2, [50, 7, 90], [700, 0, 400], 0.15)
    one = rng.rand(2, 3).astype(np.float32)
    _check(cuda, one, one, [1, 1], [1, 1], 0.5)


def test_groups_of_clouds(cuda):
    """Two independent pairs stacked into one call (group = 2): indices relative to the pair's first support, rows padded
    with the pair's support count, one column maximum per pair -- the pairs' own tables stacked."""
    rng = np.random.RandomState(4)
    pairs = [((rng.rand(1200, 3) * 0.6).astype(np.float32), [700, 500]), ((rng.rand(900, 3) * 0.5).astype(np.float32), [300, 600])]
    pts = np.concatenate([p for p, _ in pairs])
    lens = sum((l for _, l in pairs), [])
    single = [OF.oracle_batch_query(p, p, l, l, 0.07) for p, l in pairs]
    cols = 30
    idx, mx, status, counts, _ = _cells(cuda, pts, pts, lens, lens, 0.07, cols, group=2)
    assert status == 0 and mx == [t.shape[1] for t in single]
    row = 0
    for (p, _), t in zip(pairs, single):
        exp = np.full((len(p), cols), len(p), np.int64)
        w = min(cols, t.shape[1])
        exp[:, :w] = t[:, :w]
        assert (idx[row:row + len(p)] == exp).all()
        assert (counts[row:row + len(p)] == (t < len(p)).sum(1)).all()
        row += len(p)


def test_abi_argument_checks(cuda):
    L = _lib.lib()
    pts = torch.rand(10, 3, device=cuda)
    lens = torch.tensor([10], dtype=torch.int32, device=cuda)
    g = ops.CellGrid(pts, lens, 0.2)
    out = torch.empty((10, 4), dtype=torch.int64, device=cuda)
    meta = torch.zeros(3, dtype=torch.int32, device=cuda)
    ok = (g.grid.data_ptr(), pts.data_ptr(), 10, lens.data_ptr(), g.grid.data_ptr(), 10, lens.data_ptr(), 1, 0, ctypes.c_float(0.2),
          4, out.data_ptr(), None, meta.data_ptr(), meta[1:].data_ptr(), None, None, None)
    assert L.pcrcg_radius_query_cells(*ok) == 0
    for pos, bad in ((0, None), (4, None), (10, 0), (9, ctypes.c_float(0.0)), (11, None), (13, None), (15, meta.data_ptr())):
        args = list(ok)
        args[pos] = bad
        assert L.pcrcg_radius_query_cells(*args) != 0, pos                    # rejected, with pcrcg_last_error() set
        assert L.pcrcg_last_error()
