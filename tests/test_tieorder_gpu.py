"""The reference's neighbour order inside groups of EXACTLY equal distance (pcrcg_amd/csrc/tieorder.hip): the HIP
front end must return the reference's tables entry for entry, ties included -- against the raw golden digests of
the unmodified reference C++ (tests/golden/frontend_digests.json["sha256"], frontend_mini.npz) and against the
oracle's restatement of nanoflann + std::sort (oracle/front_end.c, itself pinned to the reference)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import frontend as OF
from pcrcg_amd import ops, synthetic
from pcrcg_amd.config import indoor_config
from pcrcg_amd.pyramid import build_pyramid

pytestmark = pytest.mark.gpu


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _stack(recipe, seed=0):
    src, tgt = synthetic.pair(recipe, seed)
    return np.concatenate([src, tgt]), np.array([len(src), len(tgt)], np.int32)


def _tables(batch):
    for l in range(4):
        yield f"conv{l}", batch["neighbors"][l]
        if l < 3:
            yield f"pool{l}", batch["pools"][l]
            yield f"up{l}", batch["upsamples"][l]


def _pyramid(cuda, recipe, limits, tie_order):
    pts, lens = _stack(recipe)
    return build_pyramid(torch.from_numpy(pts).to(cuda), torch.from_numpy(lens).to(cuda), indoor_config(), limits,
                         tie_order=tie_order)


@pytest.mark.parametrize("tie_order", ["auto", "reference"])
def test_mini_full_tables_equal_reference(cuda, golden_dir, tie_order):
    g = np.load(os.path.join(golden_dir, "frontend_mini.npz"))
    batch = _pyramid(cuda, "mini", [1000] * 4, tie_order)
    for name, t in _tables(batch):
        got = t.cpu().numpy()
        assert got.shape == g[name].shape and (got == g[name]).all(), name


@pytest.mark.parametrize("recipe", ["C1", "S30k", "T8k"])
@pytest.mark.parametrize("tie_order", ["auto", "reference"])
def test_untruncated_tables_hash_to_the_reference_digests(cuda, golden_dir, recipe, tie_order):
    """T8k: 13 705 of 16 000 level-0 rows hold a tie.  `auto` redoes only the reported rows, `reference` all rows."""
    dig = json.load(open(os.path.join(golden_dir, "frontend_digests.json")))[recipe]
    batch = _pyramid(cuda, recipe, [1000] * 4, tie_order)
    for name, t in _tables(batch):
        got = t.cpu().numpy().astype(np.int32)
        assert list(got.shape) == dig[name]["shape"], name
        assert _sha(got) == dig[name]["sha256"], name


@pytest.mark.parametrize("recipe,limits", [("T8k", [25, 36, 45, 42]), ("C1", [20, 30, 33, 34]), ("mini", [9, 17, 22, 30])])
def test_truncated_tables_keep_what_the_reference_keeps(cuda, recipe, limits):
    """`[:, :limit]` of the reference's tables (ref:datasets/dataloader.py:65-69): the cut falls inside tie groups."""
    batch = _pyramid(cuda, recipe, limits, "auto")
    pts, lens = _stack(recipe)
    r, dl = 0.0625, 0.05
    differs_from_index_order = 0
    for l in range(4):
        want = {f"conv{l}": (pts, pts, lens, lens, r)}
        if l < 3:
            sp, sl = OF.oracle_subsample_batch(pts, lens, dl)
            want[f"pool{l}"] = (sp, pts, sl, lens, r)
            want[f"up{l}"] = (pts, sp, lens, sl, 2 * r)
        for name, (q, s, ql, sl_, rad) in want.items():
            key = {"conv": "neighbors", "pool": "pools", "up": "upsamples"}[name[:-1]]
            got = batch[key][l].cpu().numpy()
            ref = OF.oracle_batch_query(q, s, ql, sl_, rad, tie_order="reference")[:, :limits[l]]
            assert got.shape == ref.shape and (got == ref).all(), name
            differs_from_index_order += int((OF.oracle_batch_query(q, s, ql, sl_, rad)[:, :limits[l]] != ref).any(1).sum())
        if l < 3:
            pts, lens, r, dl = sp, sl, r * 2, dl * 2
    if recipe == "T8k":
        assert differs_from_index_order > 5000      # the order is far from the (d2, index) order on this pair


def _reorder_case(cuda, q, s, ql, sl, radius, cols):
    """CellGrid.query + KdForest.reorder on one table; -> (table, status)."""
    qd, sd = torch.from_numpy(q).to(cuda), torch.from_numpy(s).to(cuda)
    qld, sld = torch.from_numpy(np.asarray(ql, np.int32)).to(cuda), torch.from_numpy(np.asarray(sl, np.int32)).to(cuda)
    grid = ops.CellGrid(sd, sld, radius)
    idx, meta, counts, ties = grid.query(qd, qld, cols, want_ties=True)
    max_count, status, n_ties = meta.tolist()
    assert status == 0
    forest = ops.KdForest(sd, sld)
    st = forest.reorder(idx, qd, qld, 0, radius, max_count, rows=ties, nrows=n_ties, counts=counts)
    return idx[:, :max(max_count, 1)].cpu().numpy(), (0 if st is None else int(st.item())), n_ties


def test_random_lattice_clouds(cuda):
    """Coarse lattices: deep groups of equal distance, duplicates, 1-3 clouds of 0-400 supports."""
    rng = np.random.default_rng(5)
    seen_ties = 0
    for _ in range(60):
        nb = int(rng.integers(1, 4))
        sl = rng.integers(0, 400, nb).astype(np.int32)
        ql = rng.integers(1, 200, nb).astype(np.int32)
        if sl.sum() == 0:
            continue
        step = float(rng.choice([1 / 4, 1 / 8, 1 / 16, 1 / 64, 1 / 1024]))
        s = (np.round(rng.random((sl.sum(), 3)) / step) * step).astype(np.float32)
        q = (np.round(rng.random((ql.sum(), 3)) / step) * step).astype(np.float32)
        r = float(rng.choice([0.2, 0.35, 0.5]))
        try:
            want = OF.oracle_batch_query(q, s, ql, sl, r, tie_order="reference")
        except RuntimeError:
            continue
        cols = int(rng.choice([want.shape[1], max(1, want.shape[1] // 2), 7]))
        got, status, n_ties = _reorder_case(cuda, q, s, ql, sl, r, cols)
        seen_ties += n_ties
        assert status == 0
        w = want[:, :cols]
        assert got.shape == w.shape and (got == w).all()
    assert seen_ties > 1000


def test_large_nodes_and_long_rows(cuda):
    """Clouds above the 256-point wavefront-subtree limit (workgroup partition path), rows of hundreds of equal
    keys (introsort partition loop), thousands of duplicates of one point."""
    rng = np.random.default_rng(9)
    s = (rng.integers(0, 6, (8000, 3)) / 8).astype(np.float32)
    q = s[:300].copy()
    want = OF.oracle_batch_query(q, s, [300], [8000], 0.2, tie_order="reference")
    assert 300 < want.shape[1] < 1000                    # the cell-grid search stages at most 1024 hits per row
    got, status, _ = _reorder_case(cuda, q, s, [300], [8000], 0.2, want.shape[1])
    assert status == 0 and (got == want).all()
    s = np.zeros((800, 3), np.float32)
    s[:400, 0] = 0.25
    s = np.concatenate([s, rng.random((5000, 3)).astype(np.float32)])
    q = np.zeros((5, 3), np.float32)
    want = OF.oracle_batch_query(q, s, [5], [5800], 0.3, tie_order="reference")
    assert 800 < want.shape[1] < 1000
    got, status, _ = _reorder_case(cuda, q, s, [5], [5800], 0.3, want.shape[1])
    assert status == 0 and (got == want).all()


def test_rows_of_thousands_of_hits(cuda):
    """Tie rows far beyond the cell-grid search's 1024-entry staging list (the reference has no bound): the search
    keeps the `cols` nearest by streaming selection and reports the true count, the restore step stages the whole row
    (up to 8192 hits: one wavefront per workgroup with 128 KB of LDS) and replays std::sort on all of it."""
    rng = np.random.default_rng(11)
    s = (rng.integers(0, 12, (20000, 3)) / 16).astype(np.float32)       # ~12 duplicates per lattice site
    q = s[:40].copy()
    for radius, lo, hi in ((0.2, 1024, 2000), (0.3, 2500, 8192)):
        want = OF.oracle_batch_query(q, s, [40], [20000], radius, tie_order="reference")
        assert lo < want.shape[1] <= hi, want.shape
        qd, sd = torch.from_numpy(q).to(cuda), torch.from_numpy(s).to(cuda)
        qld, sld = torch.tensor([40], dtype=torch.int32, device=cuda), torch.tensor([20000], dtype=torch.int32, device=cuda)
        grid = ops.CellGrid(sd, sld, radius)
        for cols in (47, 300):
            idx, meta, counts, ties = grid.query(qd, qld, cols, want_ties=True)
            max_count, status, n_ties = meta.tolist()
            assert status == 0 and max_count == want.shape[1] and n_ties == 40
            st = ops.KdForest(sd, sld).reorder(idx, qd, qld, 0, radius, max_count, rows=ties, nrows=n_ties, counts=counts)
            assert (0 if st is None else int(st.item())) == 0
            assert (idx.cpu().numpy() == want[:, :cols]).all(), (radius, cols)


def test_skewed_tree(cuda):
    """Exponentially spaced coordinates: every midpoint split peels off a few points, so the tree is a long chain
    (about one level per binade) instead of a balanced one."""
    k = np.arange(0, 120)
    line = np.stack([2.0 ** -k, np.zeros_like(k, dtype=np.float64), np.zeros_like(k, dtype=np.float64)], 1)
    s = np.concatenate([line, line + [0, 2.0 ** -20, 0], line[:60] * [1, 0, 0] + [0, 0, 2.0 ** -21]] * 3).astype(np.float32)
    q = s[::7].copy()
    want = OF.oracle_batch_query(q, s, [len(q)], [len(s)], 0.01, tie_order="reference")
    got, status, n_ties = _reorder_case(cuda, q, s, [len(q)], [len(s)], 0.01, want.shape[1])
    assert status == 0 and n_ties > 0 and (got == want).all()


def test_more_tie_rows_than_a_workgroup_stages(cuda):
    """300 000 queries on 1024 workgroups = 293 rows per workgroup, nearly all of them with a tie: more than the 256
    a workgroup collects before it appends to the global list (the overflow goes to the list one by one)."""
    rng = np.random.default_rng(13)
    pts = (rng.integers(0, 96, (300000, 3)) / np.float32(64)).astype(np.float32)
    want = OF.oracle_batch_query(pts, pts, [300000], [300000], 0.03, tie_order="reference")
    cols = min(want.shape[1], 24)
    got, status, n_ties = _reorder_case(cuda, pts, pts, [300000], [300000], 0.03, cols)
    assert status == 0 and n_ties > 280000
    assert (got[:, :cols] == want[:, :cols]).all()
