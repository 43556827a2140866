"""CPU: the oracle's C front end (oracle/front_end.c) against the golden vectors generated from the
unmodified reference C++ (scripts/make_golden_frontend.py) and, when the prebuilt reference checker
library oracle/_ref is present, against the reference itself."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import frontend as OF
from pcrcg_amd import synthetic
from tests.tieutil import assert_tables_equal_mod_ties, canonicalise_table


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _stack(recipe, seed=0):
    src, tgt = synthetic.pair(recipe, seed)     # T8k: lattice-snapped, full of exactly equal distances + duplicates
    return np.concatenate([src, tgt]), np.array([len(src), len(tgt)], np.int32)


def test_umap_order_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "umap_order.npz"))
    for n in (1, 2, 13, 14, 29, 30, 500, 6000):
        assert (OF.oracle_umap_order(g[f"keys{n}"]) == g[f"order{n}"]).all(), n


def test_mini_pyramid_full_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "frontend_mini.npz"))
    pts, lens = _stack("mini")
    r, dl = 0.0625, 0.05
    for l in range(4):
        assert_tables_equal_mod_ties(OF.oracle_batch_query(pts, pts, lens, lens, r), g[f"conv{l}"], pts, pts)
        if l == 3:
            break
        sp, sl = OF.oracle_subsample_batch(pts, lens, dl)
        assert (sl == g[f"lens{l + 1}"]).all()
        assert (_bits(sp) == _bits(g[f"points{l + 1}"])).all()
        assert_tables_equal_mod_ties(OF.oracle_batch_query(sp, pts, sl, lens, r), g[f"pool{l}"], sp, pts)
        assert_tables_equal_mod_ties(OF.oracle_batch_query(pts, sp, lens, sl, 2 * r), g[f"up{l}"], pts, sp)
        pts, lens, r, dl = sp, sl, r * 2, dl * 2


@pytest.mark.parametrize("recipe", ["C1", "S30k", "T8k"])
def test_digests(golden_dir, recipe):
    dig = json.load(open(os.path.join(golden_dir, "frontend_digests.json")))[recipe]
    pts, lens = _stack(recipe)
    r, dl = 0.0625, 0.05
    for l in range(4):
        t = OF.oracle_batch_query(pts, pts, lens, lens, r)
        assert list(t.shape) == dig[f"conv{l}"]["shape"]
        assert _sha(canonicalise_table(t, pts, pts)[0]) == dig[f"conv{l}"]["sha256_canonical"]
        if l == 3:
            break
        sp, sl = OF.oracle_subsample_batch(pts, lens, dl)
        assert _sha(sp) == dig[f"points{l + 1}"]["sha256"] and _sha(sl) == dig[f"lens{l + 1}"]["sha256"]
        t = OF.oracle_batch_query(sp, pts, sl, lens, r)
        assert _sha(canonicalise_table(t, sp, pts)[0]) == dig[f"pool{l}"]["sha256_canonical"]
        t = OF.oracle_batch_query(pts, sp, lens, sl, 2 * r)
        assert _sha(canonicalise_table(t, pts, sp)[0]) == dig[f"up{l}"]["sha256_canonical"]
        pts, lens, r, dl = sp, sl, r * 2, dl * 2


def test_properties_and_errors():
    pts, lens = _stack("mini", 3)
    t = OF.oracle_batch_query(pts, pts, lens, lens, 0.07)
    assert (t[:, 0] == np.arange(len(pts))).all()             # self first
    assert t.max() == len(pts)                                # pad value = supports.size() (neighbors.cpp:324)
    sp, sl = OF.oracle_subsample_batch(pts, lens, 0.1)
    assert sl.sum() == len(sp)
    with pytest.raises(RuntimeError):
        OF.oracle_batch_query(pts[:, :2], pts, lens, lens, 0.1)
    with pytest.raises(RuntimeError):
        OF.oracle_batch_query(pts, pts, lens, lens[:1], 0.1)


@pytest.mark.skipif(not OF.have_ref(), reason="oracle/_ref (reference checker library) not built")
def test_against_reference_library():
    rng = np.random.RandomState(2)
    for n in (1, 13, 14, 700, 30000):
        k = np.unique(rng.randint(0, 1 << 50, size=2 * n).astype(np.uint64))
        rng.shuffle(k)
        assert (OF.oracle_umap_order(k[:n]) == OF.ref_umap_order(k[:n])).all()
    a = (rng.rand(900, 3) - 0.5).astype(np.float32) * 3
    b = (rng.rand(400, 3)).astype(np.float32) + 5
    pts, lens = np.concatenate([a, b]), np.array([900, 400], np.int32)
    for dl in (0.11, 0.4):
        rp, rl = OF.ref_subsample_batch(pts, lens, dl)
        op, ol = OF.oracle_subsample_batch(pts, lens, dl)
        assert (rl == ol).all() and (_bits(rp) == _bits(op)).all()
        assert_tables_equal_mod_ties(OF.oracle_batch_query(rp, pts, rl, lens, 2.5 * dl),
                                     OF.ref_batch_query(rp, pts, rl, lens, 2.5 * dl), rp, pts)
    rp, rl = OF.ref_subsample_batch(pts, lens, 0.11, max_p=50)
    op, ol = OF.oracle_subsample_batch(pts, lens, 0.11, max_p=50)
    assert (rl == ol).all() and (_bits(rp) == _bits(op)).all()


@pytest.mark.skipif(not OF.have_ref(), reason="oracle/_ref (reference checker library) not built")
def test_random_lattice_clouds_against_reference_library():
    """Property test (hypothesis): tiny clouds on coarse lattices -- duplicates, exact distance ties, points on
    cell boundaries, one-point clouds, negative coordinates -- oracle C front end vs the reference C++."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None, derandomize=True)
    @given(st.integers(1, 120), st.integers(1, 90), st.integers(2, 9), st.sampled_from([0.25, 0.5, 1.0, 1.7]),
           st.integers(0, 2 ** 31 - 1))
    def check(n0, n1, lattice, dl, seed):
        rng = np.random.RandomState(seed)
        pts = (rng.randint(-lattice, lattice + 1, size=(n0 + n1, 3)) / np.float32(4)).astype(np.float32)
        lens = np.array([n0, n1], np.int32)
        rp, rl = OF.ref_subsample_batch(pts, lens, dl)
        op, ol = OF.oracle_subsample_batch(pts, lens, dl)
        assert (rl == ol).all() and (_bits(rp) == _bits(op)).all()
        for q, s_, ql, sl, r in ((pts, pts, lens, lens, 1.2 * dl), (rp, pts, rl, lens, 1.2 * dl), (pts, rp, lens, rl, 2.4 * dl)):
            assert_tables_equal_mod_ties(OF.oracle_batch_query(q, s_, ql, sl, r), OF.ref_batch_query(q, s_, ql, sl, r), q, s_)

    check()


# ---- the reference's own order inside tie groups (nanoflann traversal + libstdc++ std::sort restated) ----------
def _ref_order_pyramid(recipe):
    pts, lens = _stack(recipe)
    r, dl = 0.0625, 0.05
    for l in range(4):
        yield f"conv{l}", OF.oracle_batch_query(pts, pts, lens, lens, r, tie_order="reference")
        if l == 3:
            break
        sp, sl = OF.oracle_subsample_batch(pts, lens, dl)
        yield f"pool{l}", OF.oracle_batch_query(sp, pts, sl, lens, r, tie_order="reference")
        yield f"up{l}", OF.oracle_batch_query(pts, sp, lens, sl, 2 * r, tie_order="reference")
        pts, lens, r, dl = sp, sl, r * 2, dl * 2


def test_reference_tie_order_mini_full_tables(golden_dir):
    g = np.load(os.path.join(golden_dir, "frontend_mini.npz"))
    for name, t in _ref_order_pyramid("mini"):
        assert t.shape == g[name].shape and (t == g[name]).all(), name


@pytest.mark.parametrize("recipe", ["C1", "S30k", "T8k"])
def test_reference_tie_order_digests(golden_dir, recipe):
    """Entry-for-entry equality with the tables the reference returns (T8k: 13 705 of 16 000 level-0 rows hold a tie)."""
    dig = json.load(open(os.path.join(golden_dir, "frontend_digests.json")))[recipe]
    for name, t in _ref_order_pyramid(recipe):
        assert list(t.shape) == dig[name]["shape"] and _sha(t) == dig[name]["sha256"], name


@pytest.mark.skipif(not OF.have_ref(), reason="oracle/_ref (reference checker library) not built")
def test_reference_tie_order_against_reference_library():
    """Coarse lattices (deep groups of equal distance, duplicates), 1-3 clouds, empty support clouds; plus two
    cases whose rows are long enough for the introsort partition loop and hold thousands of equal keys."""
    rng = np.random.default_rng(2)
    for _ in range(150):
        nb = int(rng.integers(1, 4))
        sl = rng.integers(0, 400, nb).astype(np.int32)
        ql = rng.integers(1, 200, nb).astype(np.int32)     # an EMPTY query cloud before a non-empty one is outside the
        if sl.sum() == 0:                                  # contract: the reference advances one cloud per query (:270)
            continue
        step = float(rng.choice([1 / 4, 1 / 8, 1 / 16, 1 / 64, 1 / 1024]))
        s = (np.round(rng.random((sl.sum(), 3)) / step) * step).astype(np.float32)
        q = (np.round(rng.random((ql.sum(), 3)) / step) * step).astype(np.float32)
        r = float(rng.choice([0.2, 0.35, 0.5, 1.0]))
        try:
            want = OF.ref_batch_query(q, s, ql, sl, r)
        except RuntimeError:
            with pytest.raises(RuntimeError):
                OF.oracle_batch_query(q, s, ql, sl, r, tie_order="reference")
            continue
        got = OF.oracle_batch_query(q, s, ql, sl, r, tie_order="reference")
        assert got.shape == want.shape and (got == want).all()
    s = np.zeros((5000, 3), np.float32)
    s[:2500, 0] = 0.25
    q = np.zeros((7, 3), np.float32)
    assert (OF.oracle_batch_query(q, s, [7], [5000], 1.0, tie_order="reference") == OF.ref_batch_query(q, s, [7], [5000], 1.0)).all()
    s = (rng.integers(0, 6, (20000, 3)) / 8).astype(np.float32)
    assert (OF.oracle_batch_query(s[:100], s, [100], [20000], 0.4, tie_order="reference")
            == OF.ref_batch_query(s[:100], s, [100], [20000], 0.4)).all()


def test_sanitizer_job_of_the_cpu_checkers():
    """`make -C oracle asan`: oracle/front_end.c (+ oracle/ref_shim.cpp over the unmodified reference C++ when
    /root/reference is present) under -fsanitize=address,undefined, driven by oracle/asan_driver.c over ragged, EMPTY,
    duplicate-ridden and dense clouds; with the reference linked the driver also checks restatement == reference entry for
    entry.  A sanitizer report or a mismatch fails the target.  (the reference has no sanitizer job;
    GPU AddressSanitizer is not available on the pool, so the CPU build is what can be instrumented.)"""
    import os
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-s", "-C", os.path.join(here, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "asan_driver ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
