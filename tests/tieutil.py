"""Helpers for comparing radius-neighbour tables modulo the order inside groups of exactly equal
squared distance (the reference's order there is an artefact of KD-tree traversal + introsort,
SURVEY.md 8a-2; the oracle and the HIP path use ascending index)."""
import numpy as np


def row_d2(table, queries, supports):
    """fp32 squared distances exactly as nanoflann's L2_Simple_Adaptor evaluates them
    (zip:cpp_utils/nanoflann/nanoflann.hpp:432-440): ((dx*dx)+dy*dy)+dz*dz, each op rounded."""
    ns = supports.shape[0]
    sp = np.concatenate([supports, np.full((1, 3), np.inf, np.float32)]).astype(np.float32)
    d = queries[:, None, :].astype(np.float32) - sp[np.minimum(table, ns)]
    d2 = (d[..., 0] * d[..., 0]).astype(np.float32)
    d2 = (d2 + (d[..., 1] * d[..., 1]).astype(np.float32)).astype(np.float32)
    d2 = (d2 + (d[..., 2] * d[..., 2]).astype(np.float32)).astype(np.float32)
    return d2


def canonicalise_table(table, queries, supports):
    """Re-sort every row by (d2, index); returns (canonical table, number of rows that changed)."""
    with np.errstate(invalid="ignore"):
        d2 = row_d2(table, queries, supports)
    d2 = np.where(table >= supports.shape[0], np.float32(np.inf), d2)
    order = np.lexsort((table, d2), axis=1)
    canon = np.take_along_axis(table, order, 1)
    return canon, int((canon != table).any(1).sum())


def assert_tables_equal_mod_ties(a, b, queries, supports):
    """a, b: [Nq, cols] index tables over the same (queries, supports).  They must have the same
    shape, every row must be sorted by d2, and they may differ only by a permutation inside runs of
    exactly equal d2."""
    assert a.shape == b.shape, (a.shape, b.shape)
    ca, _ = canonicalise_table(a, queries, supports)
    cb, _ = canonicalise_table(b, queries, supports)
    bad = (ca != cb).any(1)
    assert not bad.any(), f"{int(bad.sum())} rows differ beyond tie order, first row {int(np.argmax(bad))}"
    with np.errstate(invalid="ignore"):
        for t in (a, b):
            d2 = row_d2(t, queries, supports)
            d2 = np.where(t >= supports.shape[0], np.float32(np.inf), d2)
            assert (np.diff(d2, axis=1) >= 0).all() or np.isinf(d2).any(), "row not sorted by distance"
            finite_ok = (d2[:, 1:] >= d2[:, :-1]) | np.isinf(d2[:, 1:])
            assert finite_ok.all(), "row not sorted by distance"
