"""ORACLE (test infrastructure only -- never imported by the product): get_correspondences restated.

ref:lib/benchmark_utils.py:121-134 builds an open3d KD-tree over the target cloud and, for every source
point moved by `trans`, appends (i, j) for the hits of search_radius_vector_3d(point, radius).  open3d and
its FLANN back end are third-party dependencies absent from /root/reference and from this image
(requirements.txt pins open3d==0.10.0.0), so parity is UNPINNED against a reference run; the restatement
follows the published semantics: points and transform in float64, hits are the targets with Euclidean
distance strictly below the radius (FLANN compares squared distances: d2 < r2), sorted by increasing
distance.  Ties in distance are ordered by index here."""
import numpy as np


def get_correspondences(src, tgt, trans, radius, K=None, chunk=512):
    src = np.asarray(src, np.float64)
    tgt = np.asarray(tgt, np.float64)
    trans = np.asarray(trans, np.float64)
    moved = src @ trans[:3, :3].T + trans[:3, 3]
    out = []
    for s in range(0, len(moved), chunk):
        blk = moved[s:s + chunk]
        d = np.sqrt(((blk[:, None, :] - tgt[None, :, :]) ** 2).sum(-1))
        for r in range(len(blk)):
            js = np.nonzero(d[r] < radius)[0]
            js = js[np.argsort(d[r, js], kind="stable")]
            if K is not None:
                js = js[:K]
            out.extend((s + r, j) for j in js)
    return np.asarray(out, np.int64).reshape(-1, 2)
