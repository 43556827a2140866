"""Synthetic point-cloud recipes shared by tests, smoke() and bench.py.

All generators use ``np.random.RandomState(seed)`` and return float32 arrays.  The ``shell`` recipe
reproduces the neighbourhood statistics of a real 3DMatch fragment pair (points on the faces of a
cube, i.e. 2-D surfaces embedded in 3-D, with a little jitter).
"""
import numpy as np

# name -> (points per cloud, cube side [m], jitter [m])
RECIPES = {
    "C1": (5000, 0.75, 0.02),       # BASELINE.json configs[0]
    "S30k": (30000, 1.3, 0.02),     # BASELINE.json configs[1], "3DMatch-shaped"
    "mini": (1500, 0.45, 0.02),     # small parity case
    "T8k": (8000, 0.9, 0.02),       # tie-rich case: coordinates snapped to a 1/128 m lattice (see pair())
    "T30k": (30000, 1.3, 0.02),     # S30k snapped to a 1/256 m lattice: the bench's "voxelised scan" workload
}
LATTICE = {"T8k": 128.0, "T30k": 256.0}

# neighbourhood limits measured on the recipes with the reference's calibrate_neighbors formula
# (ref:datasets/dataloader.py:402-434); see scripts/make_golden_frontend.py
LIMITS = {
    "C1": [24, 37, 45, 48],
    "S30k": [43, 42, 47, 43],
    "K120k": [62, 58, 60, 60],
    "U30k": [29, 65, 75, 63],      # scripts/calib_u30k.py (pcrcg_amd.pyramid.calibrate_neighbors on the GPU)
}


def shell(rng, n, side, jitter):
    """n points on the six faces of a ``side``-cube plus uniform jitter."""
    face = rng.randint(0, 6, n)
    uv = rng.rand(n, 2).astype(np.float64) * side
    p = np.empty((n, 3), np.float64)
    axis = face // 2                       # the axis normal to the face
    level = (face % 2).astype(np.float64) * side
    for a in range(3):
        sel = axis == a
        o = [d for d in range(3) if d != a]
        p[sel, a] = level[sel]
        p[sel, o[0]] = uv[sel, 0]
        p[sel, o[1]] = uv[sel, 1]
    p += (rng.rand(n, 3) - 0.5) * jitter
    return p.astype(np.float32)


def pair(recipe="S30k", seed=0):
    """(src, tgt) float32 [n,3] clouds; src then tgt drawn from the same stream."""
    n, side, jitter = RECIPES[recipe]
    rng = np.random.RandomState(seed)
    src = shell(rng, n, side, jitter)
    tgt = shell(rng, n, side, jitter)
    if recipe in LATTICE:
        # Real scans (voxelised depth maps) are full of EXACTLY equal point distances and a few duplicate points;
        # uniform random floats have none.  Snapping to a lattice reproduces that: 13 705 of the 16 000 level-0
        # rows of T8k hold a tie group, 517 points are duplicates.
        q = np.float32(LATTICE[recipe])
        src = (np.round(src * q) / q).astype(np.float32)
        tgt = (np.round(tgt * q) / q).astype(np.float32)
    return src, tgt


def uniform_pair(n=30000, side=1.07, seed=0):
    """U30k: uniform-random points in a cube (the north_star's wording; SURVEY.md 8d "uniform-volume
    alternative"): denser neighbourhoods than a surface scan and an 8x decay per level."""
    rng = np.random.RandomState(seed)
    return (rng.rand(n, 3) * side).astype(np.float32), (rng.rand(n, 3) * side).astype(np.float32)


def slab_pair(n=120000, seed=0, extent=104.0, height=0.6):
    """K120k: KITTI-shaped outdoor slab (BASELINE.json configs[4])."""
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(2):
        p = rng.rand(n, 3)
        p[:, :2] *= extent
        p[:, 2] *= height
        out.append(p.astype(np.float32))
    return out[0], out[1]


def lomatch_pair(recipe="S30k", seed=0, overlap=0.2):
    """3DLoMatch-shaped pair (BASELINE.json configs[2]): tgt = R*(subset of src) + t plus fresh
    points so that roughly ``overlap`` of the points coincide.  Returns src, tgt, rot, trans."""
    n, side, jitter = RECIPES[recipe]
    rng = np.random.RandomState(seed)
    src = shell(rng, n, side, jitter)
    k = int(n * overlap)
    keep = rng.permutation(n)[:k]
    fresh = shell(rng, n - k, side, jitter) + np.float32(side * 0.8)
    ang = rng.rand(3) * 2 * np.pi
    cz, sz, cy, sy, cx, sx = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    rot = Rz @ Ry @ Rx
    trans = rng.rand(3, 1) - 0.5
    tgt_src = np.concatenate([src[keep], fresh], 0).astype(np.float64)
    tgt = (rot @ tgt_src.T + trans).T.astype(np.float32)
    return src, tgt, rot.astype(np.float32), trans.astype(np.float32)


def image_inputs(n_src, n_tgt, seed=0, img_num=2, h=120, w=160, frac=0.45, channels=128):
    """Synthetic stand-ins for what PCR-CG's 2-D branch hands to KPFCNN.forward (ref:models/architectures.py:195-514,
    ref:datasets/indoor.py:192-829): per cloud and image a [channels, h, w] feature map (the ResUNet's output shape for the
    3DMatch frames: 128 x 120 x 160), the projected points' pixel coordinates `inds2d` [k, 2] (column, row) and point
    indices `inds3d` [k] -- a random `frac` of the cloud per image, overlapping between images, so the write order matters
    -- and, for img_num < 3, a [w, h] valid mask.  Keys as in the reference's batch dict, with the maps under
    '{side}{i}_feature2d' (what pcrcg_amd.KPFCNN takes in place of a backbone).  numpy arrays, seeded: the same inputs in the
    fixture generator (scripts/make_golden_image_s30k.py), the tests and bench.py."""
    rng = np.random.RandomState(1000 + seed)
    out = {}
    for side, n in (("src", n_src), ("tgt", n_tgt)):
        for i in range(1, img_num + 1):
            k = int(n * frac)
            out[f"{side}{i}_feature2d"] = rng.rand(channels, h, w).astype(np.float32)
            out[f"{side}{i}_inds3d"] = rng.permutation(n)[:k].astype(np.int64)
            out[f"{side}{i}_inds2d"] = np.stack([rng.randint(0, w, k), rng.randint(0, h, k)], 1).astype(np.int64)
            if img_num < 3:
                out[f"{side}_valid_map{i}"] = (rng.rand(w, h) > 0.2).astype(np.float32)
    return out
