"""Kernel-point dispositions for KPConv (ctor-time only).

The reference reads a pre-optimised 15-point disposition from a .ply file, adds N(0, 0.01) noise,
scales it by the convolution radius and applies a random rotation about z
(ref:kernels/kernel_points.py:388-470, called from KPConv.init_KP ref:models/blocks.py:214-227).
The result is stored in the state_dict (``...KPConv.kernel_points``), so trained or reference-
initialised models carry their own kernel points and never call this module.

For models created from scratch we use a closed-form disposition with the same structure (one centre
point + 14 points at 0.66 of the radius: the 6 axis and 8 diagonal directions of a cube) and follow
the reference's noise / scale / rotation recipe, consuming ``np.random`` in the same order."""
import numpy as np

_RATIO = 0.66


def base_disposition(num_kpoints=15):
    if num_kpoints != 15:
        raise ValueError("pcrcg_amd ships the 15-point 'center' disposition only")
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float64)
    diag = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float64) / np.sqrt(3.0)
    return np.concatenate([np.zeros((1, 3)), axes * _RATIO, diag * _RATIO], 0)


def load_kernels(radius, num_kpoints=15, dimension=3, fixed="center"):
    """-> float32 [num_kpoints, 3] (cf. ref:kernels/kernel_points.py:388-470)."""
    if dimension != 3 or fixed != "center":
        raise ValueError("pcrcg_amd supports 3-D 'center' kernel dispositions only")
    pts = base_disposition(num_kpoints)
    theta = np.random.rand() * 2 * np.pi
    c, s = np.cos(theta), np.sin(theta)
    rot = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)
    pts = pts + np.random.normal(scale=0.01, size=pts.shape)
    pts = radius * pts
    pts = np.matmul(pts, rot)
    return pts.astype(np.float32)
