"""Kernel-point dispositions for KPConv (ctor-time only).

The reference reads a pre-optimised 15-point disposition from a data file
(``kernels/dispositions/k_015_center_3D.ply``: 15 float64 xyz rows), draws a random rotation angle
about z, adds N(0, 0.01) noise, scales by the convolution radius and rotates
(ref:kernels/kernel_points.py:388-470, called from KPConv.init_KP ref:models/blocks.py:214-227).
The result is stored in the state_dict (``...KPConv.kernel_points``).

``DISPOSITION_15_CENTER_3D`` below is that data table, value for value (float64, written as hex
floats so that no decimal rounding is involved).  ``load_kernels`` consumes ``np.random`` in the
reference's order (one ``rand()`` for the angle, then ``normal(size=(15, 3))``), so a model built
here under ``np.random.seed(s)`` carries bit-identical ``kernel_points`` to a reference model built
under the same seed (tests/test_host_logic.py against tests/golden/model_mini.pt)."""
import numpy as np

_HEX_15_CENTER_3D = (
    ("0x0.0p+0", "0x0.0p+0", "0x0.0p+0"),
    ("-0x1.fe29bf0cfaceap-2", "0x1.ac4e701f952c0p-2", "0x1.e0bc6956f7dc0p-4"),
    ("-0x1.ee0cf4b870cafp-3", "-0x1.5e5a132fa3822p-2", "-0x1.05e9a1d52e9eap-1"),
    ("-0x1.21ab812f601ffp-2", "-0x1.2c1ae3ffc763ep-1", "0x1.d9385f834b457p-4"),
    ("0x1.298369c9ed7bcp-2", "-0x1.9d6af804cac5dp-4", "-0x1.2b910c3b9ce62p-1"),
    ("0x1.b67a29c476231p-2", "0x1.98e1cae520d45p-2", "-0x1.3a2e886d17a45p-2"),
    ("-0x1.4590167bb1652p-1", "-0x1.4fb9e92a055c4p-4", "-0x1.49880d4f52be4p-3"),
    ("-0x1.ba2c9d8980332p-2", "-0x1.2da8905e0c182p-3", "0x1.e9c9fc344c9d4p-2"),
    ("-0x1.6de76507150dfp-5", "0x1.1e721a3b2ee5dp-2", "0x1.31c888d5dfe46p-1"),
    ("0x1.cddf9dbd72189p-3", "-0x1.60e57d79d85c4p-2", "0x1.04119351580ccp-1"),
    ("0x1.471cde4ce8843p-1", "-0x1.5a6ad26aa8478p-3", "-0x1.85f981e73eff6p-7"),
    ("-0x1.cddf9ad90e0efp-3", "0x1.60e57e87e63a7p-2", "-0x1.04119347f5f60p-1"),
    ("0x1.f651dcc955727p-2", "0x1.1342269385a22p-2", "0x1.68a5094cf1994p-2"),
    ("0x1.026303e299f34p-2", "-0x1.31b2b425ca407p-1", "-0x1.093d48b6b8947p-3"),
    ("0x1.17c9ffcfd5bc8p-5", "0x1.5131d85628b38p-1", "0x1.71c891818c4cdp-5"),
)
DISPOSITION_15_CENTER_3D = np.array([[float.fromhex(v) for v in row] for row in _HEX_15_CENTER_3D], np.float64)
DISPOSITION_15_CENTER_3D.setflags(write=False)


def base_disposition(num_kpoints=15):
    if num_kpoints != 15:
        raise ValueError("pcrcg_amd ships the 15-point 'center' disposition only")
    return DISPOSITION_15_CENTER_3D.copy()


def load_kernels(radius, num_kpoints=15, dimension=3, fixed="center"):
    """-> float32 [num_kpoints, 3] (ref:kernels/kernel_points.py:388-470: angle, noise, scale, rotate)."""
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
