"""Import the UNMODIFIED Python reference (/root/reference) in the build container.

Only golden-vector generators under scripts/ use this; nothing in tests/, bench.py, smoke() or the
product imports it, and /root/reference does not exist on the GPU box.  The reference's modules
import absent third-party packages at module top (open3d, cv2, torchvision, easydict,
tensorboardX) and two CPython extensions that no longer build against NumPy 2.x, so those names
are pre-seeded in ``sys.modules`` with inert stubs; the extension stubs forward to
oracle/_ref/libpcrcg_ref.so (the reference's own C++ cores, see oracle/ref_shim.cpp).
"""
import os
import sys
import types

REF = "/root/reference"
sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


class AttrDict(dict):
    """Stand-in for easydict.EasyDict (attribute access on a flat dict)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def setup():
    from oracle import frontend as F
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.chdir(REF)  # kernels/dispositions is a relative path (ref:kernels/kernel_points.py:391)

    def subsample_batch(points, batches, features=None, classes=None, sampleDl=0.1,
                        method="barycenters", max_p=0, verbose=0):
        import numpy as np
        return F.ref_subsample_batch(np.asarray(points), np.asarray(batches), sampleDl=sampleDl, max_p=max_p)

    def batch_query(queries, supports, q_batches, s_batches, radius=0.1):
        import numpy as np
        return F.ref_batch_query(np.asarray(queries), np.asarray(supports), np.asarray(q_batches),
                                 np.asarray(s_batches), radius=radius)

    _stub("cpp_wrappers")
    _stub("cpp_wrappers.cpp_subsampling")
    _stub("cpp_wrappers.cpp_neighbors")
    _stub("cpp_wrappers.cpp_subsampling.grid_subsampling", subsample_batch=subsample_batch)
    _stub("cpp_wrappers.cpp_neighbors.radius_neighbors", batch_query=batch_query)
    _stub("datasets.indoor", IndoorDataset=object)
    _stub("datasets.kitti", KITTIDataset=object)
    _stub("datasets.modelnet", get_train_datasets=None, get_test_datasets=None)
    _stub("easydict", EasyDict=AttrDict)
    for name in ("open3d", "cv2", "torchvision", "torchvision.transforms", "tensorboardX", "coloredlogs",
                 "nibabel", "h5py"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name)
    return F


def kitti_config(**over):
    """Flattened configs/test/kitti.yaml (ref:lib/utils.py:46-65), geometry-only like indoor_config().  The reference's
    kitti.yaml predates PCR-CG's image branch and lacks the five keys KPFCNN.__init__ reads for it
    (ref:models/architectures.py:48-52); they are supplied here with their "branch off" values."""
    base = dict(img_num=0, init_mode="", node_overlap=False, quaternion=False)
    base.update(over)
    return indoor_config(_yaml="configs/test/kitti.yaml", **base)


def indoor_config(_yaml="configs/test/indoor.yaml", **over):
    """Flattened configs/test/indoor.yaml (ref:lib/utils.py:46-65) with the geometry-only overrides
    named in BASELINE.json configs[0] (image_feature False, in_feats_dim 1)."""
    import yaml
    with open(os.path.join(REF, _yaml)) as f:
        cfg = yaml.safe_load(f)
    flat = AttrDict()
    for _, v in cfg.items():
        flat.update(v)
    flat["image_feature"] = False
    flat["in_feats_dim"] = 1
    flat.update(over)
    from configs.models import architectures
    flat["architecture"] = architectures[flat["dataset"]]
    return flat
