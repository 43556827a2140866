"""MI355X replacement for the reference extension module ``grid_subsampling``
(zip:cpp_subsampling/wrapper.cpp:28-33: methods ``subsample`` and ``subsample_batch``).

Only the points-only form is on the hot path (ref:datasets/dataloader.py:18-24); the optional
``features`` / ``classes`` inputs of the reference raise RuntimeError here."""
import numpy as np
import torch

from ... import ops
from .._common import as_tensor, to_device

_METHODS = ("barycenters", "voxelcenters")


def _finish(points, lengths, was_numpy):
    if was_numpy:
        return points.cpu().numpy(), lengths.cpu().numpy()
    return points, lengths


def subsample_batch(points, batches, *, features=None, classes=None, sampleDl=0.1, method="barycenters", max_p=0,
                    verbose=0):
    """(points f32 [N,3], batches i32 [B]) -> (sub_points f32 [M,3], sub_batches i32 [B])
    (zip:cpp_subsampling/wrapper.cpp:62-330).  numpy in -> numpy out; device tensors in -> device
    tensors out."""
    if method not in _METHODS:  # wrapper.cpp:92-96
        raise RuntimeError('Error parsing method. Valid method names are "barycenters" and "voxelcenters" ')
    if features is not None or classes is not None:
        raise RuntimeError("pcrcg_amd: subsample_batch with features/classes is outside the KPFCNN hot path")
    p, was_numpy = as_tensor(points, torch.float32, "Error converting input points to numpy arrays of type float32")
    b, _ = as_tensor(batches, torch.int32, "Error converting input batches to numpy arrays of type int32")
    if p.dim() != 2 or p.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : points.shape is not (N, 3)")
    if b.dim() > 1:
        raise RuntimeError("Wrong dimensions : batches.shape is not (B,) ")
    sub, sub_len = ops.grid_subsample(to_device(p), to_device(b), float(sampleDl), int(max_p))
    if sub.shape[0] < 1:  # wrapper.cpp:266-270
        raise RuntimeError("Error")
    return _finish(sub, sub_len, was_numpy)


def subsample(points, *, features=None, classes=None, sampleDl=0.1, method="barycenters", verbose=0):
    """Single-cloud variant (zip:cpp_subsampling/wrapper.cpp:338-565): returns the sub-sampled points."""
    if method not in _METHODS:
        raise RuntimeError('Error parsing method. Valid method names are "barycenters" and "voxelcenters" ')
    if features is not None or classes is not None:
        raise RuntimeError("pcrcg_amd: subsample with features/classes is outside the KPFCNN hot path")
    p, was_numpy = as_tensor(points, torch.float32, "Error converting input points to numpy arrays of type float32")
    if p.dim() != 2 or p.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : points.shape is not (N, 3)")
    p = to_device(p)
    b = torch.tensor([p.shape[0]], dtype=torch.int32, device=p.device)
    sub, _ = ops.grid_subsample(p, b, float(sampleDl), 0)
    if sub.shape[0] < 1:
        raise RuntimeError("Error")
    return sub.cpu().numpy() if was_numpy else sub
