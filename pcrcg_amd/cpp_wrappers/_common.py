import numpy as np
import torch


def as_tensor(a, dtype, err):
    """Mirror of PyArray_FROM_OTF(obj, NPY_FLOAT/NPY_INT, NPY_IN_ARRAY): any array-like is accepted and
    converted (RuntimeError(err) if that fails); torch tensors are used as they are.  Returns
    (tensor on its original device, was_host) -- shape checks happen before anything touches the GPU."""
    try:
        if isinstance(a, torch.Tensor):
            return a.to(dtype), not a.is_cuda
        np_dtype = np.float32 if dtype == torch.float32 else np.int32
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=np_dtype)), True
    except (TypeError, ValueError) as e:
        raise RuntimeError(err) from e


def to_device(t):
    return t.to("cuda").contiguous()
