"""Ground-truth correspondences on the device (mirror of get_correspondences,
ref:lib/benchmark_utils.py:121-134, which loops over the source points in Python and asks an open3d
KD-tree for the target points within `search_voxel_size` of each).

Same result contract: an int64 [K,2] tensor of (src index, tgt index), source-major, the targets of one
source point ordered by increasing distance (FLANN returns radius-search hits sorted), optionally the K
nearest only.  open3d holds points and the 4x4 transform in float64, so membership `d < radius` is decided in
float64 here too: the cell grid of the hot path's radius search (fp32, built with a slightly inflated radius)
supplies the candidates, and the kernel re-measures every candidate in float64 from the float64-moved source point,
ranks the hits by (distance, target index) and writes them out (csrc/radius.hip: k_correspond_rows / _emit).  Two
launches and one scan; the host reads two integers (longest list, number of pairs) to size the result.
Equal distances are ordered by target index."""
import ctypes

import numpy as np
import torch

from . import _lib, ops

_INFLATE = 1.0 + 1e-4     # fp32 candidate radius: never loses a pair that is inside in float64
_ROW_CAP = 1024           # hits one source point can stage (csrc/radius.hip kCorrCap)


def get_correspondences(src_pcd, tgt_pcd, trans, search_voxel_size, K=None):
    """src_pcd [N,3], tgt_pcd [M,3] float32 device tensors, trans [4,4] (any float dtype, host or device)."""
    if not (isinstance(src_pcd, torch.Tensor) and src_pcd.is_cuda and tgt_pcd.is_cuda):
        raise RuntimeError("pcrcg_amd.get_correspondences: point clouds must be tensors on a HIP device")
    dev = src_pcd.device
    n, m = src_pcd.shape[0], tgt_pcd.shape[0]
    if n == 0 or m == 0:
        return torch.empty((0, 2), dtype=torch.int64, device=dev)
    L = _lib.lib()
    t64 = np.ascontiguousarray(torch.as_tensor(trans, dtype=torch.float64).cpu().numpy().reshape(4, 4))
    src = src_pcd.float().contiguous()
    radius = float(search_voxel_size)
    grid = ops.CellGrid(tgt_pcd.float().contiguous(), torch.tensor([m], dtype=torch.int32, device=dev), radius * _INFLATE)
    keep = int(K) if K is not None else 0
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    cols = 32
    while True:
        stage = torch.empty((n, cols), dtype=torch.int32, device=dev)
        head = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.pcrcg_correspondences_rows(src.data_ptr(), n, t64.ctypes.data_as(ctypes.c_void_p), radius, keep, m,
                                                grid.grid.data_ptr(), cols, stage.data_ptr(), counts.data_ptr(),
                                                head.data_ptr(), stream), "pcrcg_correspondences_rows")
        ends = torch.cumsum(counts, 0, dtype=torch.int64)
        longest, total = (int(v) for v in torch.stack([head[0].to(torch.int64), ends[-1]]).tolist())   # ONE read-back
        if longest > _ROW_CAP:
            raise RuntimeError(f"pcrcg_amd.get_correspondences: a source point has {longest} targets within the radius "
                               f"(more than the {_ROW_CAP} a row can stage)")
        need = min(longest, keep) if keep else longest
        if need <= cols:
            break
        cols = need
    out = torch.empty((total, 2), dtype=torch.int64, device=dev)
    if total:
        offsets = ends - counts
        _lib.check(L.pcrcg_correspondences_emit(stage.data_ptr(), cols, counts.data_ptr(), offsets.data_ptr(), n,
                                                out.data_ptr(), stream), "pcrcg_correspondences_emit")
    return out
