"""Ground-truth correspondences on the device (mirror of get_correspondences,
ref:lib/benchmark_utils.py:121-134, which loops over the source points in Python and asks an open3d
KD-tree for the target points within `search_voxel_size` of each).

Same result contract: an int64 [K,2] tensor of (src index, tgt index), source-major, the targets of one
source point ordered by increasing distance (FLANN returns radius-search hits sorted), optionally the K
nearest only.  open3d holds points and the 4x4 transform in float64, so membership `d < radius` is decided in
float64 here too: the fp32 cell-grid radius kernel of the hot path (pcrcg_radius_query) produces the candidates
with a slightly inflated radius, and candidates are then re-measured and re-ordered in float64 (a few
elementwise ops on an [N, cols] table).  Equal distances are ordered by target index."""
import torch

from . import ops

_INFLATE = 1.0 + 1e-4     # fp32 candidate radius: never loses a pair that is inside in float64


def get_correspondences(src_pcd, tgt_pcd, trans, search_voxel_size, K=None):
    """src_pcd [N,3], tgt_pcd [M,3] float32 device tensors, trans [4,4] (any float dtype, host or device)."""
    if not (isinstance(src_pcd, torch.Tensor) and src_pcd.is_cuda and tgt_pcd.is_cuda):
        raise RuntimeError("pcrcg_amd.get_correspondences: point clouds must be tensors on a HIP device")
    dev = src_pcd.device
    n, m = src_pcd.shape[0], tgt_pcd.shape[0]
    if n == 0 or m == 0:
        return torch.empty((0, 2), dtype=torch.int64, device=dev)
    t64 = torch.as_tensor(trans, dtype=torch.float64).to(dev)
    src64 = src_pcd.double() @ t64[:3, :3].t() + t64[:3, 3]
    tgt64 = tgt_pcd.double()
    moved = src64.float().contiguous()
    radius = float(search_voxel_size)
    grid = ops.CellGrid(tgt_pcd.float().contiguous(), torch.tensor([m], dtype=torch.int32, device=dev), radius * _INFLATE)
    q_len = torch.tensor([n], dtype=torch.int32, device=dev)
    cols = 32
    while True:
        idx, meta = grid.query(moved, q_len, cols)
        max_count = int(meta[0].item())
        if max_count <= cols:
            break
        cols = max_count
    idx = idx[:, :max(max_count, 1)]
    real = idx < m                                             # shadow entries are == m
    d = (tgt64[idx.clamp(max=m - 1)] - src64[:, None, :]).pow(2).sum(-1).sqrt()
    d = torch.where(real & (d < radius), d, torch.full_like(d, float("inf")))
    # order by (float64 distance, target index); torch.sort is stable on request
    by_idx = torch.argsort(idx, dim=1, stable=True)
    d, idx = torch.gather(d, 1, by_idx), torch.gather(idx, 1, by_idx)
    by_d = torch.argsort(d, dim=1, stable=True)
    d, idx = torch.gather(d, 1, by_d), torch.gather(idx, 1, by_d)
    if K is not None:
        d, idx = d[:, :K], idx[:, :K]
    hit = torch.nonzero(torch.isfinite(d))                     # row-major: source-major, distance order
    return torch.stack([hit[:, 0], idx[hit[:, 0], hit[:, 1]]], dim=1)
