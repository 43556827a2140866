"""MI355X replacement for the reference extension module ``radius_neighbors``
(ref:cpp_wrappers/cpp_neighbors/wrapper.cpp:25-29: method ``batch_query``)."""
import torch

from ... import ops
from .._common import as_tensor, to_device

_FIRST_GUESS_COLS = 128


def batch_query(queries, supports, q_batches, s_batches, *, radius=0.1, tie_order="auto"):
    """-> int32 [Nq, max_count]: per query the supports of the same batch element within `radius`,
    ascending distance, padded with the total support count
    (ref:cpp_wrappers/cpp_neighbors/wrapper.cpp:58-238, neighbors.cpp:211-333).
    numpy in -> numpy out; device tensors in -> device tensor out.
    tie_order "auto" (default): supports at EXACTLY equal distance come in the reference's own order (nanoflann
    traversal + std::sort replayed by csrc/tieorder.hip), so the table equals the reference's entry for entry;
    "index": ascending index inside such groups (no KD-forest)."""
    q, was_numpy = as_tensor(queries, torch.float32, "Error converting query points to numpy arrays of type float32")
    s, _ = as_tensor(supports, torch.float32, "Error converting support points to numpy arrays of type float32")
    qb, _ = as_tensor(q_batches, torch.int32, "Error converting query batches to numpy arrays of type int32")
    sb, _ = as_tensor(s_batches, torch.int32, "Error converting support batches to numpy arrays of type int32")
    if q.dim() != 2 or q.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : query.shape is not (N, 3)")
    if s.dim() != 2 or s.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : support.shape is not (N, 3)")
    if qb.dim() > 1:
        raise RuntimeError("Wrong dimensions : queries_batches.shape is not (B,) ")
    if sb.dim() > 1:
        raise RuntimeError("Wrong dimensions : supports_batches.shape is not (B,) ")
    if qb.shape[0] != sb.shape[0]:
        raise RuntimeError("Wrong number of batch elements: different for queries and supports ")
    q, s, qb, sb = to_device(q), to_device(s), to_device(qb), to_device(sb)
    grid = ops.CellGrid(s, sb, float(radius))
    if tie_order not in ("auto", "index"):
        raise ValueError("pcrcg_amd: tie_order must be 'auto' or 'index'")
    want_ties = tie_order == "auto"
    # the queries walk a cell grid of their own (the supports' when the call is a self query): the cell-cooperative,
    # LDS-staged search (csrc/radius.hip: k_radius_cells)
    same = q.data_ptr() == s.data_ptr() and q.shape == s.shape and torch.equal(qb, sb)
    qgrid = grid if same else (ops.CellGrid(q, qb, float(radius)) if q.shape[0] > 0 else None)
    res = grid.query(q, qb, _FIRST_GUESS_COLS, want_ties=want_ties, query_grid=qgrid)
    max_count, status, n_ties = (int(v) for v in res[1].tolist())
    if max_count > _FIRST_GUESS_COLS and status == 0:
        res = grid.query(q, qb, max_count, want_ties=want_ties, query_grid=qgrid)
        max_count, status, n_ties = (int(v) for v in res[1].tolist())
    if status != 0:
        raise RuntimeError("pcrcg_amd: radius search capacity exceeded (status %d)" % status)
    if q.shape[0] * max_count < 1:  # wrapper.cpp:201-205
        raise RuntimeError("Error")
    idx = res[0]
    if want_ties and n_ties > 0:
        st = ops.KdForest(s, sb).reorder(idx, q, qb, 0, float(radius), max_count, rows=res[3], nrows=n_ties, counts=res[2])
        if int(st.item()) != 0:
            raise RuntimeError("pcrcg_amd: restoring the reference's tie order failed (status %d)" % int(st.item()))
    out = idx[:, :max_count].to(torch.int32)
    return out.cpu().numpy() if was_numpy else out.contiguous()
