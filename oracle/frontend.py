"""ctypes bindings of the two CPU checker libraries (TEST INFRASTRUCTURE ONLY).

* ``oracle_*``  -> oracle/_build/libpcrcg_oracle.so, this repo's plain-C restatement
                   (oracle/front_end.c) of the reference's native front end.
* ``ref_*``     -> oracle/_ref/libpcrcg_ref.so, the UNMODIFIED reference C++
                   (ref:cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-333,
                   zip:cpp_subsampling/grid_subsampling/grid_subsampling.cpp:109-211) behind
                   oracle/ref_shim.cpp; built in the build container only, travels prebuilt.

Signatures mirror the reference's Python extension modules
(ref:cpp_wrappers/cpp_neighbors/wrapper.cpp:58-75, zip:cpp_subsampling/wrapper.cpp:62-82):
``batch_query(queries, supports, q_batches, s_batches, radius=) -> int32 [Nq, max_count]`` and
``subsample_batch(points, batches, sampleDl=, max_p=) -> (float32 [M,3], int32 [B])``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "_build", "libpcrcg_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libpcrcg_ref.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)
_u64p = ctypes.POINTER(ctypes.c_uint64)


def build(ref=True):
    """Compile the checker libraries (gcc/g++ only).  ``ref`` also rebuilds oracle/_ref when
    /root/reference is present; otherwise a prebuilt oracle/_ref is kept."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"] + (["ref"] if ref else []))


def _load(path, what):
    if not os.path.exists(path):
        raise RuntimeError(f"{what} not built: {path} (run `make -C oracle`)")
    return ctypes.CDLL(path)


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(_ORACLE_SO):
            build(ref=False)
        lib = _load(_ORACLE_SO, "oracle library")
        lib.oracle_umap_order.argtypes = [_u64p, ctypes.c_int, _i32p]
        lib.oracle_umap_order.restype = ctypes.c_int
        lib.oracle_grid_subsample_batch.argtypes = [_f32p, ctypes.c_int, _i32p, ctypes.c_int,
                                                    ctypes.c_float, ctypes.c_int, _f32p, _i32p]
        lib.oracle_grid_subsample_batch.restype = ctypes.c_int
        lib.oracle_radius_neighbors_batch.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, _i32p,
                                                      _i32p, ctypes.c_int, ctypes.c_float, _i32p]
        lib.oracle_radius_neighbors_batch.restype = ctypes.c_void_p
        lib.oracle_radius_neighbors_batch_reforder.argtypes = lib.oracle_radius_neighbors_batch.argtypes
        lib.oracle_radius_neighbors_batch_reforder.restype = ctypes.c_void_p
        lib.oracle_free.argtypes = [ctypes.c_void_p]
        _oracle = lib
    return _oracle


def have_ref():
    return os.path.exists(_REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        lib = _load(_REF_SO, "reference library")
        lib.ref_batch_query.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, _i32p, _i32p,
                                        ctypes.c_int, ctypes.c_float, _i32p]
        lib.ref_batch_query.restype = ctypes.c_void_p
        lib.ref_subsample_batch.argtypes = [_f32p, ctypes.c_int, _i32p, ctypes.c_int, ctypes.c_float,
                                            ctypes.c_int, _i32p, _i32p]
        lib.ref_subsample_batch.restype = ctypes.c_void_p
        lib.ref_umap_order.argtypes = [_u64p, ctypes.c_int, _i32p]
        lib.ref_free.argtypes = [ctypes.c_void_p]
        _ref = lib
    return _ref


def _f32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


def _pts(a, what):
    a = _f32(a)
    if a.ndim != 2 or a.shape[1] != 3:
        raise RuntimeError(f"Wrong dimensions : {what}.shape is not (N, 3)")
    return a


def _query(fn, free, queries, supports, q_batches, s_batches, radius):
    q, s = _pts(queries, "query"), _pts(supports, "support")
    qb, sb = _i32(q_batches), _i32(s_batches)
    if qb.shape[0] != sb.shape[0]:
        raise RuntimeError("Wrong number of batch elements: different for queries and supports ")
    cols = ctypes.c_int(0)
    ptr = fn(q.ctypes.data_as(_f32p), q.shape[0], s.ctypes.data_as(_f32p), s.shape[0],
             qb.ctypes.data_as(_i32p), sb.ctypes.data_as(_i32p), qb.shape[0],
             ctypes.c_float(radius), ctypes.byref(cols))
    if not ptr or cols.value < 1:
        raise RuntimeError("Error")
    n = q.shape[0] * cols.value
    out = np.ctypeslib.as_array(ctypes.cast(ptr, _i32p), shape=(n,)).copy().reshape(q.shape[0], cols.value)
    free(ptr)
    return out


def oracle_batch_query(queries, supports, q_batches, s_batches, radius=0.1, tie_order="index"):
    """tie_order "index": equal distances in ascending index order (the order the HIP path defines);
    "reference": the reference's own order (nanoflann 1.3.0 traversal + libstdc++ std::sort restated)."""
    lib = oracle_lib()
    fn = {"index": lib.oracle_radius_neighbors_batch, "reference": lib.oracle_radius_neighbors_batch_reforder}[tie_order]
    return _query(fn, lib.oracle_free, queries, supports, q_batches, s_batches, radius)


def ref_batch_query(queries, supports, q_batches, s_batches, radius=0.1):
    lib = ref_lib()
    return _query(lib.ref_batch_query, lib.ref_free, queries, supports, q_batches, s_batches, radius)


def oracle_subsample_batch(points, batches, sampleDl=0.1, max_p=0):
    lib = oracle_lib()
    p, b = _pts(points, "points"), _i32(batches)
    out = np.empty((max(p.shape[0], 1), 3), np.float32)
    ob = np.zeros(b.shape[0], np.int32)
    m = lib.oracle_grid_subsample_batch(p.ctypes.data_as(_f32p), p.shape[0], b.ctypes.data_as(_i32p),
                                        b.shape[0], ctypes.c_float(sampleDl), int(max_p),
                                        out.ctypes.data_as(_f32p), ob.ctypes.data_as(_i32p))
    if m < 1:
        raise RuntimeError("Error")
    return out[:m].copy(), ob


def ref_subsample_batch(points, batches, sampleDl=0.1, max_p=0):
    lib = ref_lib()
    p, b = _pts(points, "points"), _i32(batches)
    m = ctypes.c_int(0)
    ob = np.zeros(b.shape[0], np.int32)
    ptr = lib.ref_subsample_batch(p.ctypes.data_as(_f32p), p.shape[0], b.ctypes.data_as(_i32p),
                                  b.shape[0], ctypes.c_float(sampleDl), int(max_p), ctypes.byref(m),
                                  ob.ctypes.data_as(_i32p))
    if m.value < 1:
        lib.ref_free(ptr)
        raise RuntimeError("Error")
    out = np.ctypeslib.as_array(ctypes.cast(ptr, _f32p), shape=(m.value * 3,)).copy().reshape(m.value, 3)
    lib.ref_free(ptr)
    return out, ob


def oracle_umap_order(keys):
    lib = oracle_lib()
    k = np.ascontiguousarray(np.asarray(keys), dtype=np.uint64)
    order = np.empty(k.shape[0], np.int32)
    rc = lib.oracle_umap_order(k.ctypes.data_as(_u64p), k.shape[0], order.ctypes.data_as(_i32p))
    if rc != 0:
        raise RuntimeError("oracle_umap_order failed")
    return order


def ref_umap_order(keys):
    lib = ref_lib()
    k = np.ascontiguousarray(np.asarray(keys), dtype=np.uint64)
    order = np.empty(k.shape[0], np.int32)
    lib.ref_umap_order(k.ctypes.data_as(_u64p), k.shape[0], order.ctypes.data_as(_i32p))
    return order


def oracle_pyramid(points, lengths, cfg, limits, tie_order="reference", query=None, subsample=None):
    """The reference's collate_fn_descriptor pyramid (ref:datasets/dataloader.py:230-361) on the CPU
    checker: per level one conv table, then grid subsampling, one pool and one upsample table; radius
    and cell size double per level; tables cut at [:, :limit] and cast to int64.  Returns the batch
    dict the model reads (torch CPU tensors).  ``query`` / ``subsample`` default to this repo's C
    restatement; pass ref_batch_query / ref_subsample_batch to run the unmodified reference cores."""
    import torch
    if query is None:
        def query(q, s, ql, sl, radius):
            return oracle_batch_query(q, s, ql, sl, radius, tie_order=tie_order)
    subsample = subsample or oracle_subsample_batch
    pts, lens = _f32(points), _i32(lengths)
    r = cfg["first_subsampling_dl"] * cfg["conv_radius"]
    dl = 2 * cfg["first_subsampling_dl"]
    batch = {"points": [], "neighbors": [], "pools": [], "upsamples": [], "stack_lengths": []}
    empty = torch.zeros((0, 1), dtype=torch.int64)
    n_layers = cfg["num_layers"]
    for l in range(n_layers):
        batch["points"].append(torch.from_numpy(pts))
        batch["stack_lengths"].append(torch.from_numpy(lens))
        batch["neighbors"].append(torch.from_numpy(query(pts, pts, lens, lens, r)[:, :limits[l]]).long())
        if l == n_layers - 1:
            batch["pools"].append(empty)
            batch["upsamples"].append(empty)
            break
        sp, sl = subsample(pts, lens, dl)
        batch["pools"].append(torch.from_numpy(query(sp, pts, sl, lens, r)[:, :limits[l]]).long())
        batch["upsamples"].append(torch.from_numpy(query(pts, sp, lens, sl, 2 * r)[:, :limits[l]]).long())
        pts, lens, r, dl = sp, sl, r * 2, dl * 2
    batch["features"] = torch.ones((batch["points"][0].shape[0], 1))
    return batch
