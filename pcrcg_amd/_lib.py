"""ctypes binding of libpcrcg_hip.so (the C ABI declared in include/pcrcg.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.  The
product path never imports anything from ``oracle/``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# Always the in-tree library.  (scripts/knockout.py, a measurement aid, assigns LIB_PATH before the first call to load
# its instrumented build; no environment variable can redirect the product.)
LIB_PATH = os.path.join(_HERE, "libpcrcg_hip.so")


def lib_identity():
    """(path, first 16 hex digits of the SHA-256) of the shared library that is loaded -- bench.py records both."""
    import hashlib
    with open(LIB_PATH, "rb") as f:
        return LIB_PATH, hashlib.sha256(f.read()).hexdigest()[:16]

ABI_VERSION = 4      # PCRCG_ABI_VERSION of the include/pcrcg.h these signatures were written against

c_int, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/pcrcg.h and include/pcrcg_train.h one to one
SIGNATURES = {
    "pcrcg_last_error": (ctypes.c_char_p, []),
    "pcrcg_abi_version": (c_int, []),
    "pcrcg_debug_set": (c_int, [ctypes.c_char_p]),
    "pcrcg_check_status": (c_int, [c_void_p, c_void_p]),
    "pcrcg_grid_subsample_ws_bytes": (c_size_t, [c_int, c_int]),
    "pcrcg_grid_subsample_batch": (c_int, [c_void_p, c_int, c_void_p, c_int, c_float, c_int, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_size_t, c_void_p]),
    "pcrcg_umap_order_ws_bytes": (c_size_t, [c_int]),
    "pcrcg_umap_order": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pcrcg_cellgrid_ws_bytes": (c_size_t, [c_int, c_int]),
    "pcrcg_cellgrid_build": (c_int, [c_void_p, c_int, c_void_p, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "pcrcg_radius_query": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_float, c_void_p, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcrcg_radius_query_ex": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_float, c_void_p, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcrcg_radius_query_groups": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_void_p, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcrcg_radius_query_cells": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_float,
                                         c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcrcg_kdforest_ws_bytes": (c_size_t, [c_int, c_int]),
    "pcrcg_kdforest_build": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "pcrcg_radius_reorder": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                     c_float, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pcrcg_radius_reorder_jobs": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pcrcg_radius_neighbors_ws_bytes": (c_size_t, [c_int, c_int]),
    "pcrcg_radius_neighbors_batch": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_float,
                                             c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                             c_void_p]),
    "pcrcg_kpconv_ws_bytes": (c_size_t, [c_int]),
    "pcrcg_kpconv_aggregate": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                       c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pcrcg_kpconv_aggregate_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                            c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                            c_void_p]),
    "pcrcg_profile_kpconv": (None, [c_int]),
    "pcrcg_profile_kpconv_read": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "pcrcg_gemm_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                               c_void_p, c_void_p, c_void_p]),
    "pcrcg_gemm_colstats_bytes": (c_size_t, [c_int, c_int]),
    "pcrcg_gemm_f32_colstats": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "pcrcg_gemm_bf16a_f32_colstats": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                              c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "pcrcg_gemm_f32_fused": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.c_double, c_float,
                                     c_float, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "pcrcg_gemm_set_mode": (None, [c_int]),
    "pcrcg_gemm_get_mode": (c_int, []),
    "pcrcg_gemm_redo_counts": (c_int, [c_void_p, c_int]),
    "pcrcg_instnorm_stats_from_partials": (c_int, [c_void_p, c_int, c_int, ctypes.c_double, c_float, c_void_p,
                                                   c_void_p]),
    "pcrcg_gather_max": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pcrcg_gather_first": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "pcrcg_instnorm_ws_bytes": (c_size_t, [c_int]),
    "pcrcg_instnorm_stats": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pcrcg_instnorm_apply": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_float,
                                     c_void_p, c_int, c_void_p]),
    "pcrcg_instnorm_colsums": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pcrcg_instnorm_apply_sums": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, ctypes.c_double, c_float, c_void_p, c_int,
                                          c_void_p, c_float, c_void_p, c_int, c_void_p]),
    "pcrcg_fill2d": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "pcrcg_inject_image_features": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, ctypes.c_long,
                                            ctypes.c_long, c_void_p, c_int, c_void_p]),
    "pcrcg_knn": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "pcrcg_edgeconv_ws_bytes": (c_size_t, [c_int]),
    "pcrcg_edgeconv_reduce": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float,
                                      c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pcrcg_edgeconv_reduce_sums": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                           c_void_p, c_void_p]),
    "pcrcg_softmax_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "pcrcg_attention_supported": (c_int, [c_int]),
    "pcrcg_attention": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                c_int, c_float, c_void_p]),
    "pcrcg_softmax_matvec": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "pcrcg_copy2d": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "pcrcg_add": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_long, c_void_p]),
    "pcrcg_l2norm_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "pcrcg_sigmoid_scores": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    # struct-based entry points: pcrcg_amd/runner.py declares the ctypes.Structure mirrors
    "pcrcg_kpfcnn_ws_bytes": (c_size_t, [c_void_p, c_void_p]),
    "pcrcg_kpfcnn_group_ws_bytes": (c_size_t, [c_void_p, c_void_p, c_int]),
    "pcrcg_kpfcnn_forward_group": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "pcrcg_kpfcnn_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pcrcg_pyramid_ws_bytes": (c_size_t, [c_int, c_int, c_void_p]),
    "pcrcg_pyramid_build": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcrcg_pyramid_build_parts": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcrcg_pyramid_restore_run": (c_int, [c_void_p, c_void_p, c_void_p]),
    "pcrcg_debug_release": (c_int, []),
    "pcrcg_stream_pipe_classes": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "pcrcg_thread_shares_gpu": (None, [c_int]),
    "pcrcg_stream_create": (c_int, [c_void_p, c_int]),
    "pcrcg_stream_destroy": (c_int, [c_void_p]),
    # include/pcrcg_train.h -- the "next" rows
    "pcrcg_gemm_f32_ex": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                  c_void_p, c_void_p, c_void_p]),
    "pcrcg_gemm_f32_grad": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                  c_void_p, c_void_p, c_int, c_void_p]),
    "pcrcg_kpconv_backward_dx": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                         c_void_p, c_float, c_void_p, c_void_p]),
    "pcrcg_kpconv_forward_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pcrcg_kpconv_forward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                     c_float, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "pcrcg_kpconv_backward_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pcrcg_kpconv_backward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_float,
                                      c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                                      c_size_t, c_void_p]),
    "pcrcg_gather_max_backward": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                          c_void_p, c_void_p]),
    "pcrcg_gather_first_backward": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pcrcg_instnorm_backward_ws_bytes": (c_size_t, [c_int]),
    "pcrcg_instnorm_backward": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p,
                                        c_int, c_void_p, c_size_t, c_void_p]),
    "pcrcg_attention_backward_supported": (c_int, [c_int] * 7),
    "pcrcg_attention_backward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                         c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                         c_void_p]),
    "pcrcg_softmax_rows_backward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int,
                                            c_void_p]),
    "pcrcg_edgeconv_backward_ws_bytes": (c_size_t, [c_int]),
    "pcrcg_edgeconv_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                        c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pcrcg_correspondences_rows": (c_int, [c_void_p, c_int, c_void_p, ctypes.c_double, c_int, c_int, c_void_p, c_int, c_void_p,
                                           c_void_p, c_void_p, c_void_p]),
    "pcrcg_correspondences_emit": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "pcrcg_kpfcnn_train_ws_bytes": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcrcg_kpfcnn_train_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p,
                                           c_void_p, c_void_p]),
    "pcrcg_kpfcnn_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcrcg_kpfcnn_train_free": (None, [c_void_p]),
    "pcrcg_circle_loss": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float,
                                  c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                  c_void_p]),
    "pcrcg_circle_loss_ws_bytes": (c_size_t, [c_int]),
    "pcrcg_weighted_bce_ws_bytes": (c_size_t, []),
    "pcrcg_gather_jobs": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "pcrcg_nonfinite_flag": (c_int, [c_void_p, ctypes.c_long, c_void_p, c_void_p]),
    "pcrcg_sgd_step": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_long, c_float, c_float, c_float, c_int, c_void_p]),
    "pcrcg_weighted_bce": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pcrcg_feature_argmax_ws_bytes": (c_size_t, [c_int]),
    "pcrcg_feature_argmax": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
}

_lib = None


def lib():
    """Load libpcrcg_hip.so once; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C pcrcg_amd/csrc` "
                "(or __graft_entry__.build()).  pcrcg_amd has no CPU or PyTorch fallback.")
        # torch first: its wheel bundles its own libamdhip64, and the library must bind to THAT copy of the HIP runtime (the
        # one that owns the device memory and streams it is handed).  Loaded the other way round -- this library first,
        # pulling in the system's libamdhip64, torch afterwards -- the process holds two runtimes and every call in here
        # fails with "no ROCm-capable device is detected" (seen with build() and smoke() in one process).
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        got = handle.pcrcg_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} speaks ABI version {got}, this binding was written against {ABI_VERSION}: "
                               "rebuild it with `make -C pcrcg_amd/csrc`")
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().pcrcg_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
