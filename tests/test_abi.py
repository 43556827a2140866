"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/pcrcg.h and
include/pcrcg_train.h declare; argument validation works without a GPU (no compute is launched here)."""
import ctypes
import os
import re

from pcrcg_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for header in ("pcrcg.h", "pcrcg_train.h"):
        text = open(os.path.join(REPO, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(pcrcg_\w+)\s*\(", text))
    return sorted(names)


def test_library_exports_whole_header():
    lib = _lib.lib()
    names = _declared()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/*.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "pcrcg_amd/_lib.py must bind exactly the declared ABI"
    assert lib.pcrcg_abi_version() == 3


def test_workspace_queries():
    lib = _lib.lib()
    assert lib.pcrcg_grid_subsample_ws_bytes(60000, 2) > 60000 * 40
    assert lib.pcrcg_grid_subsample_ws_bytes(60000, 2) < 64 << 20
    assert lib.pcrcg_cellgrid_ws_bytes(60000, 2) >= 60000 * (16 + 16 + 8)
    assert lib.pcrcg_cellgrid_ws_bytes(0, 1) > 0
    assert lib.pcrcg_instnorm_ws_bytes(2048) >= 128 * 2 * 2048 * 8
    assert lib.pcrcg_umap_order_ws_bytes(1000) > 1000 * 4 * 7
    assert lib.pcrcg_kpconv_ws_bytes(60000) >= 60000


def test_bad_arguments_are_rejected_before_any_launch():
    lib = _lib.lib()
    rc = lib.pcrcg_gemm_f32(None, 4, None, 4, 0, None, 4, 4, 4, 4, None, None, None)
    assert rc == -1 and b"bad argument" in lib.pcrcg_last_error()
    rc = lib.pcrcg_gemm_f32(ctypes.c_void_p(16), 2, ctypes.c_void_p(16), 4, 0, ctypes.c_void_p(16), 4, 4, 4, 4, None,
                            None, None)
    assert rc == -1  # lda < k
    rc = lib.pcrcg_grid_subsample_batch(None, 10, None, 1, 0.1, 0, None, None, None, None, 0, None)
    assert rc == -1
    rc = lib.pcrcg_radius_query(None, 10, None, 5, None, 1, 0.1, None, 0, None, None, None, None, None)
    assert rc == -1
    rc = lib.pcrcg_kpconv_aggregate(None, 5, None, 5, None, 0, 0, None, 1, None, 0.1, None, None, None, 0, None)
    assert rc == -1
