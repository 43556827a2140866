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
    assert lib.pcrcg_abi_version() == 4


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
    rc = lib.pcrcg_stream_pipe_classes(None, 4, None, None)        # (checked before the probe touches the GPU)
    assert rc == -1 and b"bad argument" in lib.pcrcg_last_error()


def test_struct_mirrors_match_the_header(tmp_path):
    """The ctypes mirrors of the C structs (pcrcg_amd/runner.py, ops.py) have the header's sizes: a plain-C program that
    includes include/pcrcg.h prints sizeof(...) and the mirrors must agree (ABI version 4 changed three of them)."""
    import subprocess
    from pcrcg_amd import runner
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "pcrcg.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(pcrcg_reorder_job), sizeof(pcrcg_pyramid_restore), sizeof(pcrcg_pyramid_cfg), sizeof(pcrcg_batch), '
                   'sizeof(pcrcg_model), sizeof(pcrcg_table));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    mirrors = [runner.ReorderJobC, runner.PyramidRestore, runner.PyramidCfg, runner.Batch, runner.Model, runner.Table]
    assert sizes == [ctypes.sizeof(m) for m in mirrors]


def test_pyramid_workspace_follows_the_row_bound():
    """pcrcg_pyramid_ws_bytes sizes the arena from cfg.shrink alone (round 6: no row count is read back while a pyramid is
    built): a smaller bound gives a smaller arena, 0 / out-of-range means 1.0."""
    from pcrcg_amd import indoor_config
    from pcrcg_amd.runner import PyramidCfg
    from pcrcg_amd.pyramid import _layer_plan, as_config
    plan = _layer_plan(as_config(indoor_config()))
    c = PyramidCfg()
    c.n_levels = len(plan)
    for l, lv in enumerate(plan):
        c.r_conv[l], c.r_pool[l], c.dl[l] = float(lv["r_conv"]), float(lv["r_pool"]), float(lv["dl"])
        c.has_conv[l], c.pooled[l], c.limit[l] = int(lv["has_conv"]), int(lv["pooled"]), 40
    c.tie_order = 1
    lib = _lib.lib()
    sizes = {}
    for shrink in (0.25, 0.5, 1.0, 0.0, 7.0):
        c.shrink = shrink
        sizes[shrink] = lib.pcrcg_pyramid_ws_bytes(60000, 2, ctypes.byref(c))
    assert 0 < sizes[0.25] < sizes[0.5] < sizes[1.0]
    assert sizes[0.0] == sizes[1.0] == sizes[7.0]
    assert lib.pcrcg_pyramid_ws_bytes(60000, 0, ctypes.byref(c)) == 0
