"""GPU: the C ABI driven by a plain C host program (examples/c_host_frontend.c, no Python / torch in the
process) gives the same result as the Python binding on the same input."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest
import torch

from pcrcg_amd import ops
from pcrcg_amd.cpp_wrappers.cpp_neighbors import radius_neighbors

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lcg_points(n3):
    out = np.empty(n3, np.float32)
    s = 12345
    for i in range(n3):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = np.float32(s >> 8) * np.float32(1.0 / 16777216.0)
    return out


def test_c_host_program_matches_python_binding(cuda, tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("gcc / ROCm headers not available on this box")
    exe = str(tmp_path / "c_host_frontend")
    subprocess.run([gcc, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(REPO, "include"),
                    os.path.join(REPO, "examples", "c_host_frontend.c"), "-o", exe, "-L", os.path.join(REPO, "pcrcg_amd"),
                    "-l:libpcrcg_hip.so", "-L/opt/rocm/lib", "-lamdhip64"], check=True, timeout=300)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(REPO, "pcrcg_amd") + ":/opt/rocm/lib:"
               + os.environ.get("LD_LIBRARY_PATH", ""))
    res = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    got = {k: v for k, v in re.findall(r"(\w+)=([\d.]+)", res.stdout)}

    pts = torch.from_numpy(_lcg_points(3 * 8000).reshape(-1, 3)).to(cuda)
    lens = torch.tensor([4000, 4000], dtype=torch.int32, device=cuda)
    sub, sub_len = ops.grid_subsample(pts, lens, 0.05)
    grid = ops.CellGrid(sub, sub_len, 0.125)
    idx, meta = grid.query(sub, sub_len, 40)
    m = sub.shape[0]
    assert int(got["m"]) == m and [int(got["len0"]), int(got["len1"])] == sub_len.tolist()
    assert int(got["max_count"]) == int(meta[0]) and int(got["status"]) == 0
    assert int(got["idx_sum"]) == int(idx.sum()) and int(got["shadow"]) == int((idx == m).sum())
    assert abs(float(got["coord_sum"]) - float(sub.double().sum())) < 1e-3

    # part 3 of the program: lattice-snapped clouds, the reference's order inside tie groups -- against the Python
    # mirror of batch_query (same ABI calls) and, through it, the oracle-pinned order
    raw = _lcg_points(3 * 8000)[:3 * 3000]
    snapped = (np.floor(raw * np.float32(32.0)).astype(np.int32).astype(np.float32) * np.float32(1.0 / 32.0)).reshape(-1, 3)
    want = radius_neighbors.batch_query(snapped, snapped, [1500, 1500], [1500, 1500], radius=0.11)
    by_index = radius_neighbors.batch_query(snapped, snapped, [1500, 1500], [1500, 1500], radius=0.11, tie_order="index")
    keep = min(want.shape[1], 48)
    w = np.arange(1, keep + 1, dtype=np.uint64)[None, :]
    assert int(got["tie_status"]) == 0 and int(got["reorder_status"]) == 0 and int(got["tie_rows"]) > 1000
    assert int(got["tie_max_count"]) == want.shape[1]
    assert int(got["order_sum"]) == int((want[:, :keep].astype(np.uint64) * w).sum())
    assert int(got["order_sum"]) != int((by_index[:, :keep].astype(np.uint64) * w).sum())
