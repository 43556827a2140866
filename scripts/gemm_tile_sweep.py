"""GPU box: the fp16-form GEMM on the large-M shapes of a forward with the tile shape forced (PCRCG_DEBUG x6_tile), us per call
from a HIP-graph replay."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pcrcg_amd import _lib, ops
dev = torch.device("cuda:0")
L = _lib.lib()
SHAPES = [(60000, 64, 960), (60000, 256, 64), (60000, 256, 128), (60000, 64, 256), (15456, 128, 1920), (15456, 512, 128),
          (15456, 512, 256), (15456, 128, 512), (15456, 128, 256), (3934, 256, 3840), (3934, 1024, 256), (3934, 1024, 512), (3934, 256, 1024)]
def timeit(fn, reps=7, inner=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner): fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / inner)
    return float(np.median(ts))
print(f"{'m':>6} {'n':>5} {'k':>5} | auto    64x64   64x128  128x64  128x128 (us)")
for (m, n, k) in SHAPES:
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5
    row = []
    for spec in (None, b"x6_tile=3", b"x6_tile=2", b"x6_tile=1", b"x6_tile=0"):
        _lib.check(L.pcrcg_debug_set(spec), "debug")
        row.append(timeit(lambda: ops.gemm(a, w.t())))
    _lib.check(L.pcrcg_debug_set(None), "debug")
    print(f"{m:6d} {n:5d} {k:5d} | " + "  ".join(f"{t:6.1f}" for t in row))
