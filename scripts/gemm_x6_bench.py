"""GPU box: pcrcg_gemm_f32 in both arithmetic modes (0 = fp32 MFMA, 1 = split-bf16 x6) on the C = A @ B^T
shapes of one S30k forward (KPConv contractions taken in their K-contiguous form): microseconds, TFLOP/s, and
the error of each mode against a float64 product (max|c - ref| / max|ref|)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pcrcg_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
# (m, n, k, count per forward)
SHAPES = [(60000, 128, 16, 1), (60000, 64, 128, 1), (60000, 64, 960, 1), (60000, 256, 64, 1), (60000, 256, 128, 1),
          (60000, 64, 256, 1), (15456, 64, 960, 1), (15456, 256, 64, 1), (15456, 128, 256, 1), (15456, 128, 1920, 2),
          (15456, 512, 128, 2), (15456, 512, 256, 1), (15456, 128, 512, 2), (3934, 128, 1920, 1), (3934, 512, 128, 1),
          (3934, 256, 512, 1), (3934, 256, 3840, 2), (3934, 1024, 256, 2), (3934, 1024, 512, 1), (3934, 256, 1024, 2),
          (763, 256, 3840, 1), (763, 1024, 256, 1), (763, 512, 1024, 1), (763, 512, 7680, 2), (763, 2048, 512, 2),
          (763, 2048, 1024, 1), (763, 512, 2048, 2), (381, 512, 2048, 4), (381, 512, 512, 8), (381, 1024, 1024, 2),
          (381, 512, 1024, 2), (3934, 257, 1540, 1), (15456, 128, 772, 1), (60000, 34, 384, 1)]


def timeit(fn, reps=7, inner=10):
    """Median microseconds per call with the calls replayed from a HIP graph (no host launch cost in the number)."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / inner)
    return float(np.median(ts))


L = _lib.lib()
tot = [0.0, 0.0, 0.0]
print(f"{'m':>6} {'n':>5} {'k':>5} cnt |  fp32 us    TF     err |   x6 us     TF     err | torch us    TF")
only = os.environ.get("SHAPES")
for (m, n, k, c) in SHAPES:
    if only and f"{m}x{n}x{k}" not in only.split(","):
        continue
    torch.manual_seed(m + n + k)
    a = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    ref = (a.double() @ w.double().t())
    scale = ref.abs().max()
    row = [m, n, k, c]
    fl = 2.0 * m * n * k
    for mode in (0, 1):
        L.pcrcg_gemm_set_mode(mode)
        out = ops.gemm(a, w.t())
        err = float((out.double() - ref).abs().max() / scale)
        out = torch.empty(m, n, device=dev)
        us = timeit(lambda: ops.gemm(a, w.t(), out=out))
        tot[mode] += us * c
        row += [us, fl / us / 1e6, err]
    if os.environ.get("SWEEP"):
        L.pcrcg_gemm_set_mode(1)
        best = []
        for t in range(4):
            for sk in (1, 2, 4, 8):
                if sk > 1 and k // sk < 128:
                    continue
                L.pcrcg_debug_set(("x6_tile=%d,x6_splitk=%d" % (t, sk)).encode())
                best.append((timeit(lambda: ops.gemm(a, w.t(), out=out), reps=3), t, sk))
        L.pcrcg_debug_set(b"x6_tile=-1,x6_splitk=0")
        best.sort()
        print("      sweep:", " ".join("t%d/k%d=%.1f" % (t, sk, u) for u, t, sk in best[:5]), flush=True)
    us = timeit(lambda: torch.matmul(a, w.t(), out=out))
    tot[2] += us * c
    row += [us, fl / us / 1e6]
    print("%6d %5d %5d %3d | %8.1f %5.1f %.1e | %8.1f %6.1f %.1e | %8.1f %5.1f" % tuple(row), flush=True)
print(f"sum over one forward: fp32 MFMA {tot[0]:.0f} us, split-bf16 x6 {tot[1]:.0f} us, torch.matmul {tot[2]:.0f} us")
