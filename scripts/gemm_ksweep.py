"""GPU box: what a small C = A @ B^T product costs as a function of K (M x N fixed), back to back from a HIP graph: the
slope is the k-step, the intercept the per-launch fixed cost (boundary + prologue + epilogue).
    PCRCG_DEBUG=x6_splitk=1 python scripts/gemm_ksweep.py [M N]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pcrcg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 381
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512


def timeit(fn, reps=7, inner=20):
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


xs, ys = [], []
for k in (32, 64, 128, 256, 512, 1024, 2048, 4096):
    a = torch.randn(M, k, device=dev)
    w = torch.randn(N, k, device=dev)
    c = torch.empty(M, N, device=dev)
    t = timeit(lambda: ops.gemm(a, w.t(), out=c))
    xs.append(k / 32)
    ys.append(t)
    print(f"{M} x {N} x {k:5d}: {t:7.2f} us  ({k // 32} k-steps)")
slope, icpt = np.polyfit(xs[2:], ys[2:], 1)
print(f"fit: {icpt:.2f} us fixed + {slope:.3f} us per 32-deep k-step")
