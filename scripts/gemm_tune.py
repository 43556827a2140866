"""Tile / split-K sweep of pcrcg_gemm_f32 on the path's GEMM shapes (tuning aid, GPU box only)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pcrcg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=12, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


SHAPES = [(4096, 4096, 4096, 0), (60000, 960, 64, 0), (15456, 1920, 128, 0), (3934, 3840, 256, 0),
          (763, 7680, 512, 0), (60000, 128, 256, 1), (15456, 256, 512, 1), (3934, 512, 1024, 1),
          (763, 1024, 2048, 1), (381, 512, 1024, 1), (15456, 769, 128, 1), (60000, 384, 34, 1)]
TILES = {0: "128x128", 1: "128x64", 2: "64x128", 3: "64x64"}

for m, k, n, tb in SHAPES:
    a = torch.randn(m, k, device=dev)
    b = torch.randn(n, k, device=dev).t() if tb else torch.randn(k, n, device=dev)
    tt = timeit(lambda: torch.matmul(a, b))
    fl = 2.0 * m * n * k
    res = []
    for tile in range(4):
        if n <= 64 and tile in (0, 2):
            continue
        for sk in (1, 2, 4, 8, 16):
            if sk > 1 and k // sk < 128:
                continue
            os.environ["PCRCG_GEMM_TILE"] = str(tile)
            os.environ["PCRCG_GEMM_SPLITK"] = str(sk)
            t = timeit(lambda: ops.gemm(a, b))
            res.append((t, tile, sk))
    os.environ.pop("PCRCG_GEMM_TILE")
    os.environ.pop("PCRCG_GEMM_SPLITK")
    tauto = timeit(lambda: ops.gemm(a, b))
    res.sort()
    best = ", ".join(f"{TILES[t]}/sk{s}:{x:.0f}us" for x, t, s in res[:4])
    print(f"M={m:6d} K={k:5d} N={n:5d} tb={tb}: torch {tt:7.1f}us ({fl / tt / 1e6:5.1f}TF) auto {tauto:7.1f}us "
          f"({fl / tauto / 1e6:5.1f}TF) | best {best} | worst {res[-1][0]:.0f}us")
