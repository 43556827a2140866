"""Times pcrcg_gemm_f32 on every distinct GEMM shape of one S30k forward (read from a PCRCG_GEMM_LOG
capture, default gpurun_out/gemm_shapes.log or the built-in list) against torch.matmul (hipBLASLt):
us, TFLOP/s, and GB/s of compulsory traffic (A + B + C once).  GPU box only."""
import collections
import os
import re
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pcrcg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BUILTIN = """60000 128 15 0 1|60000 64 128 1 0|60000 64 960 0 1|60000 256 64 1 0|60000 256 128 1 0|60000 64 256 1 0
15433 64 960 0 1|15433 256 64 1 0|15433 128 256 1 0|15433 128 1920 0 1|15433 512 128 1 0|15433 512 256 1 0
15433 128 512 1 0|3917 128 1920 0 1|3917 512 128 1 0|3917 256 512 1 0|3917 256 3840 0 1|3917 1024 256 1 0
3917 1024 512 1 0|3917 256 1024 1 0|759 256 3840 0 1|759 1024 256 1 0|759 512 1024 1 0|759 512 7680 0 1
759 2048 512 1 0|759 2048 1024 1 0|759 512 2048 1 0|379 1024 512 0 0|379 2048 512 0 0|379 512 2048 1 0
379 512 512 1 0|379 380 128 1 0|379 128 380 0 0|379 1024 1024 1 0|379 512 1024 1 0|3917 257 1538 1 0
15433 128 769 1 0|60000 34 384 1 0"""


def shapes():
    path = sys.argv[1] if len(sys.argv) > 1 else None
    cnt = collections.Counter()
    if path and os.path.exists(path):
        for l in open(path):
            m = re.search(r"m=(\d+) n=(\d+) k=(\d+).*tb=(\d).*rs=(\d)", l)
            if m:
                cnt[tuple(int(x) for x in m.groups())] += 1
    else:
        for tok in BUILTIN.replace("\n", "|").split("|"):
            cnt[tuple(int(x) for x in tok.split())] += 1
    return cnt


def timeit(fn, reps=20, warm=3):
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


tot_own = tot_torch = 0.0
print(f"{'m':>6} {'n':>5} {'k':>5} tb cnt |   own us    TF   GB/s | torch us    TF")
for (m, n, k, tb, rs), c in shapes().items():
    kk = (k + 3) // 4 * 4
    a = torch.randn(m, kk, device=dev)[:, :k]
    b = torch.randn(n, kk, device=dev)[:, :k].t() if tb else torch.randn(k, n, device=dev)
    scale = torch.rand(m, device=dev) if rs else None
    own = timeit(lambda: ops.gemm(a, b, row_scale=scale))
    ref = timeit(lambda: torch.matmul(a, b))
    fl, by = 2.0 * m * n * k, 4.0 * (m * k + n * k + m * n)
    tot_own += own * c
    tot_torch += ref * c
    print(f"{m:6d} {n:5d} {k:5d} {tb:2d} {c:3d} | {own:8.1f} {fl / own / 1e6:5.1f} {by / own / 1e3:6.0f} | "
          f"{ref:8.1f} {fl / ref / 1e6:5.1f}")
print(f"sum over one forward: own {tot_own:.0f} us, torch.matmul {tot_torch:.0f} us")
