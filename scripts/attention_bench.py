"""GPU box: pcrcg_attention (one launch, all heads) against the per-head GEMM / softmax / GEMM sequence it replaces
in the runner, on the full-width GNN's shape (4 heads x 128, ~380 coarse points per cloud): microseconds per call from
a HIP-graph replay, and the error of both against float64."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcrcg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, inner=20, reps=7):
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


for (n, ms, heads, d) in [(381, 382, 4, 128), (763, 760, 4, 128), (381, 382, 4, 64), (1900, 1900, 4, 128)]:
    torch.manual_seed(0)
    ch = heads * d
    q, k, v = (torch.randn(r, ch, device=dev) for r in (n, ms, ms))

    def per_head():
        out = torch.empty(n, ch, device=dev)
        for h in range(heads):
            sl = slice(h * d, (h + 1) * d)
            sc = ops.gemm(q[:, sl], k[:, sl].t())
            ops.softmax_rows_(sc, d ** -0.5)
            out[:, sl] = ops.gemm(sc, v[:, sl].contiguous())
        return out

    want = torch.cat([torch.softmax(q[:, h * d:(h + 1) * d].double() @ k[:, h * d:(h + 1) * d].double().t() / d ** 0.5, 1)
                      @ v[:, h * d:(h + 1) * d].double() for h in range(heads)], 1)
    err = lambda x: float((x.double() - want).abs().max() / want.abs().max())
    print(f"n={n} ms={ms} heads={heads} d={d}: one launch {timeit(lambda: ops.attention(q, k, v, heads)):.1f} us "
          f"(err {err(ops.attention(q, k, v, heads)):.1e}), per head {timeit(per_head):.1f} us (err {err(per_head()):.1e})")
