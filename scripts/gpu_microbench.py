"""Per-shape timings of the hot-path kernels at S30k sizes (run on the GPU box):
    python scripts/gpu_microbench.py [gemm] [kpconv] [radius] [misc]
Prints one line per case: microseconds (median of reps), derived TF/s or GB/s."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pcrcg_amd import indoor_config, ops, synthetic  # noqa: E402
from pcrcg_amd.pyramid import build_pyramid  # noqa: E402

dev = torch.device("cuda:0")


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


GEMM_SHAPES = [  # (M, K, N, tag)
    (60000, 15, 128, "kp0 1->128"), (60000, 960, 64, "kp 64->64 L0"), (15456, 960, 64, "kp strided L0"),
    (15456, 1920, 128, "kp L1"), (3934, 1920, 128, "kp strided L1"), (3934, 3840, 256, "kp L2"),
    (763, 3840, 256, "kp strided L2"), (763, 7680, 512, "kp L3"),
    (60000, 128, 64, "unary1 L0"), (60000, 64, 256, "unary2 L0"), (60000, 128, 256, "shortcut L0"),
    (60000, 256, 64, "unary1 strided"), (15456, 64, 256, "unary2"), (15456, 256, 128, "unary1 L1"),
    (15456, 128, 512, "unary2 L1"), (15456, 256, 512, "shortcut L1"), (3934, 512, 256, "unary1 L2"),
    (3934, 256, 1024, "unary2 L2"), (3934, 512, 1024, "shortcut L2"), (763, 1024, 512, "unary1 L3"),
    (763, 512, 2048, "unary2 L3"), (763, 1024, 2048, "shortcut L3"), (763, 2048, 512, "bottle"),
    (381, 512, 1024, "edge1"), (381, 512, 2048, "edge2"), (381, 2048, 512, "conv3"), (381, 1024, 1024, "mlp0"),
    (3934, 1538, 257, "dec1"), (15456, 769, 128, "dec2"), (60000, 384, 34, "dec3"),
]


def bench_gemm():
    print("== GEMM: mine vs torch.matmul (hipBLASLt/rocBLAS)")
    tot_m = tot_t = 0.0
    for m, k, n, tag in GEMM_SHAPES:
        a = torch.randn(m, k, device=dev)
        b = torch.randn(k, n, device=dev)
        tm = timeit(lambda: ops.gemm(a, b))
        tt = timeit(lambda: torch.matmul(a, b))
        fl = 2.0 * m * n * k
        tot_m += tm
        tot_t += tt
        print(f"{tag:16s} M={m:6d} K={k:5d} N={n:5d}  mine {tm:8.1f} us {fl / tm / 1e6:7.1f} TF | torch {tt:8.1f} us "
              f"{fl / tt / 1e6:7.1f} TF")
    print(f"sum mine {tot_m:.0f} us, torch {tot_t:.0f} us")


def s30k_batch():
    cfg = indoor_config()
    src, tgt = synthetic.pair("S30k", 0)
    pts = torch.from_numpy(np.concatenate([src, tgt])).to(dev)
    lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev)
    return cfg, pts, lens, build_pyramid(pts, lens, cfg, synthetic.LIMITS["S30k"])


def bench_kpconv():
    print("== KPConv aggregate (gather kernel) per launch")
    cfg, pts, lens, b = s30k_batch()
    L = ops._lib.lib()
    cases = [(0, False, 1, 128), (0, False, 64, 64), (0, True, 64, 64), (1, False, 128, 128), (1, True, 128, 128),
             (2, False, 256, 256), (2, True, 256, 256), (3, False, 512, 512)]
    for l, strided, cin, cout in cases:
        s = b["points"][l]
        q = b["points"][l + 1] if strided else s
        idx = b["pools"][l] if strided else b["neighbors"][l]
        x = torch.randn(s.shape[0], cin, device=dev)
        kp = (torch.rand(15, 3, device=dev) - 0.5) * 0.1 * 2 ** l
        w = torch.randn(15, cin, cout, device=dev)
        nq, h = idx.shape
        wf = torch.empty((nq, 15 * cin), device=dev)
        inv_n = torch.empty(nq, device=dev)
        nbytes = L.pcrcg_kpconv_ws_bytes(s.shape[0])
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def agg():
            L.pcrcg_kpconv_aggregate(q.data_ptr(), nq, s.data_ptr(), s.shape[0], idx.data_ptr(), h, idx.stride(0),
                                     x.data_ptr(), cin, kp.data_ptr(), 0.05 * 2 ** l, wf.data_ptr(), inv_n.data_ptr(),
                                     ws.data_ptr(), nbytes, st)
        t = timeit(agg)
        t2 = timeit(lambda: ops.kpconv(q, s, idx, x, kp, w, 0.05 * 2 ** l))
        byt = nq * h * (cin * 4 + 20) + nq * cout * 4
        print(f"L{l} strided={int(strided)} Nq={nq:6d} H={h} Cin={cin:4d}: aggregate {t:7.1f} us = {byt / t / 1e3:7.1f} GB/s "
              f"algorithmic | two-stage kpconv {t2:7.1f} us")


def bench_radius():
    print("== front end per call")
    cfg, pts, lens, b = s30k_batch()
    r = 0.0625
    for l in range(4):
        p, ln = b["points"][l], b["stack_lengths"][l]
        t_build = timeit(lambda: ops.CellGrid(p, ln, r))
        g = ops.CellGrid(p, ln, r)
        t_q = timeit(lambda: g.query(p, ln, synthetic.LIMITS["S30k"][l]))
        line = f"L{l} N={p.shape[0]:6d}: grid build {t_build:7.1f} us, conv query {t_q:7.1f} us"
        if l < 3:
            p2, ln2 = b["points"][l + 1], b["stack_lengths"][l + 1]
            t_p = timeit(lambda: g.query(p2, ln2, synthetic.LIMITS["S30k"][l]))
            g2 = ops.CellGrid(p2, ln2, 2 * r)
            t_u = timeit(lambda: g2.query(p, ln, synthetic.LIMITS["S30k"][l]))
            t_s = timeit(lambda: ops.grid_subsample(p, ln, 0.05 * 2 ** l))
            line += f", pool query {t_p:7.1f} us, up query {t_u:7.1f} us, subsample {t_s:7.1f} us"
        print(line)
        r *= 2
    t = timeit(lambda: build_pyramid(pts, lens, cfg, synthetic.LIMITS["S30k"]), reps=10)
    print(f"build_pyramid total {t:.0f} us")


def bench_misc():
    print("== point-wise blocks")
    cfg, pts, lens, b = s30k_batch()
    for n, c in ((60000, 64), (60000, 256), (15456, 512), (3934, 1024), (763, 2048)):
        x = torch.randn(n, c, device=dev)
        t1 = timeit(lambda: ops.instnorm_stats(x))
        st = ops.instnorm_stats(x)
        t2 = timeit(lambda: ops.instnorm_apply(x, st, 0.1))
        print(f"instnorm N={n:6d} C={c:5d}: stats {t1:6.1f} us, apply {t2:6.1f} us ({2 * n * c * 4 / t2 / 1e3:7.1f} GB/s)")
    for l, c in ((0, 256), (1, 512), (2, 1024)):
        x = torch.randn(b["points"][l].shape[0], c, device=dev)
        idx = b["pools"][l]
        t = timeit(lambda: ops.gather_max(x, idx))
        byt = idx.shape[0] * idx.shape[1] * (c * 4 + 8)
        print(f"gather_max L{l} Nq={idx.shape[0]} H={idx.shape[1]} C={c}: {t:6.1f} us = {byt / t / 1e3:7.1f} GB/s algorithmic")


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "kpconv", "radius", "misc"]
    for w in which:
        globals()["bench_" + w]()
