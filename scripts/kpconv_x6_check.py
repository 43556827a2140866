"""GPU box: the fused KPConv kernel (pcrcg_kpconv_x6) against the two-stage path on the S30k layers it serves:
max relative difference and time per call (HIP-graph replay)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcrcg_amd import _lib, indoor_config, ops, synthetic  # noqa: E402
from pcrcg_amd.pyramid import build_pyramid  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
cfg = indoor_config()
src, tgt = synthetic.pair("S30k", 0)
b = build_pyramid(torch.from_numpy(np.concatenate([src, tgt])).to(dev),
                  torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev), cfg, synthetic.LIMITS["S30k"])
torch.manual_seed(0)
kp = (torch.rand(15, 3, device=dev) - 0.5) * 0.06


def timeit(fn, inner=5, reps=5):
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
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) * 1e3 / inner)
    return float(np.median(ts))


for name, level, strided, cin, cout in (("L0 64->64", 0, False, 64, 64), ("L0 strided 64->64", 0, True, 64, 64),
                                        ("L1 128->128", 1, False, 128, 128), ("L1 strided 128->128", 1, True, 128, 128),
                                        ("L0 64->128", 0, False, 64, 128), ("L1 128->256", 1, False, 128, 256)):
    s_pts = b["points"][level]
    q_pts = b["points"][level + 1] if strided else s_pts
    idx = (b["pools"][level] if strided else b["neighbors"][level]).contiguous()
    ns, nq, h = s_pts.shape[0], q_pts.shape[0], idx.shape[1]
    x = torch.randn(ns, cin, device=dev).abs_() * (torch.rand(ns, 1, device=dev) > 0.1)
    w = torch.randn(15, cin, cout, device=dev) / (15 * cin) ** 0.5
    extent = 0.05 * 2 ** level
    ref = ops.kpconv(q_pts, s_pts, idx, x, kp, w, extent)
    wt = w.reshape(15 * cin, cout).t().contiguous()
    planes = torch.empty(int(L.pcrcg_split_bf16x3_bytes(cout, 15 * cin)), dtype=torch.uint8, device=dev)
    _lib.check(L.pcrcg_split_bf16x3(wt.data_ptr(), 15 * cin, cout, 15 * cin, planes.data_ptr(), None), "split")
    out = torch.empty(nq, cout, device=dev)
    wsb = L.pcrcg_kpconv_ws_bytes(ns)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

    def fused():
        _lib.check(L.pcrcg_kpconv_x6(q_pts.data_ptr(), nq, s_pts.data_ptr(), ns, idx.data_ptr(), h, idx.stride(0),
                                     x.data_ptr(), cin, kp.data_ptr(), extent, planes.data_ptr(), cout, out.data_ptr(), cout,
                                     ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream), "kpconv_x6")
    fused()
    torch.cuda.synchronize()
    err = float((out - ref).abs().max() / ref.abs().max())
    t_f = timeit(fused)
    t_2 = timeit(lambda: ops.kpconv(q_pts, s_pts, idx, x, kp, w, extent))
    print(f"{name:22s} nq {nq:6d} h {h:3d}: max rel diff {err:.2e}   fused {t_f:7.1f} us   two-stage {t_2:7.1f} us", flush=True)
