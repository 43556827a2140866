"""Debug: T30k pyramid in auto mode, timed, with the limits bench.py would calibrate."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcrcg_amd import synthetic
from pcrcg_amd.config import indoor_config
from pcrcg_amd.pyramid import build_pyramid, calibrate_neighbors
dev = torch.device("cuda:0")
cfg = indoor_config()
src, tgt = synthetic.pair("T30k", 0)
pts = torch.from_numpy(np.concatenate([src, tgt])).to(dev)
lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev)
t0 = time.perf_counter()
limits = [int(v) for v in calibrate_neighbors([(pts, lens)], cfg, samples_threshold=0)]
torch.cuda.synchronize()
print("limits", limits, "calibration %.1f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
for mode in ("index", "auto"):
    for i in range(3):
        t0 = time.perf_counter()
        b = build_pyramid(pts, lens, cfg, limits, tie_order=mode)
        torch.cuda.synchronize()
        print(mode, i, "%.2f ms" % (1e3 * (time.perf_counter() - t0)), [int(t.shape[1]) for t in b["neighbors"]], flush=True)
