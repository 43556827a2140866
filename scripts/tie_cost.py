"""Cost of restoring the reference's tie order (csrc/tieorder.hip): build_pyramid wall time per pair, synchronised,
for tie_order index / auto / reference.  GPU box only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcrcg_amd import synthetic  # noqa: E402
from pcrcg_amd.config import indoor_config  # noqa: E402
from pcrcg_amd.pyramid import build_pyramid  # noqa: E402

import numpy as np  # noqa: E402

dev = torch.device("cuda:0")
cfg = indoor_config()
for recipe, limits in (("S30k", synthetic.LIMITS["S30k"]), ("T8k", [25, 36, 45, 42])):
    src, tgt = synthetic.pair(recipe, 0)
    pts = torch.from_numpy(np.concatenate([src, tgt])).to(dev)
    lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev)
    for mode in ("index", "auto", "reference"):
        for _ in range(5):
            build_pyramid(pts, lens, cfg, limits, tie_order=mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            build_pyramid(pts, lens, cfg, limits, tie_order=mode)
        torch.cuda.synchronize()
        print(f"{recipe} tie_order={mode:9s} {1e3 * (time.perf_counter() - t0) / n:7.3f} ms / pair", flush=True)
