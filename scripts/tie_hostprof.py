"""cProfile of build_pyramid (host side): where does the front-end thread's time go?  argv: recipe mode"""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcrcg_amd import synthetic  # noqa: E402
from pcrcg_amd.config import indoor_config  # noqa: E402
from pcrcg_amd.pyramid import build_pyramid  # noqa: E402

recipe, mode = sys.argv[1], sys.argv[2]
limits = synthetic.LIMITS.get(recipe, [25, 36, 45, 42])
dev = torch.device("cuda:0")
src, tgt = synthetic.pair(recipe, 0)
pts = torch.from_numpy(np.concatenate([src, tgt])).to(dev)
lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev)
cfg = indoor_config()
for _ in range(5):
    build_pyramid(pts, lens, cfg, limits, tie_order=mode)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    build_pyramid(pts, lens, cfg, limits, tie_order=mode)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
