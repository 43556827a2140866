"""rocprofv3 target: build_pyramid(tie_order=...) on one recipe, a few iterations.  argv: recipe mode iters"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcrcg_amd import synthetic  # noqa: E402
from pcrcg_amd.config import indoor_config  # noqa: E402
from pcrcg_amd.pyramid import build_pyramid  # noqa: E402

recipe, mode, iters = sys.argv[1], sys.argv[2], int(sys.argv[3])
limits = synthetic.LIMITS.get(recipe, [42, 41, 47, 43] if recipe == "T30k" else [25, 36, 45, 42])
dev = torch.device("cuda:0")
src, tgt = synthetic.pair(recipe, 0)
pts = torch.from_numpy(np.concatenate([src, tgt])).to(dev)
lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev)
for _ in range(iters):
    build_pyramid(pts, lens, indoor_config(), limits, tie_order=mode)
torch.cuda.synchronize()
