"""Neighbourhood limits of the secondary U30k workload with the reference formula (GPU box only)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pcrcg_amd import indoor_config, synthetic
from pcrcg_amd.pyramid import calibrate_neighbors
dev = torch.device("cuda:0")
cfg = indoor_config()
pairs = []
for s in range(3):
    a, b = synthetic.uniform_pair(30000, 1.07, s)
    pairs.append((torch.from_numpy(np.concatenate([a, b])).to(dev),
                  torch.tensor([len(a), len(b)], dtype=torch.int32, device=dev)))
print("U30k limits", calibrate_neighbors(pairs, cfg))
