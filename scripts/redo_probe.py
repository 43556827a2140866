"""How often the fp16 GEMM form hands a tile to the bf16 redo (pcrcg_gemm_redo_counts) in one forward of a workload.
python scripts/redo_probe.py [S30k|K120k|U30k]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcrcg_amd import _lib, indoor_config, kitti_config, synthetic  # noqa: E402
from pcrcg_amd.architectures import KPFCNN  # noqa: E402
from pcrcg_amd.pyramid import build_pyramid  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "S30k"
dev = torch.device("cuda:0")
cfg = kitti_config() if name == "K120k" else indoor_config()
src, tgt = synthetic.slab_pair(120000, 0) if name == "K120k" else synthetic.pair(name, 0)
pts = torch.from_numpy(np.concatenate([src, tgt])).to(dev)
lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev)
batch = build_pyramid(pts, lens, cfg, synthetic.LIMITS[name])
torch.manual_seed(0)
np.random.seed(0)
net = KPFCNN(cfg).to(dev).eval()
L = _lib.lib()
with torch.no_grad():
    net(batch)
    torch.cuda.synchronize()
    L.pcrcg_gemm_redo_counts(None, 1)
    net(batch)
    torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 2)()
L.pcrcg_gemm_redo_counts(out, 1)
print(f"{name}: one forward: tiles redone beyond fp16's range {out[0]}, below it {out[1]}")
