"""KD-forest build time (HIP events) for a few cloud shapes.  GPU box only."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcrcg_amd import ops
rng = np.random.default_rng(0)
for n, nb in ((900, 1), (900, 8), (2000, 1), (4000, 1), (8000, 1), (16000, 1), (30000, 1), (30000, 2), (30000, 8)):
    pts = torch.from_numpy(rng.random((n * nb, 3)).astype(np.float32)).cuda()
    lens = torch.tensor([n] * nb, dtype=torch.int32).cuda()
    for _ in range(3):
        ops.KdForest(pts, lens)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.KdForest(pts, lens)
    b.record()
    torch.cuda.synchronize()
    print(f"n={n:6d} x {nb}: {a.elapsed_time(b) / 20 * 1e3:8.1f} us per forest", flush=True)
