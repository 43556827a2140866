"""GPU box: the front end alone (pcrcg_pyramid_build, one call per pair) on an otherwise idle GPU: ms per pyramid,
single thread and with T threads sharing one stream.  python scripts/front_only.py [recipe] [threads]"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcrcg_amd import indoor_config, synthetic  # noqa: E402
from pcrcg_amd.pyramid import NativePyramid  # noqa: E402

recipe = sys.argv[1] if len(sys.argv) > 1 else "S30k"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
cfg = indoor_config()
limits = synthetic.LIMITS[recipe]
src, tgt = synthetic.pair(recipe, 0)
pts = torch.from_numpy(np.concatenate([src, tgt])).to(dev)
lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev)
stream = torch.cuda.Stream()
N = 100


def work(n, out):
    torch.cuda.set_device(dev)
    nat = NativePyramid(cfg, limits, os.environ.get("PCRCG_TIE_ORDER", "auto"))
    with torch.cuda.stream(stream):
        for _ in range(5):
            nat.build(pts, lens)
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            nat.build(pts, lens)
        stream.synchronize()
        out.append(time.perf_counter() - t0)


res = []
ts = [threading.Thread(target=work, args=(N, res)) for _ in range(threads)]
t0 = time.perf_counter()
for t in ts:
    t.start()
for t in ts:
    t.join()
wall = time.perf_counter() - t0
print(f"{recipe}: {threads} thread(s) on one stream, {N} pyramids each: {1e3 * max(res) / N:.3f} ms per call, "
      f"{1e3 * max(res) / (N * threads):.3f} ms per pyramid")
