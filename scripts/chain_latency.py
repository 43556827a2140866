"""GPU: latency of one pyramid build on an otherwise idle GPU (wall time of NativePyramid.build + stream drain), for one and
four S30k pairs per chain, with the chain in line on one stream and as a DAG over one / two side streams of the same
dispatcher class.  python scripts/chain_latency.py"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pcrcg_amd import indoor_config, kitti_config, ops, synthetic  # noqa: E402
from pcrcg_amd.pyramid import NativePyramid  # noqa: E402

dev = torch.device("cuda:0")
cands = [torch.cuda.Stream(device=dev) for _ in range(12)]
cls = ops.stream_pipe_classes(cands)
same = [s for s, c in zip(cands, cls) if c == cls[0]]
other = [s for s, c in zip(cands, cls) if c != cls[0]]
main, sides_same, sides_other = same[0], same[1:3], other[:2]
print("classes", cls, "-> main + %d side candidates of its class" % len(sides_same))
for recipe, cfg, limits in (("S30k", indoor_config(), synthetic.LIMITS["S30k"]), ("K120k", kitti_config(), synthetic.LIMITS["K120k"])):
    pool = []
    for s in range(4):
        a, b = synthetic.pair("S30k", s) if recipe == "S30k" else synthetic.slab_pair(120000, s)
        pool.append((torch.from_numpy(np.concatenate([a, b])).to(dev), torch.tensor([len(a), len(b)], dtype=torch.int32, device=dev)))
    for k in (1, 4) if recipe == "S30k" else (1, 3):
        for name, sides in (("in line", (None, None)), ("1 side stream (same class)", (sides_same[0], None)),
                            ("2 side streams (same class)", tuple(sides_same[:2])), ("2 side streams (other classes)", tuple(sides_other))):
            if sides[0] is None and name != "in line":
                continue
            nat = NativePyramid(cfg, limits, "auto")
            nat.set_side_streams(*sides)
            ts = []
            with torch.cuda.stream(main):
                for it in range(12):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    if k == 1:
                        nat.build(*pool[it % 4])
                    else:
                        nat.build([p for p, _ in pool[:k]], [l for _, l in pool[:k]], group=2)
                    main.synchronize()
                    for s_ in sides:
                        if s_ is not None:
                            s_.synchronize()
                    ts.append(1e3 * (time.perf_counter() - t0))
            ts = sorted(ts[2:])
            print(f"{recipe}: {k} pair(s) per chain, {name:32s}: median {ts[len(ts) // 2]:6.2f} ms  min {ts[0]:6.2f}  ({ts[len(ts) // 2] / k:5.2f} ms per pair)")
