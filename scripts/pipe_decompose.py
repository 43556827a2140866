"""Where does the pipelined step time go?  Measures, on one GPU and the S30k workload:
  front-only : pyramids/s of the front-end worker alone
  model-only : forwards/s over one prebuilt batch on 1, 2, 3 model streams
  both       : the bench configuration
Run on the GPU box: python scripts/pipe_decompose.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcrcg_amd import indoor_config, kitti_config, synthetic  # noqa: E402
from pcrcg_amd.architectures import KPFCNN  # noqa: E402
from pcrcg_amd.pipeline import PairPipeline  # noqa: E402

K = int(os.environ.get("K", 40))
dev = torch.device("cuda:0")
RECIPE = os.environ.get("RECIPE", "S30k")          # S30k | K120k
cfg = kitti_config() if RECIPE == "K120k" else indoor_config()
limits = synthetic.LIMITS[RECIPE]
torch.manual_seed(0)
net = KPFCNN(cfg).to(dev).eval()
pairs = []
for s in range(4):
    a, b = synthetic.slab_pair(120000, 100 + s) if RECIPE == "K120k" else synthetic.pair("S30k", 100 + s)
    pts = torch.from_numpy(__import__("numpy").concatenate([a, b])).to(dev)
    lens = torch.tensor([len(a), len(b)], dtype=torch.int32, device=dev)
    pairs.append((pts, lens))


def front_only():
    pipe = PairPipeline(net, cfg, limits, dev)
    for i in range(4):
        pipe.request(*pairs[i % 4])
    pipe.drain()
    t0 = time.perf_counter()
    for i in range(K):
        pipe.request(*pairs[i % 4])
    pipe.drain()
    dt = time.perf_counter() - t0
    pipe.close()
    return K / dt


def model_only(nstreams):
    pipe = PairPipeline(net, cfg, limits, dev, model_streams=nstreams)
    pipe.request(*pairs[0])
    prepared = pipe.next_prepared()
    for _ in range(4):
        pipe.run(prepared)
    pipe.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        pipe.run(prepared)
    th = time.perf_counter() - t0
    pipe.synchronize()
    dt = time.perf_counter() - t0
    pipe.close()
    return K / dt, th / K * 1e3


print(f"front-only  {front_only():7.1f} pyramids/s")
for n in (1, 2, 3):
    r, h = model_only(n)
    print(f"model-only  {r:7.1f} forwards/s on {n} stream(s), host enqueue {h:.2f} ms/forward")
