"""Train-step throughput of the 8f rank-1 row (config #3: S30k 3DLoMatch-shaped pair, full-width model, fp32):
pyramid + labels are built once, then K x (differentiable forward -> MetricLoss -> backward -> SGD step).
Secondary measurement (bench.py stays the forward benchmark BASELINE.json names).  GPU box only.

  python scripts/bench_train.py [--steps 10] [--warmup 2] [--recipe S30k]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_train.py
        (configs[3]: every rank trains on its own pair, ONE RCCL all-reduce of the flat gradient bucket per step;
         value = pairs/s over all ranks, MAX of the per-rank times)"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcrcg_amd import indoor_config, synthetic  # noqa: E402
from pcrcg_amd.architectures import KPFCNN  # noqa: E402
from pcrcg_amd.config import Config  # noqa: E402
from pcrcg_amd.correspondences import get_correspondences  # noqa: E402
from pcrcg_amd.loss import MetricLoss  # noqa: E402
from pcrcg_amd.pyramid import collate_fn_descriptor  # noqa: E402
from pcrcg_amd.trainer import Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--recipe", default="S30k")
args = ap.parse_args()

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="nccl", device_id=dev)
cfg = indoor_config()
torch.manual_seed(0)
np.random.seed(0)
net = KPFCNN(cfg).to(dev)
loss = MetricLoss(Config(pos_margin=0.1, neg_margin=1.4, pos_radius=0.0375, safe_radius=0.1,
                         matchability_radius=0.05, max_points=256))
trainer = Trainer(net, loss)

src, tgt, rot, trans = synthetic.lomatch_pair(args.recipe, rank, overlap=0.2)     # one pair per rank
tsfm = np.eye(4)
tsfm[:3, :3], tsfm[:3, 3] = rot, trans.flatten()
src_d, tgt_d = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
get_correspondences(src_d, tgt_d, tsfm, 0.0375)        # (first call: library load, allocator warm-up)
torch.cuda.synchronize()
t0 = time.perf_counter()
corr = get_correspondences(src_d, tgt_d, tsfm, 0.0375)
torch.cuda.synchronize()
t_corr = time.perf_counter() - t0
item = dict(src_pcd=src, tgt_pcd=tgt, src_feats=np.ones((len(src), 1), np.float32),
            tgt_feats=np.ones((len(tgt), 1), np.float32), rot=rot, trans=trans, correspondences=corr, sample=0)
limits = synthetic.LIMITS.get(args.recipe, [43, 42, 47, 43])
inputs = collate_fn_descriptor([item], cfg, limits, device=dev)

stats = None
for _ in range(args.warmup):
    stats = trainer.train_step(inputs)
torch.cuda.synchronize()
if dist is not None:
    dist.barrier()
torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter()
for _ in range(args.steps):
    stats = trainer.train_step(inputs)
torch.cuda.synchronize()
if dist is not None:
    dist.barrier()
dt = (time.perf_counter() - t0) / args.steps
if dist is not None:
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    same = torch.stack([p.detach().flatten()[:64].double().sum() for p in net.parameters()]).sum().reshape(1)
    lo, hi = same.clone(), same.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert float(lo) == float(hi), "replicas diverged"
    dist.destroy_process_group()
if rank == 0:
  print(json.dumps({
    "metric": "train pairs/s (fwd + MetricLoss + bwd + gradient all-reduce + SGD), 1 pair/rank/step",
    "value": round(world / dt, 3), "n_gpus": world,
    "ms_per_step": round(dt * 1e3, 2), "recipe": args.recipe, "points": [len(src), len(tgt)],
    "correspondences": int(corr.shape[0]), "get_correspondences_ms": round(t_corr * 1e3, 1),
    "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
    "parameters": sum(p.numel() for p in net.parameters()), "last_stats": {k: round(v, 4) for k, v in stats.items()},
}))
