"""Train-step throughput of the 8f rank-1 row (configs[2] / configs[3]: S30k 3DLoMatch-shaped pair, full-width model,
fp32): pyramid + labels are built once, then K x (differentiable forward -> MetricLoss -> backward -> gradient
all-reduce -> SGD step).  Secondary measurement (bench.py stays the forward benchmark BASELINE.json names; its
`secondary.train_step` block runs the 1-GPU case of this file).  GPU box only, except --launcher-dry-run.

  python scripts/bench_train.py [--gpus N] [--steps 10] [--warmup 2] [--recipe S30k]
        (--gpus N > 1 with no rank environment: this process only starts N fresh ranks, pcrcg_amd/launcher.py)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_train.py --gpus N
        (configs[3]: every rank trains on its own pair, ONE RCCL all-reduce of the flat 29.7 M-element gradient
         bucket per step; value = pairs/s over all ranks, MAX of the per-rank times; the exchange alone is timed
         separately -- `allreduce.ms` -- because inside the step it overlaps the backward pass)"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pcrcg_amd import launcher  # noqa: E402  (host logic only)


def _gpus(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


if __name__ == "__main__" and launcher.is_parent(_gpus(sys.argv[1:])):
    # before anything that could initialise the GPU: the parent only starts the ranks and relays rank 0's line
    sys.exit(launcher.launch(os.path.abspath(__file__), sys.argv[1:], _gpus(sys.argv[1:]),
                             dry_run="--launcher-dry-run" in sys.argv[1:]))
RANK_CPUS = launcher.apply_rank_affinity()

import numpy as np  # noqa: E402
import torch  # noqa: E402

if RANK_CPUS:                              # pinned to a CPU set: torch's host thread pool must not exceed it
    torch.set_num_threads(max(1, len(RANK_CPUS)))


def time_allreduce(dist, flat, reps, sync):
    """The step's one collective alone: all-reduce(sum) of the flat gradient bucket, `reps` times back to back between
    two barriers -> average milliseconds (MAX over ranks)."""
    if dist is None:
        return None
    keep = flat.clone()
    dist.all_reduce(flat)
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(flat)
    sync()
    dist.barrier()
    ms = (time.perf_counter() - t0) / reps * 1e3
    t = torch.tensor([ms], dtype=torch.float64, device=flat.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    flat.copy_(keep)
    return float(t.item())


def dry_run(args, rank, world):
    """The N-rank protocol over gloo on the CPU with a stand-in model: Trainer's bucket / all-reduce / global-skip / SGD
    block is the real code (it needs no GPU; the network's kernels do)."""
    import torch.distributed as dist
    from pcrcg_amd.trainer import Trainer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29519")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
    trainer = Trainer(model, desc_loss=None, lr=0.1, momentum=0.9, weight_decay=0.0)
    g = torch.Generator().manual_seed(100 + rank)          # one pair per rank
    x, y = torch.randn(16, 6, generator=g), torch.randn(16, 2, generator=g)

    def step():
        trainer.bucket.arm(True)
        ((model(x) - y) ** 2).mean().backward()
        return trainer.optimizer_step()

    for _ in range(args.warmup):
        step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dist.barrier()
    own = (time.perf_counter() - t0) / args.steps
    t = torch.tensor([own], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ar_ms = time_allreduce(dist, trainer.bucket.flat, 5, lambda: None)
    same = torch.cat([p.detach().flatten().double() for p in model.parameters()]).sum().reshape(1)
    lo, hi = same.clone(), same.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    fields = launcher.rank_fields(dist, world, rank, round(1.0 / own, 3), RANK_CPUS)
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({
            "metric": "train pairs/s (fwd + MetricLoss + bwd + gradient all-reduce + SGD), 1 pair/rank/step",
            "value": round(world / dt, 3), "unit": "fragment-pairs/s", "n_gpus": world, "ranks_seen": fields["ranks_seen"],
            "per_rank_pairs_per_s": fields["per_rank_value"], "per_rank_cpus": fields["per_rank_cpus"],
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
            "allreduce": {"elements": int(trainer.bucket.flat.numel()), "bytes": int(trainer.bucket.flat.numel()) * 4,
                          "ms": round(ar_ms, 4), "backend": "gloo"},
            "replicas_identical": float(lo) == float(hi),
            "config": {"workload": "launcher dry run: stand-in two-layer model on the CPU over gloo; NOT a measurement"}}),
            flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--recipe", default="S30k")
    ap.add_argument("--launcher-dry-run", action="store_true")
    args = ap.parse_args()

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launcher_dry_run:
        return dry_run(args, rank, world)

    from pcrcg_amd import indoor_config, synthetic
    from pcrcg_amd.architectures import KPFCNN
    from pcrcg_amd.config import Config
    from pcrcg_amd.correspondences import get_correspondences
    from pcrcg_amd.loss import MetricLoss
    from pcrcg_amd.pyramid import collate_fn_descriptor
    from pcrcg_amd.trainer import Trainer

    if local >= torch.cuda.device_count():
        print("bench_train.py: rank %d wants device %d, %d visible" % (rank, local, torch.cuda.device_count()), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("PCRCG_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29520")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
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
    own = (time.perf_counter() - t0) / args.steps
    dt, ar_ms, identical = own, None, None
    if dist is not None:
        t = torch.tensor([own], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ar_ms = time_allreduce(dist, trainer.bucket.flat, 10, torch.cuda.synchronize)
        same = torch.stack([p.detach().flatten()[:64].double().sum() for p in net.parameters()]).sum().reshape(1)
        lo, hi = same.clone(), same.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        identical = float(lo) == float(hi)
        assert identical, "replicas diverged"
    fields = launcher.rank_fields(dist, world, rank, round(1.0 / own, 3), RANK_CPUS, device=dev)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        nel = int(trainer.bucket.flat.numel())
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)      # RCCL's banner goes through C stdio: keep the JSON line last
        print(json.dumps({
            "metric": "train pairs/s (fwd + MetricLoss + bwd + gradient all-reduce + SGD), 1 pair/rank/step",
            "value": round(world / dt, 3), "unit": "fragment-pairs/s", "n_gpus": world, "ranks_seen": fields["ranks_seen"],
            "per_rank_pairs_per_s": fields["per_rank_value"], "per_rank_cpus": fields["per_rank_cpus"],
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "allreduce": {"elements": nel, "bytes": nel * 4, "ms": None if ar_ms is None else round(ar_ms, 3),
                          "backend": "rccl",
                          "GBs_bus": None if not ar_ms else round(2 * (world - 1) / max(world, 1) * nel * 4 / (ar_ms * 1e-3) / 1e9, 1),
                          "note": "the exchange alone, back to back; inside the step it overlaps the backward pass"},
            "replicas_identical": identical,
            "config": {"workload": f"{args.recipe} 3DLoMatch-shaped pair (overlap 0.2), full-width KPFCNN+GCN, one pair per "
                                   "rank per step, pyramid + labels prebuilt",
                       "parallelism": f"data parallel over {world} GPU(s): one all-reduce of the flat gradient bucket per step"},
            "recipe": args.recipe, "points": [len(src), len(tgt)],
            "correspondences": int(corr.shape[0]), "get_correspondences_ms": round(t_corr * 1e3, 1),
            "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            "parameters": sum(p.numel() for p in net.parameters()),
            "last_stats": {k: round(v, 4) for k, v in stats.items()}}), flush=True)


if __name__ == "__main__":
    main()
