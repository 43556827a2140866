"""CPU, world_size 2 over gloo: the N>1 protocol of bench.py -- pairs sharded over ranks with no
data-path collective, a barrier on both sides of the timed region and a MAX reduction of the
elapsed time."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pcrcg_amd import sharding, synthetic


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = sharding.pair_seeds_for_rank(steps, rank, world)
    dist.barrier()
    checks = []
    for s in seeds:                       # stand-in workload: a digest of this rank's own pairs
        src, tgt = synthetic.pair("mini", s)
        checks.append(float(src.sum() + tgt.sum()))
    dist.barrier()
    elapsed = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, seeds)
    q.put((rank, seeds, checks, float(elapsed.item()), gathered))
    dist.destroy_process_group()


def test_two_rank_sharding_protocol():
    world, steps = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_seeds = sorted(res[0][1] + res[1][1])
    assert all_seeds == list(range(world * steps))                  # every pair exactly once
    assert res[0][3] == res[1][3] == 0.2                            # MAX over ranks
    assert res[0][4] == res[1][4]                                   # both ranks saw the same partition
    for rank, seeds, checks, _, _ in res:                           # ranks worked on their own pairs only
        exp = [float(np.sum(synthetic.pair("mini", s)[0]) + np.sum(synthetic.pair("mini", s)[1])) for s in seeds]
        assert np.allclose(checks, exp)


def _train_worker(rank, world, port, q, overlap=0):
    """Data-parallel optimisation block of pcrcg_amd.trainer.Trainer on a stand-in model (the kernels need a
    GPU; the bucket / all-reduce / skip protocol does not)."""
    from pcrcg_amd.trainer import Trainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)                                   # identical replicas
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
    trainer = Trainer(model, desc_loss=None, lr=0.1, momentum=0.9, weight_decay=0.0, overlap_chunks=overlap)
    g = torch.Generator().manual_seed(100 + rank)          # each rank has its own pair
    x, y = torch.randn(16, 6, generator=g), torch.randn(16, 2, generator=g)
    out = []
    for step in range(3):
        loss = ((model(x) - y) ** 2).mean()
        if step == 1 and rank == 1:
            loss = loss * float("inf")                     # one rank overflows: BOTH must skip this step
        trainer.bucket.arm(True)                           # overlap mode: slices are all-reduced from inside backward
        loss.backward()                                    # lands in the flat bucket through the .grad views
        ok = trainer.optimizer_step()
        out.append((ok, [p.detach().clone() for p in model.parameters()]))
    q.put((rank, [(ok, [t.numpy() for t in ps]) for ok, ps in out], trainer.skipped_steps))
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [0, 3])
def test_two_rank_gradient_all_reduce_and_global_skip(overlap):
    """overlap = 3: the bucket is exchanged as three slices launched from inside backward (asynchronously) --
    same replicas, same updates, same skip decision as the single all-reduce."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, steps0, skipped0), (_, steps1, skipped1) = res
    assert skipped0 == skipped1 == 1
    assert [ok for ok, _ in steps0] == [ok for ok, _ in steps1] == [True, False, True]
    for (_, p0), (_, p1) in zip(steps0, steps1):            # replicas stay bit-identical
        for a, b in zip(p0, p1):
            assert np.array_equal(a, b)
    # step 0 equals single-process SGD on the MEAN of the two ranks' gradients
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
    loss = 0
    for rank in range(world):
        g = torch.Generator().manual_seed(100 + rank)
        x, y = torch.randn(16, 6, generator=g), torch.randn(16, 2, generator=g)
        loss = loss + ((ref(x) - y) ** 2).mean() / world
    loss.backward()
    with torch.no_grad():
        for p, got in zip(ref.parameters(), steps0[0][1]):
            assert np.allclose((p - 0.1 * p.grad).numpy(), got, atol=1e-6)
    for a, b in zip(steps0[0][1], steps0[1][1]):            # the skipped step changed nothing
        assert np.array_equal(a, b)
