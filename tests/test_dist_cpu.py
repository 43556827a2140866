"""CPU, world_size 2 over gloo: the N>1 protocol of bench.py -- pairs sharded over ranks with no
data-path collective, a barrier on both sides of the timed region and a MAX reduction of the
elapsed time."""
import os
import socket

import numpy as np
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
