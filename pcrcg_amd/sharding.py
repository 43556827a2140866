"""Sharding of independent fragment pairs over the GPUs of one node.

The path shards naturally: a pair never interacts with another pair in the forward pass
(ref:datasets/dataloader.py:207 asserts one pair per batch; InstanceNorm statistics and the GNN are
per pair), so pair i simply goes to rank i mod G and NO data-path collective exists.  The only
cross-rank operations are the timing barrier and a MAX reduction of the elapsed time in bench.py.
"""


def shard_pairs(num_pairs, rank, world):
    """Indices of the pairs rank `rank` of `world` processes owns (round robin)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, num_pairs, world))


def pair_seeds_for_rank(steps, rank, world):
    """Weak scaling: at step i rank r processes global pair index i*world + r."""
    return [i * world + rank for i in range(steps)]
