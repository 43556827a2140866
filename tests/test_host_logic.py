"""CPU: host-side logic of the product (no GPU): configuration / architecture plan, state_dict
compatibility with the reference, shim argument checking, sharding, and the rule that the product
never reaches into oracle/ nor falls back to the CPU."""
import os
import re

import numpy as np
import pytest
import torch

from pcrcg_amd import indoor_config, kitti_config, ops, sharding, synthetic
from pcrcg_amd.architectures import KPFCNN
from pcrcg_amd.cpp_wrappers.cpp_neighbors import radius_neighbors
from pcrcg_amd.cpp_wrappers.cpp_subsampling import grid_subsampling
from pcrcg_amd.kernel_points import load_kernels
from pcrcg_amd.pyramid import _layer_plan, build_pyramid

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_matches_reference_fixture(golden_dir):
    mm = torch.load(os.path.join(golden_dir, "model_mini.pt"))
    net = KPFCNN(indoor_config(first_feats_dim=32, gnn_feats_dim=64))
    ours = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    ref = {k: tuple(v.shape) for k, v in mm["state_dict"].items()}
    assert ours == ref
    net.load_state_dict(mm["state_dict"], strict=True)
    assert net.encoder_skips == [2, 5, 8, 11] and net.decoder_concats == [1, 3, 5]


def test_full_width_parameter_count():
    net = KPFCNN(indoor_config())
    assert sum(p.numel() for p in net.parameters()) == 29677811      # SURVEY.md section 6 (probe of the reference)
    sd = net.state_dict()
    assert tuple(sd["decoder_blocks.1.mlp.weight"].shape) == (257, 1538)
    assert tuple(sd["decoder_blocks.5.mlp.weight"].shape) == (34, 384)
    assert tuple(sd["bottle.weight"].shape) == (512, 2048, 1)


def test_layer_plan_radii():
    plan = _layer_plan(indoor_config())
    assert [round(p["r_conv"], 6) for p in plan] == [0.0625, 0.125, 0.25, 0.5]
    assert [round(p["dl"], 6) for p in plan[:3]] == [0.05, 0.1, 0.2]
    assert [p["pooled"] for p in plan] == [True, True, True, False]
    k = _layer_plan(kitti_config())
    assert abs(k[0]["r_conv"] - 1.275) < 1e-9 and abs(k[0]["dl"] - 0.6) < 1e-9


def test_kernel_points(golden_dir):
    """load_kernels reproduces ref:kernels/kernel_points.py:388-470 under the same np.random stream: a seed-0
    model built here equals the reference's seed-0 model (fixture state_dict) bit for bit -- kernel points and,
    because the constructors draw from torch's generator in the same order, every weight."""
    from pcrcg_amd.kernel_points import DISPOSITION_15_CENTER_3D
    assert DISPOSITION_15_CENTER_3D.shape == (15, 3) and DISPOSITION_15_CENTER_3D.dtype == np.float64
    assert (DISPOSITION_15_CENTER_3D[0] == 0).all()
    np.random.seed(0)
    kp = load_kernels(0.0625)
    assert kp.shape == (15, 3) and kp.dtype == np.float32
    mm = torch.load(os.path.join(golden_dir, "model_mini.pt"))
    torch.manual_seed(0)
    np.random.seed(0)
    sd = KPFCNN(indoor_config(first_feats_dim=32, gnn_feats_dim=64)).state_dict()
    kps = [k for k in mm["state_dict"] if k.endswith("kernel_points")]
    assert len(kps) == 11
    for k, v in mm["state_dict"].items():
        assert torch.equal(sd[k], v), k


def test_no_cpu_fallback_and_no_oracle_in_product():
    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.gemm(x, x.t())
    with pytest.raises(RuntimeError, match="HIP device"):
        build_pyramid(x, torch.tensor([4], dtype=torch.int32), indoor_config(), [4, 4, 4, 4])
    pkg = os.path.join(REPO, "pcrcg_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "/root/reference" not in src, f


def test_shim_argument_checks_match_reference_messages():
    pts = np.zeros((10, 3), np.float32)
    with pytest.raises(RuntimeError, match=r"query.shape is not \(N, 3\)"):
        radius_neighbors.batch_query(pts[:, :2], pts, [10], [10], radius=0.1)
    with pytest.raises(RuntimeError, match=r"support.shape is not \(N, 3\)"):
        radius_neighbors.batch_query(pts, pts.reshape(-1), [10], [10], radius=0.1)
    with pytest.raises(RuntimeError, match="different for queries and supports"):
        radius_neighbors.batch_query(pts, pts, [10], [5, 5], radius=0.1)
    with pytest.raises(TypeError):
        radius_neighbors.batch_query(pts, pts, [10], [10], 0.1)            # radius is keyword-only ("OOOO|$f")
    with pytest.raises(RuntimeError, match=r"points.shape is not \(N, 3\)"):
        grid_subsampling.subsample_batch(pts[:, :2], [10], sampleDl=0.1)
    with pytest.raises(RuntimeError, match="Valid method names"):
        grid_subsampling.subsample_batch(pts, [10], sampleDl=0.1, method="median")
    with pytest.raises(RuntimeError, match="float32"):
        grid_subsampling.subsample_batch([["a", "b", "c"]], [1], sampleDl=0.1)


def test_sharding_partitions_pairs():
    for world in (1, 2, 8):
        owned = [sharding.shard_pairs(64, r, world) for r in range(world)]
        assert sorted(sum(owned, [])) == list(range(64))
        assert max(map(len, owned)) - min(map(len, owned)) <= 1
        seeds = [sharding.pair_seeds_for_rank(5, r, world) for r in range(world)]
        assert sorted(sum(seeds, [])) == list(range(5 * world))
    with pytest.raises(ValueError):
        sharding.shard_pairs(4, 3, 2)


def test_synthetic_recipes_are_reproducible():
    a1, b1 = synthetic.pair("mini", 0)
    a2, b2 = synthetic.pair("mini", 0)
    assert a1.dtype == np.float32 and a1.shape == (1500, 3)
    assert (a1 == a2).all() and (b1 == b2).all() and not (a1 == b1).all()
    s, t, rot, trans = synthetic.lomatch_pair("mini", 1, 0.2)
    assert s.shape == t.shape and abs(np.linalg.det(rot) - 1) < 1e-5


def test_probabilistic_sample_draws_like_the_reference():
    """Same host generator, same call as ref:lib/tester.py:155-158 -> same indices."""
    import numpy as np
    import torch
    from pcrcg_amd.tester import probabilistic_sample
    g = torch.Generator().manual_seed(0)
    pcd, feats = torch.rand(700, 3, generator=g), torch.rand(700, 32, generator=g)
    scores = torch.rand(700, generator=g) * torch.rand(700, generator=g)
    np.random.seed(11)
    p2, f2, idx = probabilistic_sample(pcd, feats, scores, 200)
    np.random.seed(11)
    want = np.random.choice(np.arange(700), size=200, replace=False, p=(scores / scores.sum()).numpy().flatten())
    assert np.array_equal(idx, want) and len(set(idx.tolist())) == 200
    assert torch.equal(p2, pcd[want]) and torch.equal(f2, feats[want])
    p3, f3, none = probabilistic_sample(pcd, feats, scores, 700)
    assert none is None and p3 is pcd and f3 is feats


def test_engine_job_sizes():
    """pairstream.PairStreams.job_sizes: the pairs of one front-end build go to the model streams in jobs of up to
    `pairs_per_forward`, or one pair per job while the engine fills up / runs empty (every pair exactly once, in order)."""
    from pcrcg_amd.pairstream import PairStreams
    assert PairStreams.job_sizes(2, 2, False) == [2]
    assert PairStreams.job_sizes(2, 2, True) == [1, 1]
    assert PairStreams.job_sizes(4, 3, False) == [3, 1]
    assert PairStreams.job_sizes(3, 2, False) == [2, 1]
    assert PairStreams.job_sizes(1, 4, True) == [1]
    assert PairStreams.job_sizes(4, 1, False) == [1, 1, 1, 1]
    for n in range(1, 5):
        for per in range(1, 5):
            for one in (False, True):
                assert sum(PairStreams.job_sizes(n, per, one)) == n
