"""GPU: the training-side rows through the C ABI of include/pcrcg_train.h."""
import os

import numpy as np
import pytest
import torch

from pcrcg_amd import ops
from pcrcg_amd.config import Config
from pcrcg_amd.loss import MetricLoss

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


@pytest.mark.parametrize("n,m,c", [(300, 257, 32), (1000, 3000, 32), (65, 129, 64), (40, 50, 20), (1, 1, 32)])
def test_feature_argmax_matches_matmul_max(cuda, n, m, c):
    g = torch.Generator().manual_seed(n + m)
    a = torch.randn(n, c, generator=g).to(cuda)
    b = torch.randn(m, c, generator=g).to(cuda)
    arg, best = ops.feature_argmax(a, b, want_best=True)
    ref = a.double() @ b.double().t()
    ref_best, ref_arg = ref.max(1)
    assert rel(best, ref_best) < 1e-5
    assert torch.equal(arg, ref_arg)
    # strided operands (column slices of wider matrices)
    wide_a = torch.randn(n, c + 8, generator=g).to(cuda)
    wide_b = torch.randn(m, c + 4, generator=g).to(cuda)
    arg2 = ops.feature_argmax(wide_a[:, :c], wide_b[:, :c])
    assert torch.equal(arg2, (wide_a[:, :c].double() @ wide_b[:, :c].double().t()).argmax(1))


def test_feature_argmax_first_index_wins_ties(cuda):
    a = torch.ones(5, 32, device=cuda)
    b = torch.ones(7, 32, device=cuda)
    assert torch.equal(ops.feature_argmax(a, b), torch.zeros(5, dtype=torch.int64, device=cuda))
    with pytest.raises(RuntimeError):
        ops.feature_argmax(a, b[:0])


@pytest.mark.parametrize("case", ["capped", "all"])
def test_metric_loss_forward_matches_reference(cuda, golden_dir, case):
    """Full MetricLoss.forward on the device against the reference's forward (tests/golden/loss_mini.pt)."""
    gold = torch.load(os.path.join(golden_dir, "loss_mini.pt"), weights_only=False)
    cs = gold["cases"][case]
    loss = MetricLoss(Config(gold["config"]))
    inputs = {k: v.to(cuda) for k, v in cs["inputs"].items()}
    np.random.seed(cs["numpy_seed"])           # the max_points cap draws from the host generator (ref:lib/loss.py:231)
    stats = loss(inputs)
    assert set(stats) == set(cs["expected"])
    for k, want in cs["expected"].items():
        got = float(stats[k])
        assert abs(got - float(want)) <= 1e-4 * max(1.0, abs(float(want))), (k, got, float(want))


def test_metric_loss_is_differentiable(cuda, golden_dir):
    gold = torch.load(os.path.join(golden_dir, "loss_mini.pt"), weights_only=False)
    cs = gold["cases"]["all"]
    loss = MetricLoss(Config(gold["config"]))
    inputs = {k: v.to(cuda) for k, v in cs["inputs"].items()}
    for k in ("src_feats", "tgt_feats", "scores_overlap", "scores_saliency"):
        inputs[k] = inputs[k].clone().requires_grad_(True)
    np.random.seed(cs["numpy_seed"])
    stats = loss(inputs)
    total = stats["circle_loss"] + stats["overlap_loss"] + stats["saliency_loss"]    # ref:lib/trainer.py:255-260
    total.backward()
    for k in ("src_feats", "tgt_feats", "scores_overlap", "scores_saliency"):
        g = inputs[k].grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
