"""CPU: the device-agnostic pieces of pcrcg_amd.loss.MetricLoss against vectors captured from the
reference's own MetricLoss sub-methods (tests/golden/loss_mini.pt, scripts/make_golden_loss.py)."""
import os

import pytest
import torch

from pcrcg_amd.config import Config
from pcrcg_amd.loss import MetricLoss, square_distance


@pytest.fixture(scope="module")
def golden(golden_dir):
    return torch.load(os.path.join(golden_dir, "loss_mini.pt"), weights_only=False)


def close(a, b, tol=1e-5):
    a, b = float(a), float(b)
    return abs(a - b) <= tol * max(1.0, abs(b))


@pytest.mark.parametrize("case", ["capped", "all"])
def test_pure_sub_methods_match_reference(golden, case):
    loss = MetricLoss(Config(golden["config"]))
    sub = golden["cases"][case]["sub"]
    assert close(loss.get_circle_loss(sub["coords_dist"], sub["feats_dist"]), sub["circle_loss"])
    assert close(loss.get_recall(sub["coords_dist"], sub["feats_dist"]), sub["recall"])
    inputs = golden["cases"][case]["inputs"]
    bce, prec, rec = loss.get_weighted_bce_loss(inputs["scores_overlap"], sub["bce_gt"])
    assert close(bce, sub["bce_loss"]) and close(prec, sub["bce_precision"]) and close(rec, sub["bce_recall"])


def test_degenerate_labels_and_defaults(golden):
    loss = MetricLoss(Config(golden["config"]))
    assert loss.log_scale == 16 and loss.pos_optimal == 0.1 and loss.neg_optimal == 1.4   # yaml log_scale ignored
    p = torch.tensor([0.2, 0.4, 0.1])
    _, prec, rec = loss.get_weighted_bce_loss(p, torch.zeros(3))
    assert float(prec) == 0.0 and float(rec) == 0.0      # sklearn reports 0 for 0/0
    d = square_distance(torch.zeros(1, 2, 3), torch.zeros(1, 2, 3))
    assert float(d.min()) == pytest.approx(1e-12)         # clamp of ref:lib/utils.py:96
