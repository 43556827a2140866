"""CPU: the oracle's torch restatement of the model (oracle/model_ref.py) against golden vectors
generated from the imported, unmodified reference model (scripts/make_golden_model.py)."""
import os

import pytest
import torch

from oracle import model_ref as MR

TOL = 1e-4


@pytest.fixture(scope="module")
def batch(golden_dir):
    return torch.load(os.path.join(golden_dir, "collate_mini.pt"))["batch"]


def test_kpconv_cases(golden_dir, batch):
    for name, c in torch.load(os.path.join(golden_dir, "kpconv_mini.pt")).items():
        l = c["layer"]
        s = batch["points"][l]
        q = batch["points"][l + 1] if c["strided"] else s
        inds = batch["pools"][l] if c["strided"] else batch["neighbors"][l]
        y = MR.kpconv(q, s, inds, c["x"], c["kernel_points"], c["weights"], c["extent"])
        assert MR.rel_err(y, c["out"]) < 1e-6, name


def test_gcn(golden_dir):
    gc = torch.load(os.path.join(golden_dir, "gcn_mini.pt"))
    sd = {"gnn." + k: v for k, v in gc["state_dict"].items()}
    o0, o1 = MR.gcn(sd, "gnn", ["self", "cross", "self"], gc["c0"], gc["c1"], gc["d0"], gc["d1"], 10, 4)
    assert MR.rel_err(o0, gc["o0"]) < TOL and MR.rel_err(o1, gc["o1"]) < TOL


def test_kpfcnn_outputs_and_intermediates(golden_dir, batch):
    mm = torch.load(os.path.join(golden_dir, "model_mini.pt"))
    out = MR.kpfcnn_forward(mm["state_dict"], mm["config"], batch, return_intermediates=True)
    for k in ("feats_f", "scores_overlap", "scores_saliency"):
        assert out[k].shape == mm["outputs"][k].shape
        assert MR.rel_err(out[k], mm["outputs"][k]) < TOL, k
    for k, v in mm["intermediates"].items():
        assert MR.rel_err(out["_inter"][k], v) < TOL, k


def test_plan_matches_reference_shapes(golden_dir):
    mm = torch.load(os.path.join(golden_dir, "model_mini.pt"))
    pl = MR.plan(mm["config"])
    sd = mm["state_dict"]
    for i, blk in enumerate(pl["encoder"]):
        w = sd[f"encoder_blocks.{i}.KPConv.weights"]
        cin = blk["in_dim"] if "simple" in blk["name"] else blk["out_dim"] // 4
        cout = blk["out_dim"] // 2 if "simple" in blk["name"] else blk["out_dim"] // 4
        assert tuple(w.shape) == (15, cin, cout)
    assert pl["encoder_skips"] == [2, 5, 8, 11] and pl["decoder_concats"] == [1, 3, 5]
