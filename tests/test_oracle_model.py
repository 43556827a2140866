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


def _image_list(inp, img_num, with_valid):
    images = []
    for side in ("src", "tgt"):
        for i in range(img_num, 0, -1):
            images.append(dict(fmap=inp[f"{side}{i}_feature2d"], inds2d=inp[f"{side}{i}_inds2d"], inds3d=inp[f"{side}{i}_inds3d"],
                               target=side == "tgt", valid=inp.get(f"{side}_valid_map{i}") if with_valid else None))
    return images


def test_image_feature_injection_vs_reference(golden_dir, batch):
    """oracle restatement of ref:models/architectures.py:195-514 against the matrix the unmodified reference model
    fed to its first block (tests/golden/image_mini.pt, scripts/make_golden_image.py): bit-exact."""
    gold = torch.load(os.path.join(golden_dir, "image_mini.pt"))
    n_src = int(batch["stack_lengths"][0][0])
    n = int(batch["points"][0].shape[0])
    for img_num in (1, 2, 3):
        g = gold[f"img{img_num}"]
        x = MR.inject_image_features(n, n_src, _image_list(g["inputs"], img_num, img_num < 3))
        assert torch.equal(x[::g["x_stride"]], g["x_rows"]), img_num
        assert int((x[:, :128] != 1).any(1).sum()) > (1500 if img_num > 1 else 1200)


def test_topk_replay_is_torch_topk():
    """oracle/topk_replay.py (libstdc++ partial_sort / nth_element + sort as PyTorch's CPU top-k kernel drives them)
    against torch.topk itself on rows full of equal values, in both regimes (k * 64 <= n and above)."""
    import numpy as np
    from oracle.topk_replay import topk_smallest_indices
    rng = np.random.RandomState(0)
    for n in (12, 30, 100, 381, 703, 704, 705, 1000, 1936):
        for levels in (2, 3, 5, 17, 1000):
            for _ in range(3):
                v = rng.randint(0, levels, n).astype(np.float32)
                want = torch.from_numpy(v).topk(11, largest=False, sorted=True)[1].tolist()
                assert topk_smallest_indices(v, 11) == want, (n, levels)


def test_modelnet_block_list_against_the_reference(golden_dir):
    """The reference's third block list (ref:configs/models.py:42-57): three levels and a decoder with two consecutive unary
    blocks -- the CPU oracle on the reference's own collate dict and state_dict against the reference model's outputs
    (tests/golden/modelnet_mini.pt, scripts/make_golden_modelnet.py)."""
    mm = torch.load(os.path.join(golden_dir, "modelnet_mini.pt"))
    assert mm["config"]["num_layers"] == 3 and mm["config"]["architecture"][9:] == [
        "nearest_upsample", "unary", "unary", "nearest_upsample", "unary", "last_unary"]
    out = MR.kpfcnn_forward(mm["state_dict"], mm["config"], mm["batch"])
    for k in ("feats_f", "scores_overlap", "scores_saliency"):
        assert out[k].shape == mm["outputs"][k].shape
        assert MR.rel_err(out[k], mm["outputs"][k]) < TOL, k
    # the package's own config helper spells the same model
    from pcrcg_amd import modelnet_config
    mine = modelnet_config(first_feats_dim=32, gnn_feats_dim=64, final_feats_dim=32)
    for key in ("architecture", "num_layers", "first_subsampling_dl", "conv_radius", "first_feats_dim", "gnn_feats_dim",
                "final_feats_dim", "num_kernel_points", "KP_extent", "dgcnn_k", "num_head", "nets"):
        assert mine[key] == mm["config"][key], key
