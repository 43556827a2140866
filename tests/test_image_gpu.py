"""GPU: PCR-CG's image-feature injection (ref:models/architectures.py:195-514) -- the HIP gather /
scatter of per-pixel 2-D features into the [N,129] point features and the forward that consumes them -- against the
UNMODIFIED reference model run with a stand-in 2-D backbone (tests/golden/image_mini.pt)."""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as MR
from pcrcg_amd import indoor_config
from pcrcg_amd.architectures import KPFCNN

pytestmark = pytest.mark.gpu


def _to(v, dev):
    if isinstance(v, list):
        return [t.to(dev) if isinstance(t, torch.Tensor) else t for t in v]
    return v.to(dev) if isinstance(v, torch.Tensor) else v


@pytest.mark.parametrize("img_num", [1, 2, 3])
def test_image_feature_forward_vs_reference(cuda, golden_dir, img_num):
    gold = torch.load(os.path.join(golden_dir, "image_mini.pt"))[f"img{img_num}"]
    col = torch.load(os.path.join(golden_dir, "collate_mini.pt"))["batch"]
    cfg = indoor_config(first_feats_dim=32, gnn_feats_dim=64, image_feature=True, img_num=img_num, in_feats_dim=129)
    torch.manual_seed(gold["seed"])
    np.random.seed(gold["seed"])
    net = KPFCNN(cfg).eval()
    for k, v in gold["weights_check"].items():          # same seeds -> the reference's weights, bit for bit
        assert torch.equal(net.state_dict()[k], v), k
    net = net.to(cuda)
    n_src = int(col["stack_lengths"][0][0])
    batch = {k: _to(v, cuda) for k, v in col.items()}
    batch["src_pcd_raw"], batch["tgt_pcd_raw"] = batch["points"][0][:n_src], batch["points"][0][n_src:]
    for k, v in gold["inputs"].items():                  # precomputed 2-D feature maps, indices, valid maps
        batch[k] = v.to(cuda)
    x = net.image_features(batch)
    assert x.shape == (batch["points"][0].shape[0], 129)
    assert torch.equal(x[::gold["x_stride"]].cpu(), gold["x_rows"])          # copies and one multiply: bit-exact
    with torch.no_grad():
        out = net(batch)
        out_ops = net.forward_ops({**batch, "features": x})
    for k, want in gold["outputs"].items():
        assert MR.rel_err(out[k].cpu(), want) < 1e-4, k
        assert MR.rel_err(out_ops[k].cpu(), want) < 1e-4, k
    # the same maps through a `backbone2d` callable, as the reference passes them (forward(batch, backbone2d))
    batch2 = {k: v for k, v in batch.items() if not k.endswith("_feature2d")}
    lookup = {}
    for s in ("src", "tgt"):
        for i in range(1, img_num + 1):
            color = torch.full((3, 2, 2), float(len(lookup)), device=cuda)       # a tag the stand-in backbone recognises
            batch2[f"{s}_color{i}"] = color
            lookup[float(len(lookup))] = batch[f"{s}{i}_feature2d"]
    with torch.no_grad():
        out2 = net(batch2, backbone2d=lambda c: lookup[float(c.flatten()[0])].unsqueeze(0))
    for k in gold["outputs"]:
        assert torch.equal(out2[k], out[k]) or MR.rel_err(out2[k].cpu(), out[k].cpu()) < 1e-5, k
