"""tests/golden/image_mini.pt: PCR-CG's image-feature injection (ref:models/architectures.py:195-514) from the
UNMODIFIED reference model (build container only).

The reference KPFCNN is built with image_feature=True, in_feats_dim=129 and img_num = 1 / 2 (valid maps) / 3 (no valid maps)
at the reduced width of the other mini fixtures and run on the `mini` pair's collate with synthetic projections: per
cloud and image a random subset of the points (overlapping between images, so the write order matters), random pixel
coordinates and -- for img_num = 2 -- random valid masks.  The 2-D backbone is a stand-in (a seeded 3x3 convolution to
128 channels: the real ResUNet is outside the path); its feature maps are stored, so the consumers of the fixture need
no backbone.  The reference hard-codes `.cuda()` in this branch; for the run on this GPU-less container Tensor.cuda is
made the identity (harness only, nothing of it is shipped).

Stored per variant: the construction seed (the weights are reproduced from it), the feature maps / indices / valid
maps, every 5th row of the [N,129] matrix the reference feeds to its first block, and the model outputs."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

OUT = os.path.join(ref_import.REPO, "tests", "golden")
H, W = 12, 16


def main():
    ref_import.setup()
    from models.architectures import KPFCNN
    torch.Tensor.cuda = lambda self, *a, **k: self          # the branch calls .cuda() on CPU tensors
    col = torch.load(os.path.join(OUT, "collate_mini.pt"))
    batch0 = col["batch"]
    n_src, n_tgt = (int(v) for v in batch0["stack_lengths"][0])
    out = {}
    for img_num in (1, 2, 3):
        cfg = ref_import.indoor_config(first_feats_dim=32, gnn_feats_dim=64, image_feature=True, img_num=img_num,
                                       in_feats_dim=129)
        torch.manual_seed(10 + img_num)
        np.random.seed(10 + img_num)
        model = KPFCNN(cfg).eval()
        backbone = torch.nn.Conv2d(3, 128, 3, padding=1)
        rng = np.random.RandomState(img_num)
        batch = dict(batch0)
        batch["src_pcd_raw"], batch["tgt_pcd_raw"] = batch0["points"][0][:n_src], batch0["points"][0][n_src:]
        batch["id_name"] = "mini"
        stored = {}
        for side, n in (("src", n_src), ("tgt", n_tgt)):
            for i in range(1, img_num + 1):
                k = int(n * 0.45)
                color = torch.from_numpy(rng.rand(3, H, W).astype(np.float32))
                inds3d = torch.from_numpy(rng.permutation(n)[:k].astype(np.int64))
                inds2d = torch.from_numpy(np.stack([rng.randint(0, W, k), rng.randint(0, H, k)], 1).astype(np.int64))
                batch[f"{side}_color{i}"], batch[f"{side}{i}_inds3d"], batch[f"{side}{i}_inds2d"] = color, inds3d, inds2d
                with torch.no_grad():
                    stored[f"{side}{i}_feature2d"] = backbone(color.unsqueeze(0)).squeeze(0).clone()
                stored[f"{side}{i}_inds3d"], stored[f"{side}{i}_inds2d"] = inds3d, inds2d
                if img_num < 3:
                    valid = torch.from_numpy((rng.rand(W, H) > 0.2).astype(np.float32))
                    batch[f"{side}_valid_map{i}"] = valid
                    stored[f"{side}_valid_map{i}"] = valid
        seen = {}
        hook = model.encoder_blocks[0].register_forward_pre_hook(lambda m, a: seen.__setitem__("x", a[0].detach().clone()))
        with torch.no_grad():
            res = model(batch, backbone)
        hook.remove()
        cfg_plain = {k: v for k, v in cfg.items() if isinstance(v, (int, float, str, bool, list))}
        sd = model.state_dict()
        out[f"img{img_num}"] = {"config": cfg_plain, "seed": 10 + img_num,
                                # the weights are NOT stored: pcrcg_amd.KPFCNN built under the same two seeds is
                                # bit-identical (tests/test_host_logic.py); two tensors are kept as a cross-check
                                "weights_check": {k: sd[k].clone() for k in ("encoder_blocks.0.KPConv.weights",
                                                                             "decoder_blocks.5.mlp.weight")},
                                "inputs": stored, "x_stride": 5, "x_rows": seen["x"][::5].clone(),
                                "outputs": {k: v.clone() for k, v in res.items()}}
        print(img_num, "x", tuple(seen["x"].shape), "rows with image features",
              int((seen["x"][:, :128] != 1).any(1).sum()), {k: tuple(v.shape) for k, v in res.items()})
    torch.save(out, os.path.join(OUT, "image_mini.pt"))


if __name__ == "__main__":
    main()
