"""tests/golden/model_s30k_img129.pt: PCR-CG's SHIPPED configuration (ref:configs/test/indoor.yaml:21-34: image_feature True,
img_num 2, in_feats_dim 129) at full width and full size, from the UNMODIFIED reference (build container only).

The reference's full-width KPFCNN (default initialisation under torch.manual_seed(0) / np.random.seed(0): pcrcg_amd builds
the bit-identical model from the same seeds) on the reference's own collate of the S30k pair 0 (BASELINE.json configs[1],
limits [43,42,47,43]) with the synthetic 2-D inputs of pcrcg_amd.synthetic.image_inputs(seed 0): two 128 x 120 x 160 feature
maps per cloud, 45 % of the points projected per image, valid masks.  The 2-D backbone is outside the path: a stand-in hands
the reference the stored maps (its colour input carries the map's number).  The reference hard-codes `.cuda()` in this
branch; for the run on this GPU-less container Tensor.cuda is made the identity (harness only, nothing of it is shipped).

Stored: every 97th row of the [N, 129] matrix the reference feeds to its first block and of the three outputs, every output
row's projection on four fixed unit vectors, the output means."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from make_golden_scale import STRIDE, row_projections  # noqa: E402

OUT = os.path.join(ref_import.REPO, "tests", "golden")
LIMITS = [43, 42, 47, 43]


def main():
    ref_import.setup()
    from datasets.dataloader import collate_fn_descriptor
    from models.architectures import KPFCNN
    from pcrcg_amd import synthetic as S
    torch.Tensor.cuda = lambda self, *a, **k: self          # the branch calls .cuda() on CPU tensors
    cfg = ref_import.indoor_config(image_feature=True, img_num=2, in_feats_dim=129)
    src, tgt = S.pair("S30k", 0)
    item = dict(rot=np.eye(3, dtype=np.float32), trans=np.zeros((3, 1), np.float32),
                correspondences=torch.stack([torch.arange(0, 50), torch.arange(0, 50)], 1), sample=0, src_pcd=src, tgt_pcd=tgt,
                src_feats=np.ones((len(src), 1), np.float32), tgt_feats=np.ones((len(tgt), 1), np.float32))
    batch = collate_fn_descriptor([item], cfg, LIMITS)
    img = S.image_inputs(len(src), len(tgt), 0, img_num=2)
    maps = []
    for side in ("src", "tgt"):
        for i in (1, 2):
            colour = torch.zeros(3, 2, 2)
            colour[0, 0, 0] = len(maps)
            maps.append(torch.from_numpy(img[f"{side}{i}_feature2d"]))
            batch[f"{side}_color{i}"] = colour
            batch[f"{side}{i}_inds2d"] = torch.from_numpy(img[f"{side}{i}_inds2d"])
            batch[f"{side}{i}_inds3d"] = torch.from_numpy(img[f"{side}{i}_inds3d"])
            batch[f"{side}_valid_map{i}"] = torch.from_numpy(img[f"{side}_valid_map{i}"])
    batch["id_name"] = "S30k-img129"

    def backbone(colour):
        return maps[int(colour[0, 0, 0, 0])].unsqueeze(0)
    torch.manual_seed(0)
    np.random.seed(0)
    model = KPFCNN(cfg).eval()
    seen = {}
    hook = model.encoder_blocks[0].register_forward_pre_hook(lambda m, a: seen.__setitem__("x", a[0].detach().clone()))
    with torch.no_grad():
        out = model(batch, backbone)
    hook.remove()
    x = seen["x"]
    cfg_plain = {k: v for k, v in cfg.items() if isinstance(v, (int, float, str, bool, list))}
    rec = {"config": cfg_plain, "limits": LIMITS, "stride": STRIDE, "x_rows": x[::STRIDE].clone(),
           "x_rows_with_image_features": int((x[:, :128] != 1).any(1).sum()),
           "weights_check": {k: model.state_dict()[k].clone() for k in ("encoder_blocks.0.KPConv.weights",)},
           "levels": [int(p.shape[0]) for p in batch["points"]],
           "rows": {k: v[::STRIDE].clone() for k, v in out.items()},
           "means": {k: float(v.double().mean()) for k, v in out.items()},
           "feats_proj": row_projections(out["feats_f"], 4)}
    torch.save(rec, os.path.join(OUT, "model_s30k_img129.pt"))
    print("x", tuple(x.shape), "rows with image features", rec["x_rows_with_image_features"], "levels", rec["levels"],
          {k: tuple(v.shape) for k, v in out.items()}, os.path.getsize(os.path.join(OUT, "model_s30k_img129.pt")))


if __name__ == "__main__":
    main()
