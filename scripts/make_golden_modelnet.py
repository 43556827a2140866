"""Generate tests/golden/modelnet_mini.pt from the UNMODIFIED Python reference (/root/reference, imported through
scripts/ref_import.py).  Runs only in the build container.

The reference's third block list (ref:configs/models.py:42-57, `modelnet`): a THREE-level encoder (two strided blocks) and a
decoder with two consecutive `unary` blocks after the first upsampling; hyper-parameters of ref:configs/test/modelnet.yaml
(num_layers 3, first_subsampling_dl 0.06, conv_radius 2.75) at reduced width (first_feats_dim 32, gnn_feats_dim 64,
final_feats_dim 32; the shipped widths are 512 / 256 / 96).  Input: a ModelNet-shaped synthetic pair -- two partial views
(~70 %) of 1024 points on a unit-scale closed surface, the second one rotated -- through the reference's own
collate_fn_descriptor with fixed neighbourhood limits; output: the collate dict, the state_dict, the model's outputs.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

REPO = ref_import.REPO
OUT = os.path.join(REPO, "tests", "golden")
LIMITS = [24, 30, 32]


def modelnet_pair(seed=0, n=1024, keep=0.7):
    """Two partial views of one closed surface in [-1, 1]^3 (ref:datasets/modelnet.py crops a half-space, keeps 70 %)."""
    rng = np.random.RandomState(seed)
    u, v = rng.rand(n) * 2 * np.pi, rng.rand(n) * 2 * np.pi
    r_major, r_minor = 0.6, 0.25 + 0.08 * np.sin(3 * u)          # a torus with a wavy tube: no symmetry to speak of
    p = np.stack([(r_major + r_minor * np.cos(v)) * np.cos(u), (r_major + r_minor * np.cos(v)) * np.sin(u), r_minor * np.sin(v)], 1)
    p += (rng.rand(n, 3) - 0.5) * 0.01
    views = []
    for _ in range(2):
        d = rng.randn(3)
        d /= np.linalg.norm(d)
        order = np.argsort(p @ d)
        views.append(p[order[:int(n * keep)]])
    ang = 0.6
    rot = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    return views[0].astype(np.float32), (views[1] @ rot.T + np.array([0.2, -0.1, 0.05])).astype(np.float32)


def main():
    ref_import.setup()
    from datasets.dataloader import collate_fn_descriptor
    from models.architectures import KPFCNN
    cfg = ref_import.indoor_config(_yaml="configs/test/modelnet.yaml", img_num=0, init_mode="", node_overlap=False,
                                   quaternion=False, first_feats_dim=32, gnn_feats_dim=64, final_feats_dim=32)
    assert cfg["architecture"].count("unary") == 3 and cfg["num_layers"] == 3
    src, tgt = modelnet_pair(0)
    corr = torch.stack([torch.arange(0, 50), torch.arange(0, 50)], 1)
    item = dict(rot=np.eye(3, dtype=np.float32), trans=np.zeros((3, 1), np.float32), correspondences=corr, sample=0,
                src_pcd=src, tgt_pcd=tgt, src_feats=np.ones((len(src), 1), np.float32), tgt_feats=np.ones((len(tgt), 1), np.float32))
    batch = collate_fn_descriptor([item], cfg, LIMITS)
    torch.manual_seed(0)
    np.random.seed(0)
    model = KPFCNN(cfg).eval()
    with torch.no_grad():
        out = model(batch)
    keep = ("points", "neighbors", "pools", "upsamples", "features", "stack_lengths")
    cfg_plain = {k: v for k, v in cfg.items() if isinstance(v, (int, float, str, bool, list))}
    torch.save({"config": cfg_plain, "limits": LIMITS, "src": torch.from_numpy(src), "tgt": torch.from_numpy(tgt),
                "batch": {k: batch[k] for k in keep}, "state_dict": {k: v.clone() for k, v in model.state_dict().items()},
                "outputs": {k: v.clone() for k, v in out.items()}}, os.path.join(OUT, "modelnet_mini.pt"))
    print("levels", [tuple(p.shape) for p in batch["points"]], "tables", [tuple(t.shape) for t in batch["neighbors"]],
          "decoder", cfg["architecture"][9:], "params", sum(p.numel() for p in model.parameters()))
    print("modelnet_mini.pt", os.path.getsize(os.path.join(OUT, "modelnet_mini.pt")))


if __name__ == "__main__":
    main()
