"""Generate tests/golden/{collate,model,kpconv,gcn}_*.pt from the UNMODIFIED Python reference
(/root/reference, imported through scripts/ref_import.py).  Runs only in the build container.

  collate_mini.pt : reference collate_fn_descriptor (ref:datasets/dataloader.py:203-400) output for
                    the `mini` synthetic pair with fixed neighbourhood limits
  model_mini.pt   : state_dict of a reduced-width reference KPFCNN (first_feats_dim 32,
                    gnn_feats_dim 64; the full 29.7 M-parameter model is compared in-container
                    only), its outputs on collate_mini and a few intermediate activations
  kpconv_mini.pt  : reference KPConv.forward (ref:models/blocks.py:229-374) on the mini tables for
                    (Cin,Cout) in {(1,16),(8,8) plain and strided,(64,64)}
  gcn_mini.pt     : reference GCN.forward (ref:models/gcn.py:208-217) on N=(96,80), C=64
  calibration.json: reference calibrate_neighbors (ref:datasets/dataloader.py:402-434) limits
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
MINI_LIMITS = [20, 26, 30, 32]


def mini_cfg():
    return ref_import.indoor_config(first_feats_dim=32, gnn_feats_dim=64)


def ref_collate(src, tgt, cfg, limits):
    from datasets.dataloader import collate_fn_descriptor
    corr = torch.stack([torch.arange(0, 50), torch.arange(0, 50)], 1)
    item = dict(rot=np.eye(3, dtype=np.float32), trans=np.zeros((3, 1), np.float32), correspondences=corr,
                sample=0, src_pcd=src, tgt_pcd=tgt, src_feats=np.ones((len(src), 1), np.float32),
                tgt_feats=np.ones((len(tgt), 1), np.float32))
    return collate_fn_descriptor([item], cfg, limits)


def slim(batch):
    keep = ("points", "neighbors", "pools", "upsamples", "features", "stack_lengths", "node_overlap_gt",
            "points2node", "correspondences")
    return {k: batch[k] for k in keep}


def main():
    F = ref_import.setup()
    from pcrcg_amd import synthetic as S
    from models.architectures import KPFCNN
    from models.blocks import KPConv
    from models.gcn import GCN
    os.makedirs(OUT, exist_ok=True)

    cfg = mini_cfg()
    src, tgt = S.pair("mini", 0)
    batch = ref_collate(src, tgt, cfg, MINI_LIMITS)
    torch.save({"limits": MINI_LIMITS, "batch": slim(batch)}, os.path.join(OUT, "collate_mini.pt"))

    torch.manual_seed(0)
    np.random.seed(0)
    model = KPFCNN(cfg).eval()
    inter = {}
    hooks = []
    for i in (0, 1, 2, 10):
        hooks.append(model.encoder_blocks[i].register_forward_hook(
            lambda m, a, o, i=i: inter.__setitem__(f"enc{i}", o.detach().clone())))
    hooks.append(model.bottle.register_forward_hook(
        lambda m, a, o: inter.__setitem__("bottle", o.detach()[0].t().clone())))
    hooks.append(model.gnn.register_forward_hook(
        lambda m, a, o: inter.__setitem__("gnn", torch.cat([o[0][0].t(), o[1][0].t()], 0).detach().clone())))
    with torch.no_grad():
        out = model(batch)
    for h in hooks:
        h.remove()
    cfg_plain = {k: v for k, v in cfg.items() if isinstance(v, (int, float, str, bool, list))}
    torch.save({"config": cfg_plain, "state_dict": {k: v.clone() for k, v in model.state_dict().items()},
                "outputs": {k: v.clone() for k, v in out.items()}, "intermediates": inter},
               os.path.join(OUT, "model_mini.pt"))
    print("model_mini params", sum(p.numel() for p in model.parameters()))

    # KPConv.forward cases on the mini tables
    cases = {}
    g = torch.Generator().manual_seed(1)
    pts, nb, pools = batch["points"], batch["neighbors"], batch["pools"]
    for name, (cin, cout, layer, strided) in {"c1_16": (1, 16, 0, False), "c8_8": (8, 8, 0, False),
                                              "c8_8_strided": (8, 8, 0, True), "c64_64": (64, 64, 2, False),
                                              "c64_64_strided": (64, 64, 2, True)}.items():
        r = 0.0625 * 2 ** layer
        np.random.seed(3)
        conv = KPConv(15, 3, cin, cout, r * 2.0 / 2.5, r)
        with torch.no_grad():
            conv.weights.copy_(torch.randn(conv.weights.shape, generator=g) * 0.2)
        s = pts[layer]
        q = pts[layer + 1] if strided else pts[layer]
        inds = pools[layer] if strided else nb[layer]
        x = torch.ones(s.shape[0], 1) if cin == 1 else torch.randn(s.shape[0], cin, generator=g)
        with torch.no_grad():
            y = conv(q, s, inds, x)
        cases[name] = dict(layer=layer, strided=strided, extent=float(conv.KP_extent), x=x,
                           kernel_points=conv.kernel_points.detach().clone(),
                           weights=conv.weights.detach().clone(), out=y)
    torch.save(cases, os.path.join(OUT, "kpconv_mini.pt"))

    # GCN.forward
    torch.manual_seed(2)
    gnn = GCN(4, 64, 10, ["self", "cross", "self"]).eval()
    c0, c1 = torch.rand(1, 3, 96, generator=g), torch.rand(1, 3, 80, generator=g)
    d0, d1 = torch.randn(1, 64, 96, generator=g), torch.randn(1, 64, 80, generator=g)
    with torch.no_grad():
        o0, o1 = gnn(c0, c1, d0, d1)
    torch.save(dict(state_dict={k: v.clone() for k, v in gnn.state_dict().items()}, c0=c0[0].t().clone(),
                    c1=c1[0].t().clone(), d0=d0[0].t().clone(), d1=d1[0].t().clone(), o0=o0[0].t().clone(),
                    o1=o1[0].t().clone()), os.path.join(OUT, "gcn_mini.pt"))
    # calibrate_neighbors (ref:datasets/dataloader.py:402-434) on one synthetic pair per recipe
    import json
    from datasets.dataloader import calibrate_neighbors, collate_fn_descriptor
    full = ref_import.indoor_config()
    lim = {}
    for rec in ("mini", "C1", "S30k"):
        s_, t_ = S.pair(rec, 0)
        corr = torch.stack([torch.arange(0, 50), torch.arange(0, 50)], 1)
        item = dict(rot=np.eye(3, dtype=np.float32), trans=np.zeros((3, 1), np.float32), correspondences=corr,
                    sample=0, src_pcd=s_, tgt_pcd=t_, src_feats=np.ones((len(s_), 1), np.float32),
                    tgt_feats=np.ones((len(t_), 1), np.float32))

        class DS(list):
            config = full
        lim[rec] = [int(v) for v in calibrate_neighbors(DS([item]), full, collate_fn_descriptor,
                                                        samples_threshold=10 ** 9)]
    json.dump({"_doc": "reference calibrate_neighbors (keep_ratio 0.8) on one synthetic pair per recipe, seed 0",
               "limits": lim}, open(os.path.join(OUT, "calibration.json"), "w"), indent=1)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
