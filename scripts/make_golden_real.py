"""Real-data parity fixtures from the reference's own demo assets (ref:assets/cloud_bin_21.pth,
cloud_bin_34.pth: two real 3DMatch fragments, 25 337 and 14 602 points; ref:configs/train/indoor.yaml:83-86).
Runs only in the build container.

  tests/golden/real_pair.npz        the two clouds as float32 (DATA of the reference, the inputs of its demo)
  tests/golden/frontend_digests.json["REAL"]   pyramid of the UNMODIFIED reference C++ front end on that pair
                                    (same digest scheme as the synthetic recipes, scripts/make_golden_frontend.py)
  tests/golden/model_real.pt        outputs of the UNMODIFIED reference KPFCNN (reduced-width weights of
                                    model_mini.pt) on the reference's own collate of that pair with the limits the
                                    reference's calibration gives for it ([41,38,36,35]): every 29th
                                    row of feats / overlap / saliency plus whole-tensor means -- twice:
                                      rows            the reference's tables as they are;
                                      rows_canonical  the same tables with every group of EXACTLY equal distance
                                                      re-ordered by index before the `[:, :limit]` cut.
Real scans are full of exactly equal distances (17 734 of 39 939 level-0 rows hold a tie).  Inside such a group the
reference's order is whatever nanoflann's traversal + std::sort (unstable) leave, so when the cut falls inside a
group the kept SET is arbitrary: 334 rows of this pair.  The HIP path defines the order (index ascending) and is
held to `rows_canonical`; `rows` documents how far that reference-internal arbitrariness moves the outputs."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from make_golden_frontend import sha  # noqa: E402

REPO = ref_import.REPO
OUT = os.path.join(REPO, "tests", "golden")
LIMITS = [41, 38, 36, 35]
STRIDE = 29


def main():
    F = ref_import.setup()
    F.build(ref=True)
    from tests.tieutil import canonicalise_table
    src = np.asarray(torch.load("/root/reference/assets/cloud_bin_21.pth", weights_only=False)).astype(np.float32)
    tgt = np.asarray(torch.load("/root/reference/assets/cloud_bin_34.pth", weights_only=False)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "real_pair.npz"), src=src, tgt=tgt)

    # front end digests with the reference C++ cores
    pts, lens = np.concatenate([src, tgt]), np.array([len(src), len(tgt)], np.int32)
    r, dl, d = 0.0625, 0.05, {}
    for l in range(4):
        d[f"points{l}"] = {"shape": list(pts.shape), "sha256": sha(pts)}
        d[f"lens{l}"] = {"shape": list(lens.shape), "sha256": sha(lens)}
        t = F.ref_batch_query(pts, pts, lens, lens, r)
        canon, ties = canonicalise_table(t, pts, pts)
        d[f"conv{l}"] = {"shape": list(t.shape), "sha256_canonical": sha(canon), "tie_rows": int(ties)}
        if l == 3:
            break
        sp, sl = F.ref_subsample_batch(pts, lens, dl)
        for name, q, s, ql, sl_, rr in ((f"pool{l}", sp, pts, sl, lens, r), (f"up{l}", pts, sp, lens, sl, 2 * r)):
            t = F.ref_batch_query(q, s, ql, sl_, rr)
            canon, ties = canonicalise_table(t, q, s)
            d[name] = {"shape": list(t.shape), "sha256_canonical": sha(canon), "tie_rows": int(ties)}
        pts, lens, r, dl = sp, sl, r * 2, dl * 2
    path = os.path.join(OUT, "frontend_digests.json")
    digests = json.load(open(path))
    digests["REAL"] = d
    json.dump(digests, open(path, "w"), indent=1, sort_keys=True)

    # reference model on the reference collate
    from datasets.dataloader import collate_fn_descriptor
    from models.architectures import KPFCNN
    gold = torch.load(os.path.join(OUT, "model_mini.pt"))
    cfg = ref_import.indoor_config(first_feats_dim=gold["config"]["first_feats_dim"],
                                   gnn_feats_dim=gold["config"]["gnn_feats_dim"])
    item = dict(rot=np.eye(3, dtype=np.float32), trans=np.zeros((3, 1), np.float32),
                correspondences=torch.stack([torch.arange(0, 50), torch.arange(0, 50)], 1), sample=0, src_pcd=src,
                tgt_pcd=tgt, src_feats=np.ones((len(src), 1), np.float32), tgt_feats=np.ones((len(tgt), 1), np.float32))
    batch = collate_fn_descriptor([item], cfg, LIMITS)
    model = KPFCNN(cfg).eval()
    model.load_state_dict(gold["state_dict"])
    with torch.no_grad():
        out = model(batch)
    # the same batch with canonical order inside tie groups (tables recomputed untruncated, re-ordered, then cut)
    canon = {k: (list(v) if isinstance(v, list) else v) for k, v in batch.items()}
    pts, lens = np.concatenate([src, tgt]), np.array([len(src), len(tgt)], np.int32)
    r, dl, straddle = 0.0625, 0.05, {}

    def cut(t, q, s_, l, name):
        c, _ = canonicalise_table(t, q, s_)
        straddle[name] = int((np.sort(c[:, :LIMITS[l]], 1) != np.sort(t[:, :LIMITS[l]], 1)).any(1).sum())
        return torch.from_numpy(c[:, :LIMITS[l]].astype(np.int64))

    for l in range(4):
        canon["neighbors"][l] = cut(F.ref_batch_query(pts, pts, lens, lens, r), pts, pts, l, f"conv{l}")
        if l == 3:
            break
        sp, sl = F.ref_subsample_batch(pts, lens, dl)
        canon["pools"][l] = cut(F.ref_batch_query(sp, pts, sl, lens, r), sp, pts, l, f"pool{l}")
        canon["upsamples"][l] = cut(F.ref_batch_query(pts, sp, lens, sl, 2 * r), pts, sp, l, f"up{l}")
        pts, lens, r, dl = sp, sl, r * 2, dl * 2
    with torch.no_grad():
        out_c = model(canon)
    torch.save({"limits": LIMITS, "stride": STRIDE, "levels": [int(p.shape[0]) for p in batch["points"]],
                "rows": {k: v[::STRIDE].clone() for k, v in out.items()},
                "rows_canonical": {k: v[::STRIDE].clone() for k, v in out_c.items()},
                "means": {k: float(v.double().mean()) for k, v in out.items()},
                "means_canonical": {k: float(v.double().mean()) for k, v in out_c.items()},
                "rows_with_different_kept_set": straddle},
               os.path.join(OUT, "model_real.pt"))
    print("rows whose kept neighbour set depends on the tie order:", straddle)
    print("levels", [int(p.shape[0]) for p in batch["points"]], {k: tuple(v.shape) for k, v in out.items()})
    print({k: v["tie_rows"] for k, v in d.items() if "tie_rows" in v})


if __name__ == "__main__":
    main()
