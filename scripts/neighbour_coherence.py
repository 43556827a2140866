import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from pcrcg_amd import indoor_config, synthetic
from pcrcg_amd.pyramid import build_pyramid
dev = torch.device("cuda:0")
cfg = indoor_config()
src, tgt = synthetic.pair("S30k", 0)
pts = torch.from_numpy(np.concatenate([src, tgt])).to(dev)
lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev)
b = build_pyramid(pts, lens, cfg, synthetic.LIMITS["S30k"])
for name in ("neighbors", "pools"):
    for l, nb in enumerate(b[name]):
        nb = nb.cpu().numpy()
        if nb.size == 0: continue
        ns = int(b["points"][l].shape[0])
        n = nb.shape[0]
        out = []
        for B in (4, 16, 64, 256):
            tot = dis = 0
            for s in range(0, min(n, 16384), B):
                blk = nb[s:s + B]; v = blk[blk < ns]
                tot += v.size; dis += np.unique(v).size
            out.append(f"B={B}: {tot / max(dis, 1):.2f}")
        print(name, "level", l, "queries", n, "h", nb.shape[1], "valid/query", round(float((nb < ns).sum()) / n, 1), " pairs per distinct support:", ", ".join(out))
