"""Diagnosis aid (GPU box): where does the K120k forward leave the reference fixture?  Runs the runner, the op-by-op
mirror, and the mirror with the GNN's kNN rows replaced by the reference's own rows (tests/golden/model_k120k.pt)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcrcg_amd import kitti_config, ops, synthetic  # noqa: E402
from pcrcg_amd.architectures import KPFCNN  # noqa: E402
from pcrcg_amd.pyramid import build_pyramid  # noqa: E402

dev = torch.device("cuda:0")
gold = torch.load("tests/golden/model_k120k.pt")
cfg = kitti_config()
src, tgt = synthetic.slab_pair(120000, gold["seed"])
pts = torch.from_numpy(np.concatenate([src, tgt])).to(dev)
lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev)
batch = build_pyramid(pts, lens, cfg, gold["limits"])
torch.manual_seed(0)
np.random.seed(0)
model = KPFCNN(cfg).to(dev).eval()
s = gold["stride"]


def report(tag, out):
    for k, want in gold["rows"].items():
        a, b = out[k][::s].double().cpu(), want.double()
        d = (a - b).abs()
        d = d.reshape(d.shape[0], -1).max(1)[0]
        print(tag, k, "max rel %.2e" % float(d.max() / b.abs().max()), "rows above 1e-4:",
              int((d > 1e-4 * b.abs().max()).sum()), "of", len(d), flush=True)


with torch.no_grad():
    report("runner", model(batch))
    report("mirror", model.forward_ops(batch))
    real = ops.knn
    ns = int(batch["stack_lengths"][-1][0])

    def patched(coords, k):
        want = gold["knn_src"] if coords.shape[0] == ns and patched.turn % 2 == 0 else gold["knn_tgt"]
        patched.turn += 1
        got = real(coords, k)
        print("  knn call", tuple(coords.shape), "rows differing from the reference:",
              int((got.cpu() != want).any(1).sum()))
        return want.to(coords.device).contiguous()
    patched.turn = 0
    ops.knn = patched
    import pcrcg_amd.gcn as G
    if hasattr(G, "ops"):
        G.ops.knn = patched
    report("mirror+ref-knn", model.forward_ops(batch))
