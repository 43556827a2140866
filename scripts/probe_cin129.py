"""Probe: 129-channel input (PCR-CG image-feature width) through the runner vs the oracle, and the cost of the first KPConv (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import model_ref as MR
from pcrcg_amd import indoor_config, synthetic, ops
from pcrcg_amd.architectures import KPFCNN
from pcrcg_amd.pyramid import build_pyramid
dev = torch.device("cuda:0")
cfg = indoor_config(first_feats_dim=32, gnn_feats_dim=64, in_feats_dim=129)
torch.manual_seed(0); np.random.seed(0)
net = KPFCNN(cfg).eval()
sd = {k: v.clone() for k, v in net.state_dict().items()}
net = net.to(dev)
a, b = synthetic.pair("mini", 0)
pts = torch.from_numpy(np.concatenate([a, b])).to(dev); lens = torch.tensor([len(a), len(b)], dtype=torch.int32, device=dev)
batch = build_pyramid(pts, lens, cfg, [20, 26, 30, 32])
g = torch.Generator().manual_seed(1)
feats = torch.rand(pts.shape[0], 129, generator=g)
batch["features"] = feats.to(dev)
with torch.no_grad():
    out = net(batch)
cb = {k: ([t.cpu() if isinstance(t, torch.Tensor) else t for t in v] if isinstance(v, list) else (v.cpu() if isinstance(v, torch.Tensor) else v)) for k, v in batch.items()}
ref = MR.kpfcnn_forward(sd, dict(cfg), cb)
for k in ref:
    print(k, MR.rel_err(out[k].cpu(), ref[k]))
# timing of the first-layer gather at S30k with Cin = 129
a, b = synthetic.pair("S30k", 0)
pts = torch.from_numpy(np.concatenate([a, b])).to(dev); lens = torch.tensor([len(a), len(b)], dtype=torch.int32, device=dev)
cfgf = indoor_config(in_feats_dim=129)
batch = build_pyramid(pts, lens, cfgf, synthetic.LIMITS["S30k"])
x = torch.rand(60000, 129, device=dev)
netf = KPFCNN(cfgf).to(dev).eval()
blk = netf.encoder_blocks[0].KPConv
for cin, xx in ((129, x), (132, torch.rand(60000, 132, device=dev)), (128, torch.rand(60000, 128, device=dev))):
    w = torch.rand(15, cin, 128, device=dev)
    for _ in range(3):
        ops.kpconv(batch["points"][0], batch["points"][0], batch["neighbors"][0], xx, blk.kernel_points.data, w, blk.KP_extent)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        ops.kpconv(batch["points"][0], batch["points"][0], batch["neighbors"][0], xx, blk.kernel_points.data, w, blk.KP_extent)
    torch.cuda.synchronize(); print("cin", cin, "kpconv L0 ms", (time.perf_counter() - t0) / 10 * 1e3)
