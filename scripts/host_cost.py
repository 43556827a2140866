"""Host-side cost of one pair: time to ENQUEUE the forward (C++ runner) and to build the pyramid."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pcrcg_amd import indoor_config, synthetic
from pcrcg_amd.architectures import KPFCNN
from pcrcg_amd.pyramid import build_pyramid
dev = torch.device("cuda:0")
cfg = indoor_config(); torch.manual_seed(0); np.random.seed(0)
net = KPFCNN(cfg).to(dev).eval()
src, tgt = synthetic.pair("S30k", 0)
pts = torch.from_numpy(np.concatenate([src, tgt])).to(dev); lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev)
lim = synthetic.LIMITS["S30k"]
batch = build_pyramid(pts, lens, cfg, lim)
with torch.no_grad():
    for _ in range(3): net(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): net(batch)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"forward: enqueue {1e3*(t1-t0)/10:.3f} ms/pair, incl. GPU drain {1e3*(t2-t0)/10:.3f} ms/pair")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): build_pyramid(pts, lens, cfg, lim)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"pyramid alone (host+GPU, 4 syncs): {1e3*(t1-t0)/10:.3f} ms/pair")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): build_pyramid(pts, lens, cfg, lim)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
