"""GPU: the GEMM launches of the pair engine by shape (M = rows of all pairs of a grouped launch), from the kernels' own
start / stop events (pcrcg_profile_kpconv flag 2): launches and microseconds per PAIR, fp32-equivalent TFLOP/s, and the share
of the family's time -- in the engine (durations stretched by the streams beside them) and for one grouped forward of four
pairs alone.  python scripts/gemm_in_engine.py [pairs=96]"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pcrcg_amd import indoor_config, ops, synthetic  # noqa: E402
from pcrcg_amd.architectures import KPFCNN  # noqa: E402
from pcrcg_amd.pairstream import PairStreams  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device("cuda:0")
cfg = indoor_config()
limits = synthetic.LIMITS["S30k"]
torch.manual_seed(0)
np.random.seed(0)
net = KPFCNN(cfg).eval().to(dev)
pool = []
for s in range(4):
    a, b = synthetic.pair("S30k", s)
    pool.append((torch.from_numpy(np.concatenate([a, b])).to(dev), torch.tensor([len(a), len(b)], dtype=torch.int32, device=dev)))


def table(ev, pairs, title):
    by = {}
    for ms, m, n, k, prod, kind in ev:
        if kind != 3:
            continue
        d = by.setdefault((m, n, k), [0, 0.0])
        d[0] += 1
        d[1] += ms
    tot = sum(v[1] for v in by.values())
    print(f"\n{title}: {sum(v[0] for v in by.values()) / pairs:.1f} launches and {1e3 * tot / pairs:.0f} us of GEMM kernel time per pair")
    print(f"{'M (all pairs)':>14s} {'N':>6s} {'K':>6s} {'n/pair':>7s} {'avg us':>8s} {'us/pair':>8s} {'TF/s':>7s} {'share':>6s}")
    classes = {}
    for (m, n, k), (c, ms) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        tf = 2.0 * m * n * k * c / (ms * 1e-3) / 1e12
        print(f"{m:14d} {n:6d} {k:6d} {c / pairs:7.2f} {1e3 * ms / c:8.1f} {1e3 * ms / pairs:8.1f} {tf:7.1f} {100 * ms / tot:5.1f}%")
        cls = "coarse (level 3, GNN, heads: M <= 4 x 763)" if m <= 3100 else ("level 2 (M <= 4 x 3934)" if m <= 16000 else (
            "level 1 (M <= 4 x 15456)" if m <= 62000 else "level 0 (M = 4 x 60000)"))
        e = classes.setdefault(cls, [0, 0.0])
        e[0] += c
        e[1] += ms
    for cls, (c, ms) in classes.items():
        print(f"  {cls:46s} {c / pairs:6.1f} launches/pair {1e3 * ms / pairs:7.0f} us/pair {100 * ms / tot:5.1f}%")


eng = PairStreams(net, cfg, limits, dev)
for i in range(24):
    eng.submit(*pool[i % 4])
for _ in range(24):
    eng.result(wait=False)
eng.drain()
ops.kpconv_profile_start(gemm=True)
sub = 0
for i in range(n_pairs):
    while sub < min(n_pairs, i + 24):
        eng.submit(*pool[sub % 4])
        sub += 1
    eng.result(wait=False)
eng.drain()
torch.cuda.synchronize()
ev = ops.kpconv_profile_stop()
table(ev, n_pairs, "in the engine (3 model streams, groups of four pairs)")
# one grouped forward of four pairs alone
from pcrcg_amd.pyramid import NativePyramid  # noqa: E402
nat = NativePyramid(cfg, limits)
b, arena, lens_h, slot = nat.build([p for p, _ in pool], [l for _, l in pool], group=2)
runner = net.runner()
with torch.no_grad():
    for _ in range(2):
        runner.launch_group(b, 4, dev)
    torch.cuda.synchronize()
    ops.kpconv_profile_start(gemm=True)
    for _ in range(4):
        runner.launch_group(b, 4, dev)
    torch.cuda.synchronize()
ev = ops.kpconv_profile_stop()
table(ev, 16, "one grouped forward of four pairs, alone on the GPU")
eng.close()
