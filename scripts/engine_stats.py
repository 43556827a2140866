"""GPU: where the pair engine's host threads spend a long region (PairStreams.stats): the front thread idle (no input) /
waiting for an arena (its readers' forwards not yet passed) / inside pcrcg_pyramid_build; the model threads idle (no job) /
enqueueing.  python scripts/engine_stats.py [front_threads=1] [pairs=480]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pcrcg_amd import indoor_config, synthetic  # noqa: E402
from pcrcg_amd.architectures import KPFCNN  # noqa: E402
from pcrcg_amd.pairstream import PairStreams  # noqa: E402

front_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 480
dev = torch.device("cuda:0")
cfg = indoor_config()
limits = synthetic.LIMITS["S30k"]
torch.manual_seed(0)
np.random.seed(0)
net = KPFCNN(cfg).eval().to(dev)
pool = []
for s in range(8):
    a, b = synthetic.pair("S30k", s)
    pool.append((torch.from_numpy(np.concatenate([a, b])).to(dev), torch.tensor([len(a), len(b)], dtype=torch.int32, device=dev)))
for arenas in (4,):
    PairStreams.ARENAS = arenas
    eng = PairStreams(net, cfg, limits, dev, front_threads=front_threads)

    def run(n, depth=24):
        sub = 0
        for i in range(n):
            while sub < min(n, i + depth):
                eng.submit(*pool[sub % 8])
                sub += 1
            eng.result(wait=False)
        eng.drain()
        torch.cuda.synchronize()
    for depth in (24, 40):
        run(48, depth)
        eng.reset_stats()
        t0 = time.perf_counter()
        run(n_pairs, depth)
        dt = time.perf_counter() - t0
        st = eng.stats_snapshot()
        print(f"front threads {front_threads}, arenas {arenas} each, {depth} pairs ahead: {n_pairs / dt:6.1f} pairs/s over {1e3 * dt:.0f} ms; front thread: idle {1e3 * st['front_idle_s']:.0f} ms, waiting for an arena "
              f"{1e3 * st['arena_wait_s']:.0f} ms, in builds {1e3 * st['build_s']:.0f} ms ({st['builds']} builds of {st['pairs'] / max(st['builds'], 1):.2f} pairs); "
              f"model threads (3): idle {1e3 * st['model_idle_s']:.0f} ms, enqueueing {1e3 * st['launch_s']:.0f} ms", flush=True)
    eng.close()
