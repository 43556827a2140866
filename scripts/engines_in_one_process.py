import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from pcrcg_amd import indoor_config, synthetic
from pcrcg_amd.architectures import KPFCNN
from pcrcg_amd.pairstream import PairStreams
dev = torch.device("cuda:0")
cfg = indoor_config(); limits = synthetic.LIMITS["S30k"]
torch.manual_seed(0); np.random.seed(0)
net = KPFCNN(cfg).eval().to(dev)
pool = []
for s in range(4):
    a, b = synthetic.pair("S30k", s)
    pool.append((torch.from_numpy(np.concatenate([a, b])).to(dev), torch.tensor([len(a), len(b)], dtype=torch.int32, device=dev)))
def run(eng, n):
    sub = 0
    for i in range(n):
        while sub < min(n, i + 24):
            eng.submit(*pool[sub % 4]); sub += 1
        eng.result(wait=False)
    eng.drain(); torch.cuda.synchronize()
for k in range(4):
    eng = PairStreams(net, cfg, limits, dev)
    run(eng, 48)
    t0 = time.perf_counter(); run(eng, 240); dt = time.perf_counter() - t0
    print("engine", k, round(240 / dt, 1), eng.pipe_classes["candidates"], eng.pipe_classes["front"], eng.pipe_classes["model"], flush=True)
    eng.close()
    if len(sys.argv) > 1 and sys.argv[1] == "del":
        del eng
        import gc; gc.collect(); torch.cuda.empty_cache()
