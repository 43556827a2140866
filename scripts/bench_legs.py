import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
seq = sys.argv[1].split(",")
for leg in seq:
    if leg == "train":
        r = bench.secondary_train_step(dev, 12); print("train", r["ms_per_step"], flush=True)
    elif leg == "image":
        r = bench.secondary_image129(dev, 96, 8, 3, 4, 4); print("image", r["value"], r["forward_alone_ms"], flush=True)
    elif leg == "k120":
        r = bench.secondary_k120k(dev, 96, 6, 3, 3, 3); print("k120", r["value"], flush=True)
    elif leg == "empty":
        torch.cuda.empty_cache(); print("empty_cache", flush=True)
    elif leg == "mem":
        print("mem", torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, flush=True)
