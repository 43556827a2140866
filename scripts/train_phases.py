"""GPU box: the train step by phase (differentiable forward, MetricLoss, backward, optimiser), each drained before the
next is timed, host time to enqueue vs time to completion.  PATH_PY=1: the op-by-op autograd
mirror instead of the C++ tape runner."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcrcg_amd import indoor_config, synthetic
from pcrcg_amd.architectures import KPFCNN
from pcrcg_amd.config import Config
from pcrcg_amd.correspondences import get_correspondences
from pcrcg_amd.loss import MetricLoss
from pcrcg_amd.pyramid import collate_fn_descriptor
from pcrcg_amd.trainer import Trainer, LOSS_KEYS
from pcrcg_amd.train_forward import forward_train
dev = torch.device("cuda:0")
cfg = indoor_config(); torch.manual_seed(0); np.random.seed(0)
net = KPFCNN(cfg).to(dev)
loss = MetricLoss(Config(pos_margin=0.1, neg_margin=1.4, pos_radius=0.0375, safe_radius=0.1, matchability_radius=0.05, max_points=256))
tr = Trainer(net, loss)
src, tgt, rot, trans = synthetic.lomatch_pair("S30k", 0, overlap=0.2)
tsfm = np.eye(4); tsfm[:3, :3], tsfm[:3, 3] = rot, trans.flatten()
corr = get_correspondences(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), tsfm, 0.0375)
item = dict(src_pcd=src, tgt_pcd=tgt, src_feats=np.ones((len(src), 1), np.float32), tgt_feats=np.ones((len(tgt), 1), np.float32), rot=rot, trans=trans, correspondences=corr, sample=0)
inputs = collate_fn_descriptor([item], cfg, synthetic.LIMITS["S30k"], device=dev)
for _ in range(3): tr.train_step(inputs)
torch.cuda.synchronize()
T = {"fwd": 0, "fwd_host": 0, "loss": 0, "loss_host": 0, "bwd": 0, "bwd_host": 0, "opt": 0, "stats": 0}
N = 10
for _ in range(N):
    net.train(True)
    t0 = time.perf_counter()
    out = net.train_runner().forward(inputs) if os.environ.get("PATH_PY") != "1" else forward_train(net, inputs)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    len_src = int(inputs["stack_lengths_host"][0][0])
    f = out["feats_f"]
    li = {"src_feats": f[:len_src], "tgt_feats": f[len_src:], "rot": inputs["rot"], "trans": inputs["trans"], "scores_overlap": out["scores_overlap"], "scores_saliency": out["scores_saliency"], "src_pcd_raw": inputs["src_pcd_raw"], "tgt_pcd_raw": inputs["tgt_pcd_raw"], "correspondences": inputs["correspondences"]}
    res = loss(li)
    c = sum(res[k] for k in res if k in LOSS_KEYS)
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    tr.bucket.arm(True)
    c.backward()
    t5 = time.perf_counter(); torch.cuda.synchronize(); t6 = time.perf_counter()
    tr.optimizer_step()
    torch.cuda.synchronize(); t7 = time.perf_counter()
    st = {k: float(v.detach()) if isinstance(v, torch.Tensor) else float(v) for k, v in res.items()}
    t8 = time.perf_counter()
    T["fwd_host"] += t1 - t0; T["fwd"] += t2 - t0; T["loss_host"] += t3 - t2; T["loss"] += t4 - t2
    T["bwd_host"] += t5 - t4; T["bwd"] += t6 - t4; T["opt"] += t7 - t6; T["stats"] += t8 - t7
print({k: round(1e3 * v / N, 2) for k, v in T.items()})
