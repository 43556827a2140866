import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["NO_EVENTS"] = "1"
import numpy as np, torch
import pcrcg_amd.loss as L
from pcrcg_amd import ops
T = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[label] = T.get(label, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
wrap(ops, "feature_argmax", "feature_argmax")
wrap(L.MetricLoss, "get_weighted_bce_loss", "bce")
orig_apply = L._CircleLoss.apply
def capply(*a):
    t0 = time.perf_counter(); r = orig_apply(*a); T["circle"] = T.get("circle", 0.0) + time.perf_counter() - t0; return r
L._CircleLoss.apply = staticmethod(capply)
orig_fwd = L.MetricLoss.forward
def fwd(self, inputs, prepared=None):
    t0 = time.perf_counter(); r = orig_fwd(self, inputs, prepared); T["loss_forward_total"] = T.get("loss_forward_total", 0.0) + time.perf_counter() - t0; return r
L.MetricLoss.forward = fwd
src = open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts/train_timeline.py")).read()
exec(src[:src.index("for _ in range(5): tr.train_step(inputs)")])
orig_rb = type(tr)._read_back
def rb(res, extra=None):
    t0 = time.perf_counter(); r = orig_rb(res, extra); T["read_back"] = T.get("read_back", 0.0) + time.perf_counter() - t0; return r
type(tr)._read_back = staticmethod(rb)
for _ in range(5): tr.train_step(inputs)
torch.cuda.synchronize(); T.clear()
t0 = time.perf_counter()
for _ in range(20): tr.train_step(inputs)
torch.cuda.synchronize()
print("Trainer.train_step: %.2f ms per step" % ((time.perf_counter() - t0) / 20 * 1e3))
print({k: round(1e3 * v / 20, 3) for k, v in T.items()}, "(host ms per step inside)")
