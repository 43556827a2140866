"""Precision experiment (CPU, build container): how far do the model outputs move when every dense
contraction of the oracle is computed from bf16 SPLITS of its fp32 operands (the arithmetic a
v_mfma_f32_32x32x16_bf16 based GEMM would do: exact bf16 x bf16 products, fp32 accumulate)?

  x3:  a1*b1 + a1*b2 + a2*b1                       (a = a1 + a2 + r, two bf16 terms per operand)
  x6:  a1*b1 + a1*b2 + a2*b1 + a1*b3 + a2*b2 + a3*b1   (three bf16 terms per operand, 24 bits)

Compared against a float64 run of the same oracle; the plain fp32 oracle is the yardstick.
Usage: python scripts/exp_splitbf16.py [recipe] [first_feats_dim gnn_feats_dim]
"""
import sys

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

sys.path.insert(0, ".")
from oracle import frontend as OF  # noqa: E402
from oracle import model_ref as MR  # noqa: E402
from pcrcg_amd import indoor_config, synthetic  # noqa: E402
from pcrcg_amd.architectures import KPFCNN  # noqa: E402


def split(x, n):
    parts = []
    r = x
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        parts.append(p)
        r = r - p
    return parts


class SplitMM(TorchFunctionMode):
    def __init__(self, terms):
        super().__init__()
        self.terms = terms

    def mm(self, a, b):
        if a.dtype != torch.float32:
            return None
        n = 2 if self.terms == 3 else 3
        A, B = split(a, n), split(b, n)
        pairs = [(0, 0), (0, 1), (1, 0)] if self.terms == 3 else [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]
        out = None
        with torch._C.DisableTorchFunction():
            for i, j in reversed(pairs):   # small terms first
                t = torch.matmul(A[i], B[j])
                out = t if out is None else out + t
        return out

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__):
            r = self.mm(args[0], args[1])
            if r is not None:
                return r
        return func(*args, **kwargs)


def main():
    recipe = sys.argv[1] if len(sys.argv) > 1 else "C1"
    over = {}
    if len(sys.argv) > 3:
        over = dict(first_feats_dim=int(sys.argv[2]), gnn_feats_dim=int(sys.argv[3]))
    cfg = indoor_config(**over)
    torch.manual_seed(0)
    np.random.seed(0)
    net = KPFCNN(cfg).eval()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    src, tgt = synthetic.pair(recipe, 0)
    limits = synthetic.LIMITS.get(recipe, [20, 26, 30, 32])
    batch = OF.oracle_pyramid(np.concatenate([src, tgt]), [len(src), len(tgt)], cfg, limits)
    sd64 = {k: v.double() for k, v in sd.items()}
    b64 = {k: ([t.double() if t.is_floating_point() else t for t in v] if isinstance(v, list) else
               (v.double() if torch.is_tensor(v) and v.is_floating_point() else v)) for k, v in batch.items()}
    ref = MR.kpfcnn_forward(sd64, dict(cfg), b64, return_intermediates=True)
    runs = {"fp32": None, "bf16x3": 3, "bf16x6": 6}
    for name, terms in runs.items():
        if terms is None:
            out = MR.kpfcnn_forward(sd, dict(cfg), batch, return_intermediates=True)
        else:
            with SplitMM(terms):
                out = MR.kpfcnn_forward(sd, dict(cfg), batch, return_intermediates=True)
        errs = {k: MR.rel_err(out[k].double(), ref[k]) for k in ("feats_f", "scores_overlap", "scores_saliency")}
        ie = {k: MR.rel_err(out["_inter"][k].double(), ref["_inter"][k]) for k in ("enc1", "enc5", "enc10", "gnn", "coarse")}
        print(name, {k: f"{v:.2e}" for k, v in errs.items()}, {k: f"{v:.2e}" for k, v in ie.items()}, flush=True)


if __name__ == "__main__":
    main()
