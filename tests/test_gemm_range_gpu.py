"""GPU: the fp32 contract of the default GEMM arithmetic at BOTH ends of fp16's range (csrc/gemm_x6.hip, H2 kernels).

The reference's contractions are plain fp32 (ref:models/blocks.py:354-366, ref:models/gcn.py:123-132,165-173): correct
at any finite operand scale, and the network behind every product is scale-free (InstanceNorm re-amplifies).  The fp16
two-term split has an absolute floor of 2^-36 per value below 2^-14; round 5 closes that side inside the kernel: a tile
row of A or B whose values are all fp16-subnormal (and not all zero) sends the tile to the exact three-term bf16 loop.
These tests hold WHOLE operands -- not single planted elements -- at scales 1e-3 ... 1e-12 to the fp32 bar."""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as MR
from pcrcg_amd import indoor_config, ops
from pcrcg_amd.architectures import KPFCNN

pytestmark = pytest.mark.gpu
BAR = 2.5e-6          # max-norm distance from a float64 product, the bar of test_gemm_fp16_two_term_form_and_its_range_fallback
SCALES = [1e-3, 1e-4, 3e-5, 1e-5, 1e-6, 1e-7, 1e-8, 1e-10, 1e-12]


def rel64(got, ref):
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("m,n,k", [(3000, 128, 256), (763, 512, 1920), (381, 130, 96), (20000, 64, 960)])
@pytest.mark.parametrize("which", ["A", "B", "both"])
def test_forward_product_whole_operand_scales(cuda, m, n, k, which):
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g)
    b = torch.randn(k, n, generator=g) * 0.05
    for s in SCALES:
        aa = a * s if which in ("A", "both") else a
        bb = b * s if which in ("B", "both") else b
        ref = aa.double() @ bb.double()
        got = ops.gemm(aa.to(cuda), bb.to(cuda))
        assert rel64(got, ref) < BAR, (which, s, rel64(got, ref))
        # nn.Linear form: B stored [n, k] (k-contiguous weights), the layout every weight product of the path uses
        got = ops.gemm(aa.to(cuda), bb.t().contiguous().to(cuda).t())
        assert rel64(got, ref) < BAR, (which, s, "k-contiguous B")


def test_rows_and_columns_of_different_scale_in_one_tile(cuda):
    """Row granularity: output rows (rows of A) and output columns (rows of B) of very different magnitude inside the same
    64 x 64 tile -- each row / column is held to the bar against ITS OWN magnitude (an InstanceNorm behind the product
    re-amplifies a small output channel; a per-tile or per-tensor criterion would not see it)."""
    g = torch.Generator().manual_seed(5)
    m, n, k = 1024, 256, 512
    a = torch.randn(m, k, generator=g)
    b = torch.randn(k, n, generator=g) * 0.05
    rs = torch.tensor([1.0, 1e-6, 1e-9, 3e-5])[torch.arange(m) % 4]
    cs = torch.tensor([1.0, 1e-7, 1e-3, 1e-10, 1.0])[torch.arange(n) % 5]
    a2, b2 = a * rs[:, None], b * cs[None, :]
    ref = a2.double() @ b2.double()
    got = ops.gemm(a2.to(cuda), b2.t().contiguous().to(cuda).t()).double().cpu()
    per_row = ((got - ref).abs() / (rs.double()[:, None] * cs.double()[None, :])).max() / (a.double() @ b.double()).abs().max()
    assert float(per_row) < BAR, float(per_row)
    # zero rows / columns beside small ones stay exact zeros and do not hide the small ones from the check
    a3, b3 = a2.clone(), b2.clone()
    a3[::3] = 0.0
    b3[:, ::2] = 0.0
    ref3 = a3.double() @ b3.double()
    got3 = ops.gemm(a3.to(cuda), b3.t().contiguous().to(cuda).t()).double().cpu()
    assert float(got3[::3].abs().max()) == 0.0 and float(got3[:, ::2].abs().max()) == 0.0
    err = ((got3 - ref3).abs() / (rs.double()[:, None] * cs.double()[None, :])).max() / (a.double() @ b.double()).abs().max()
    assert float(err) < BAR, float(err)


@pytest.mark.parametrize("scale", [1e-6, 1e-9, 1e-10, 1e-11, 1e-12, 1e-20])
def test_backward_forms_at_every_gradient_scale(cuda, scale):
    """include/pcrcg_train.h pcrcg_gemm_f32_grad: dX = dY W (B k-major), the k-contiguous dX = dY W^T and dW = X^T dY (A
    k-major) with gradients far below what the 2^16 lift brings back into fp16's normal range (|g| < 2^-30 = 9.3e-10):
    those tiles take the bf16 redo and stay fp32-class."""
    g = torch.Generator().manual_seed(13)
    m, n, k = 3000, 128, 256
    dy = (torch.randn(m, n, generator=g) * scale).to(cuda)
    w = (torch.randn(n, k, generator=g) / n ** 0.5).to(cuda)
    x = torch.randn(m, k, generator=g).to(cuda)
    ref_dx, ref_dw = dy.double().cpu() @ w.double().cpu(), x.double().cpu().t() @ dy.double().cpu()
    assert rel64(ops.gemm(dy, w, grad_operand=1), ref_dx) < BAR
    assert rel64(ops.gemm(dy, w.t().contiguous().t(), grad_operand=1), ref_dx) < BAR
    assert rel64(ops.gemm(x.t(), dy, grad_operand=2), ref_dw) < BAR
    # the same products from a caller that does not name the gradient operand
    assert rel64(ops.gemm(dy, w), ref_dx) < BAR
    assert rel64(ops.gemm(dy, w.t().contiguous().t()), ref_dx) < BAR
    assert rel64(ops.gemm(x.t(), dy), ref_dw) < BAR


@pytest.mark.parametrize("scale", [1e-6, 1e-9])
def test_model_with_small_input_features(cuda, golden_dir, scale):
    """A 129-wide input (the image-feature branch's width, ref:models/architectures.py:195-514) whose values sit at 1e-6 /
    1e-9 through the whole KPFCNN + GCN against the CPU oracle at the 1e-4 bar: the first KPConv's aggregated rows and its
    contraction see operands uniformly below fp16's normal range; the first InstanceNorm re-amplifies whatever they lose."""
    col = torch.load(os.path.join(golden_dir, "collate_mini.pt"))["batch"]
    cfg = indoor_config(first_feats_dim=32, gnn_feats_dim=64, in_feats_dim=129)
    torch.manual_seed(3)
    np.random.seed(3)
    net = KPFCNN(cfg).eval()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    n0 = col["points"][0].shape[0]
    feats = torch.randn(n0, 129, generator=torch.Generator().manual_seed(4)) * scale
    cpu_batch = dict(col)
    cpu_batch["features"] = feats
    ref = MR.kpfcnn_forward(sd, dict(cfg), cpu_batch)
    net = net.to(cuda)
    batch = {k: ([t.to(cuda) if isinstance(t, torch.Tensor) else t for t in v] if isinstance(v, list)
                 else (v.to(cuda) if isinstance(v, torch.Tensor) else v)) for k, v in col.items()}
    batch["features"] = feats.to(cuda)
    with torch.no_grad():
        out = net(batch)
        out_ops = net.forward_ops(dict(batch))
    for key in ("feats_f", "scores_overlap", "scores_saliency"):
        assert MR.rel_err(out[key].cpu(), ref[key]) < 1e-4, (key, MR.rel_err(out[key].cpu(), ref[key]))
        assert MR.rel_err(out_ops[key].cpu(), ref[key]) < 1e-4, (key, "forward_ops")


def test_redo_counters_report_which_range_end_was_left(cuda):
    """include/pcrcg.h pcrcg_gemm_redo_counts: a product on ordinary operands redoes no tile, one with a value beyond 65504
    counts under [0], one whose A rows sit at 1e-9 under [1] -- the diagnostic bench.py / scripts/redo_probe.py read."""
    import ctypes
    from pcrcg_amd import _lib
    L = _lib.lib()
    out = (ctypes.c_ulonglong * 2)()
    g = torch.Generator().manual_seed(2)
    a = torch.randn(512, 256, generator=g).to(cuda)
    w = (torch.randn(128, 256, generator=g) * 0.05).to(cuda)

    def counts(aa):
        _lib.check(L.pcrcg_gemm_redo_counts(None, 1), "pcrcg_gemm_redo_counts")
        ops.gemm(aa, w.t())
        torch.cuda.synchronize()
        _lib.check(L.pcrcg_gemm_redo_counts(out, 1), "pcrcg_gemm_redo_counts")
        return int(out[0]), int(out[1])

    assert counts(a) == (0, 0)
    big = a.clone()
    big[3, 5] = 1.0e6
    over, under = counts(big)
    assert over >= 1 and under == 0
    over, under = counts(a * 1e-9)
    assert over == 0 and under == (512 // 64) * (128 // 64)          # every tile of the product
