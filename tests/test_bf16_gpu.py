"""GPU: the bf16 feature-storage VARIANT (include/pcrcg.h pcrcg_model.feature_bf16; BASELINE.json configs[1] lists
"bf16/fp32").  It is NOT the parity path: features are rounded to bf16 where they are stored for the KPConv gathers
and where the aggregated [nq, 15*cin] matrix goes through HBM, everything else (weights, accumulation, norms, the GNN,
the outputs) is fp32.  These tests pin what the variant computes and state its error against the fp32 path and
against the unmodified reference's outputs:

  * the aggregate kernel: its bf16 copy of x is round-to-nearest-even; its output equals the fp32 kernel's output
    on that rounded x, rounded to bf16 (same arithmetic, only the storage differs);
  * the bf16-A GEMM: against float64 on the same (already rounded) A, fp32-class error;
  * the whole S30k forward at full width: error bound BF16_TOL (relative to the largest magnitude of each output),
    two orders above the fp32 path's 1e-4 and stated as such in DESIGN.md.
"""
import os

import numpy as np
import pytest
import torch

from pcrcg_amd import indoor_config, ops, synthetic
from pcrcg_amd.architectures import KPFCNN
from pcrcg_amd.pyramid import build_pyramid

pytestmark = pytest.mark.gpu
BF16_TOL = 1e-2          # whole-forward outputs vs the fp32 path and vs the reference (measured 5.6e-3 / 2.5e-3 / 2.5e-3)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def _bf16_view(t_i16):
    return t_i16.view(torch.bfloat16)


def _layer(cuda, nq=3000, ns=4000, h=37, cin=64, cout=128, seed=0):
    g = torch.Generator().manual_seed(seed)
    s = torch.rand(ns, 3, generator=g)
    q = s[torch.randperm(ns, generator=g)[:nq]] + 0.01 * torch.randn(nq, 3, generator=g)
    idx = torch.randint(0, ns + 1, (nq, h), generator=g)            # ns = the shadow neighbour
    x = torch.randn(ns, cin, generator=g).abs()
    x[::7] = 0                                                      # rows the neighbour-count normaliser skips
    kp = 0.06 * torch.randn(15, 3, generator=g)
    w = torch.randn(15, cin, cout, generator=g) / np.sqrt(15 * cin)
    return [t.to(cuda) for t in (q, s, idx, x, kp, w)]


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 128), (256, 256)])
def test_aggregate_bf16_is_the_fp32_kernel_on_rounded_storage(cuda, cin, cout):
    q, s, idx, x, kp, w = _layer(cuda, cin=cin, cout=cout)
    extent = 0.05
    out, xb, wfb, inv_n = ops.kpconv_bf16(q, s, idx, x, kp, w, extent, intermediates=True)
    # 1. the stored copy of x is round-to-nearest-even bf16
    assert torch.equal(_bf16_view(xb), x.to(torch.bfloat16))
    # 2. the aggregate: the fp32 kernel on the rounded x, then rounded -- bit for bit
    from pcrcg_amd import _lib
    L = _lib.lib()
    xr = x.to(torch.bfloat16).float()
    nq, ns, h = q.shape[0], s.shape[0], idx.shape[1]
    wf = torch.empty((nq, 15 * cin), dtype=torch.float32, device=cuda)
    inv32 = torch.empty(nq, dtype=torch.float32, device=cuda)
    nbytes = L.pcrcg_kpconv_ws_bytes(ns)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=cuda)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.pcrcg_kpconv_aggregate(q.data_ptr(), nq, s.data_ptr(), ns, idx.data_ptr(), h, h, xr.data_ptr(), cin,
                                        kp.data_ptr(), extent, wf.data_ptr(), inv32.data_ptr(), ws.data_ptr(), nbytes, st),
               "pcrcg_kpconv_aggregate")
    assert torch.equal(_bf16_view(wfb), wf.to(torch.bfloat16))
    # the normaliser counts neighbours by the fp32 features (as the fp32 path does)
    _lib.check(L.pcrcg_kpconv_aggregate(q.data_ptr(), nq, s.data_ptr(), ns, idx.data_ptr(), h, h, x.data_ptr(), cin,
                                        kp.data_ptr(), extent, wf.data_ptr(), inv32.data_ptr(), ws.data_ptr(), nbytes, st),
               "pcrcg_kpconv_aggregate")
    assert torch.equal(inv_n, inv32)
    # 3. the contraction: float64 on the stored operands
    want = (_bf16_view(wfb).double() @ w.reshape(-1, cout).double()) * inv_n.double()[:, None]
    assert rel(out, want) < 2e-6
    # and the distance of the whole layer from the fp32 layer (storage rounding only): bf16-class, not fp32-class
    full = ops.kpconv(q, s, idx, x, kp, w, extent)
    err = rel(out, full)
    assert 1e-5 < err < 8e-3, err


@pytest.mark.parametrize("m,n,k", [(1, 32, 32), (777, 64, 480), (5000, 256, 3840), (60001, 128, 960), (130, 1024, 64)])
def test_gemm_bf16a_against_float64(cuda, m, n, k):
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(cuda)
    b = torch.randn(n, k, generator=g).to(cuda)
    rs = torch.rand(m, generator=g).to(cuda) + 0.5
    bias = torch.randn(n, generator=g).to(cuda)
    got = ops.gemm_bf16a(a, b, row_scale=rs, bias=bias)
    want = (a.double() @ b.double().t()) * rs.double()[:, None] + bias.double()
    assert rel(got, want) < 2e-6
    # refusals: K not a multiple of 32
    with pytest.raises(RuntimeError):
        ops.gemm_bf16a(a[:, : k - 8].contiguous(), b[:, : k - 8].contiguous())


def test_s30k_forward_bf16_storage_error(cuda, golden_dir):
    gold = torch.load(os.path.join(golden_dir, "model_s30k.pt"))
    torch.manual_seed(0)
    np.random.seed(0)
    net = KPFCNN(indoor_config()).to(cuda).eval()
    src, tgt = synthetic.pair("S30k", gold["seed"])
    pts = torch.from_numpy(np.concatenate([src, tgt])).to(cuda)
    lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=cuda)
    batch = build_pyramid(pts, lens, indoor_config(), gold["limits"])
    with torch.no_grad():
        ref32 = {k: v.clone() for k, v in net(batch).items()}
        net.feature_bf16 = True
        got = {k: v.clone() for k, v in net(batch).items()}
        net.feature_bf16 = False
        again = net(batch)
    torch.cuda.synchronize()
    s = gold["stride"]
    report = {}
    for k in ref32:
        assert torch.isfinite(got[k]).all()
        report[k] = (rel(got[k], ref32[k]), rel(got[k][::s], gold["rows"][k]))
        assert report[k][0] < BF16_TOL and report[k][1] < BF16_TOL, (k, report[k])
        assert report[k][0] > 1e-6, "the variant did not run (outputs equal the fp32 path's)"
        assert rel(again[k], ref32[k]) < 1e-5          # switching back restores the fp32 path
    print("bf16 storage vs (fp32 path, reference):", report)
