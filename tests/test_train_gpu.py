"""GPU: the training-side rows through the C ABI of include/pcrcg_train.h."""
import os

import numpy as np
import pytest
import torch

from pcrcg_amd import ops
from pcrcg_amd.config import Config
from pcrcg_amd.loss import MetricLoss

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


@pytest.mark.parametrize("n,m,c", [(300, 257, 32), (1000, 3000, 32), (65, 129, 64), (40, 50, 20), (1, 1, 32), (6400, 28106, 32), (28106, 6318, 32), (33, 31, 32)])
def test_feature_argmax_matches_matmul_max(cuda, n, m, c):
    g = torch.Generator().manual_seed(n + m)
    a = torch.randn(n, c, generator=g).to(cuda)
    b = torch.randn(m, c, generator=g).to(cuda)
    arg, best = ops.feature_argmax(a, b, want_best=True)
    ref = a.double() @ b.double().t()
    ref_best, ref_arg = ref.max(1)
    assert rel(best, ref_best) < 1e-5
    assert torch.equal(arg, ref_arg)
    # strided operands (column slices of wider matrices)
    wide_a = torch.randn(n, c + 8, generator=g).to(cuda)
    wide_b = torch.randn(m, c + 4, generator=g).to(cuda)
    arg2 = ops.feature_argmax(wide_a[:, :c], wide_b[:, :c])
    assert torch.equal(arg2, (wide_a[:, :c].double() @ wide_b[:, :c].double().t()).argmax(1))


def test_feature_argmax_first_index_wins_ties(cuda):
    a = torch.ones(5, 32, device=cuda)
    b = torch.ones(7, 32, device=cuda)
    assert torch.equal(ops.feature_argmax(a, b), torch.zeros(5, dtype=torch.int64, device=cuda))
    with pytest.raises(RuntimeError):
        ops.feature_argmax(a, b[:0])


@pytest.mark.parametrize("case", ["capped", "all"])
def test_metric_loss_forward_matches_reference(cuda, golden_dir, case):
    """Full MetricLoss.forward on the device against the reference's forward (tests/golden/loss_mini.pt)."""
    gold = torch.load(os.path.join(golden_dir, "loss_mini.pt"), weights_only=False)
    cs = gold["cases"][case]
    loss = MetricLoss(Config(gold["config"]))
    inputs = {k: v.to(cuda) for k, v in cs["inputs"].items()}
    np.random.seed(cs["numpy_seed"])           # the max_points cap draws from the host generator (ref:lib/loss.py:231)
    stats = loss(inputs)
    assert set(stats) == set(cs["expected"])
    for k, want in cs["expected"].items():
        got = float(stats[k])
        assert abs(got - float(want)) <= 1e-4 * max(1.0, abs(float(want))), (k, got, float(want))


@pytest.mark.parametrize("case", ["capped", "all"])
def test_fused_loss_kernels_equal_the_torch_mirror(cuda, golden_dir, case):
    """csrc/lossops.hip (circle loss + recall, weighted BCE, with gradients) against the torch formulation of the same
    MetricLoss (fused=False), values and gradients wrt descriptors and scores."""
    gold = torch.load(os.path.join(golden_dir, "loss_mini.pt"), weights_only=False)
    cs = gold["cases"][case]
    res = {}
    for fused in (True, False):
        loss = MetricLoss(Config(gold["config"]), fused=fused)
        inputs = {k: v.to(cuda) for k, v in cs["inputs"].items()}
        for k in ("src_feats", "tgt_feats", "scores_overlap", "scores_saliency"):
            inputs[k] = inputs[k].clone().requires_grad_(True)
        np.random.seed(cs["numpy_seed"])
        stats = loss(inputs)
        (stats["circle_loss"] + stats["overlap_loss"] + stats["saliency_loss"]).backward()
        res[fused] = ({k: float(v.detach()) for k, v in stats.items()},
                      {k: inputs[k].grad.clone() for k in ("src_feats", "tgt_feats", "scores_overlap", "scores_saliency")})
    for k, want in res[False][0].items():
        assert abs(res[True][0][k] - want) <= 1e-5 * max(1.0, abs(want)), (k, res[True][0][k], want)
    for k, want in res[False][1].items():
        got = res[True][1][k]
        assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max()) + 1e-9, k


def test_fused_loss_propagates_nan_and_marks_statistics_non_differentiable(cuda, golden_dir):
    """A NaN descriptor makes the reference's circle loss NaN (torch.clamp propagates NaN, ref:lib/loss.py:20-34 via
    square_distance): the fused kernel must report NaN too, not the clamp's lower bound; and the statistics that come out
    of the fused autograd functions (recall, precision) carry no grad_fn."""
    from pcrcg_amd.loss import _CircleLoss, _WeightedBCE
    gold = torch.load(os.path.join(golden_dir, "loss_mini.pt"), weights_only=False)
    cs = next(iter(gold["cases"].values()))
    out = {}
    for fused in (True, False):
        loss = MetricLoss(Config(gold["config"]), fused=fused)
        inputs = {k: v.to(cuda) for k, v in cs["inputs"].items()}
        inputs["src_feats"] = inputs["src_feats"].clone()
        inputs["src_feats"][:] = float("nan")                      # every source descriptor: whatever the max_points draw keeps
        np.random.seed(cs["numpy_seed"])
        out[fused] = {k: float(v) for k, v in loss(inputs).items()}
    assert np.isnan(out[False]["circle_loss"]) and np.isnan(out[True]["circle_loss"]), out
    # torch.min treats NaN as the minimum (first NaN of the row): the recall statistic reads the same column in both forms --
    # and the kernel's arg-min is a valid column even when no comparison ever succeeds (it once stayed at INT_MAX: a read
    # 8 GB past the matrix, an intermittent memory fault)
    assert abs(out[True]["recall"] - out[False]["recall"]) <= 1e-6, out
    g = torch.Generator().manual_seed(0)
    a = torch.nn.functional.normalize(torch.randn(40, 32, generator=g), dim=1).to(cuda).requires_grad_(True)
    b = torch.nn.functional.normalize(torch.randn(40, 32, generator=g), dim=1).to(cuda).requires_grad_(True)
    cd = torch.rand(40, 40, generator=g).to(cuda) * 0.2
    lv, rec = _CircleLoss.apply(a, b, cd, (0.0375, 0.1, 0.1, 1.4, 0.1, 1.4, 16.0))
    assert lv.requires_grad and not rec.requires_grad and rec.grad_fn is None
    p = torch.rand(100, generator=g).to(cuda).requires_grad_(True)
    lv, prec, rec = _WeightedBCE.apply(p, (torch.rand(100, generator=g) > 0.5).float().to(cuda))
    assert lv.requires_grad and not prec.requires_grad and not rec.requires_grad


def test_prepared_loss_equals_the_loss(cuda, golden_dir):
    """MetricLoss.prepare() (the geometry-only part, which a trainer runs beside the network's forward) handed to forward()
    gives the same statistics as forward() alone, and consumes the host generator identically."""
    gold = torch.load(os.path.join(golden_dir, "loss_mini.pt"), weights_only=False)
    for case, cs in gold["cases"].items():
        loss = MetricLoss(Config(gold["config"]))
        inputs = {k: v.to(cuda) for k, v in cs["inputs"].items()}
        np.random.seed(cs["numpy_seed"])
        plain = {k: float(v) for k, v in loss(inputs).items()}
        after_plain = np.random.rand()
        np.random.seed(cs["numpy_seed"])
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            prepared = loss.prepare(inputs)
        torch.cuda.current_stream().wait_stream(side)
        ahead = {k: float(v) for k, v in loss(inputs, prepared=prepared).items()}
        assert np.random.rand() == after_plain, case
        # (the BCE's sums are accumulated with atomics: the last bits may differ from run to run)
        for k, v in plain.items():
            assert abs(ahead[k] - v) <= 1e-6 * max(1.0, abs(v)), (case, k, ahead[k], v)


def test_metric_loss_is_differentiable(cuda, golden_dir):
    gold = torch.load(os.path.join(golden_dir, "loss_mini.pt"), weights_only=False)
    cs = gold["cases"]["all"]
    loss = MetricLoss(Config(gold["config"]))
    inputs = {k: v.to(cuda) for k, v in cs["inputs"].items()}
    for k in ("src_feats", "tgt_feats", "scores_overlap", "scores_saliency"):
        inputs[k] = inputs[k].clone().requires_grad_(True)
    np.random.seed(cs["numpy_seed"])
    stats = loss(inputs)
    total = stats["circle_loss"] + stats["overlap_loss"] + stats["saliency_loss"]    # ref:lib/trainer.py:255-260
    total.backward()
    for k in ("src_feats", "tgt_feats", "scores_overlap", "scores_saliency"):
        g = inputs[k].grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0


def _tsfm(rot, trans):
    t = np.eye(4)
    t[:3, :3], t[:3, 3] = rot, trans.flatten()
    return t


@pytest.mark.parametrize("recipe,seed,K", [("mini", 3, None), ("mini", 5, 3), ("C1", 1, None)])
def test_get_correspondences_matches_oracle(cuda, recipe, seed, K):
    """Device get_correspondences == float64 brute force (oracle/correspondences.py), pair for pair."""
    from oracle.correspondences import get_correspondences as oracle_corr
    from pcrcg_amd import synthetic
    from pcrcg_amd.correspondences import get_correspondences
    src, tgt, rot, trans = synthetic.lomatch_pair(recipe, seed, overlap=0.3)
    got = get_correspondences(torch.from_numpy(src).to(cuda), torch.from_numpy(tgt).to(cuda), _tsfm(rot, trans),
                              0.0375, K=K)
    want = oracle_corr(src, tgt, _tsfm(rot, trans), 0.0375, K=K)
    assert got.dtype == torch.int64 and tuple(got.shape) == want.shape and len(want) > 1000
    assert np.array_equal(got.cpu().numpy(), want)


def test_get_correspondences_edge_cases(cuda):
    from pcrcg_amd.correspondences import get_correspondences
    pts = torch.rand(100, 3, device=cuda)
    eye = np.eye(4)
    far = eye.copy()
    far[:3, 3] = 50.0
    assert get_correspondences(pts, pts, far, 0.05).shape == (0, 2)           # no overlap at all
    assert get_correspondences(pts[:0], pts, eye, 0.05).shape == (0, 2)       # empty cloud
    same = get_correspondences(pts, pts, eye, 1e-6)                           # identity: every point finds itself
    assert torch.equal(same, torch.arange(100, device=cuda)[:, None].repeat(1, 2))
    dense = get_correspondences(pts, pts, eye, 10.0)                          # radius covers everything: N*N pairs
    assert dense.shape == (100 * 100, 2) and torch.equal(dense[::100, 1], torch.arange(100, device=cuda))
    with pytest.raises(RuntimeError):
        get_correspondences(pts.cpu(), pts.cpu(), eye, 0.05)
    many = torch.rand(700, 3, device=cuda)                                    # 700 hits per row (second, wider pass)
    dense = get_correspondences(many, many, eye, 10.0, K=650)
    assert dense.shape == (700 * 650, 2) and torch.equal(dense[::650, 1], torch.arange(700, device=cuda))
    with pytest.raises(RuntimeError):                                         # beyond the 1024 hits a row can stage
        get_correspondences(torch.rand(1100, 3, device=cuda), torch.rand(1100, 3, device=cuda), eye, 10.0)


def test_get_correspondences_at_s30k_size(cuda):
    """configs[2]'s pair: the rows of 400 sampled source points against the float64 brute force, and the whole call in a
    few milliseconds (round 2's torch formulation took 527 ms)."""
    import time
    from oracle.correspondences import get_correspondences as oracle_corr
    from pcrcg_amd import synthetic
    from pcrcg_amd.correspondences import get_correspondences
    src, tgt, rot, trans = synthetic.lomatch_pair("S30k", 1, 0.2)
    s, t = torch.from_numpy(src).to(cuda), torch.from_numpy(tgt).to(cuda)
    got = get_correspondences(s, t, _tsfm(rot, trans), 0.0375)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        got = get_correspondences(s, t, _tsfm(rot, trans), 0.0375)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print("get_correspondences, S30k-LoMatch pair: %.2f ms, %d pairs" % (ms, got.shape[0]))
    assert got.shape[0] > 10000 and ms < 20.0
    rows = np.random.RandomState(0).permutation(len(src))[:400]
    rows.sort()
    want = oracle_corr(src[rows], tgt, _tsfm(rot, trans), 0.0375)
    g = got.cpu().numpy()
    sel = g[np.isin(g[:, 0], rows)]
    assert np.array_equal(np.searchsorted(rows, sel[:, 0]), want[:, 0]) and np.array_equal(sel[:, 1], want[:, 1])


def test_evaluate_pair_record_and_recall(cuda):
    """Tester loop body: forward -> MetricLoss recall -> the dict the reference dumps per pair."""
    from pcrcg_amd import indoor_config, synthetic
    from pcrcg_amd.architectures import KPFCNN
    from pcrcg_amd.correspondences import get_correspondences
    from pcrcg_amd.pyramid import collate_fn_descriptor
    from pcrcg_amd.tester import evaluate_pair
    cfg = indoor_config(first_feats_dim=64, gnn_feats_dim=128)
    torch.manual_seed(0)
    net = KPFCNN(cfg).to(cuda).eval()
    src, tgt, rot, trans = synthetic.lomatch_pair("mini", 2, overlap=0.3)
    corr = get_correspondences(torch.from_numpy(src).to(cuda), torch.from_numpy(tgt).to(cuda), _tsfm(rot, trans), 0.0375)
    item = dict(src_pcd=src, tgt_pcd=tgt, src_feats=np.ones((len(src), 1), np.float32),
                tgt_feats=np.ones((len(tgt), 1), np.float32), rot=rot, trans=trans, correspondences=corr.cpu(), sample=0)
    inputs = collate_fn_descriptor([item], cfg, [20, 26, 30, 32], device=cuda)
    loss = MetricLoss(Config(pos_margin=0.1, neg_margin=1.4, pos_radius=0.0375, safe_radius=0.1,
                             matchability_radius=0.05, max_points=256))
    np.random.seed(0)
    rec, stats = evaluate_pair(net, loss, inputs)
    n = len(src) + len(tgt)
    assert rec["pcd"].shape == (n, 3) and rec["feats"].shape == (n, 32) and rec["len_src"] == len(src)
    assert not rec["feats"].is_cuda and rec["overlaps"].shape == (n,) and rec["saliency"].shape == (n,)
    assert torch.equal(rec["rot"], torch.from_numpy(rot)) and rec["trans"].shape == (3, 1)
    assert 0.0 <= float(stats["recall"]) <= 1.0 and torch.isfinite(stats["circle_loss"])


@pytest.mark.parametrize("m,n,k,tb", [(64, 64, 6000, 0), (960, 64, 15433, 0), (130, 70, 999, 1), (256, 1, 777, 0),
                                       (1, 33, 50, 1), (2048, 512, 379, 0)])
def test_gemm_transposed_a(cuda, m, n, k, tb):
    """dW-shaped products: A^T given as a transposed view of a row-major [k, m] matrix (no copy)."""
    g = torch.Generator().manual_seed(m * 7 + n)
    at = torch.randn(k, m, generator=g).to(cuda)
    b = torch.randn(n, k, generator=g).to(cuda).t() if tb else torch.randn(k, n, generator=g).to(cuda)
    got = ops.gemm(at.t(), b)
    ref = at.double().t() @ b.double()
    assert rel(got, ref) < 2e-5
    scale, bias = torch.rand(m, generator=g).to(cuda), torch.randn(n, generator=g).to(cuda)
    got = ops.gemm(at.t(), b, row_scale=scale, bias=bias)
    assert rel(got, ref * scale.double()[:, None] + bias.double()) < 2e-5
    # a column slice of a wider matrix as A^T (lda > m, unaligned start)
    wide = torch.randn(k, m + 5, generator=g).to(cuda)
    assert rel(ops.gemm(wide[:, 1:m + 1].t(), b), wide[:, 1:m + 1].double().t() @ b.double()) < 2e-5


@pytest.mark.parametrize("scale", [1.0, 1e-4, 1e-8])
def test_backward_products_keep_precision_for_tiny_gradients(cuda, scale):
    """include/pcrcg_train.h pcrcg_gemm_f32_grad (ops.gemm(..., grad_operand=)): the three backward forms dX = dY W,
    dW = X^T dY and the k-contiguous dX = dY W^T with gradient values around `scale` -- far below fp16's normal range at
    1e-8 -- stay within fp32-class distance of a float64 product (the gradient operand is lifted by 2^16 before the fp16
    split, csrc/gemm_x6.hip); a gradient beyond 1 (here 300) takes the bf16 redo and is exact as well."""
    g = torch.Generator().manual_seed(11)
    m, n, k = 3000, 128, 256
    dy = (torch.randn(m, n, generator=g) * scale).to(cuda)
    w = (torch.randn(n, k, generator=g) / n ** 0.5).to(cuda)         # [Cout, Cin]: dX = dY @ W
    x = torch.randn(m, k, generator=g).to(cuda)

    def rel64(got, ref):
        return float((got.double() - ref).abs().max() / ref.abs().max())

    dx = ops.gemm(dy, w, grad_operand=1)                              # A = gradient, B k-major
    assert rel64(dx, dy.double() @ w.double()) < 2e-6
    dx2 = ops.gemm(dy, w.t().contiguous().t(), grad_operand=1)        # the k-contiguous form (B^T stored)
    assert rel64(dx2, dy.double() @ w.double()) < 2e-6
    dw = ops.gemm(x.t(), dy, grad_operand=2)                          # A^T activations, B = gradient
    assert rel64(dw, x.double().t() @ dy.double()) < 2e-6
    big = dy.clone()
    big[7, 3] = 300.0
    dxb = ops.gemm(big, w.t().contiguous().t(), grad_operand=1)
    assert torch.isfinite(dxb).all() and rel64(dxb, big.double() @ w.double()) < 2e-6


def test_nonfinite_flag_over_a_flat_buffer(cuda):
    """pcrcg_nonfinite_flag (include/pcrcg_train.h; validate_gradient, ref:lib/utils.py:100-111): 0 for finite values of
    any magnitude (subnormals, the largest float), 1 for a NaN or an Inf anywhere -- including the last, unaligned elements."""
    import ctypes
    from pcrcg_amd import _lib
    L = _lib.lib()
    flag = torch.full((1,), 7.0, device=cuda)

    def check(x):
        _lib.check(L.pcrcg_nonfinite_flag(x.data_ptr(), x.numel(), flag.data_ptr(), ops._stream()), "pcrcg_nonfinite_flag")
        torch.cuda.synchronize()
        return float(flag[0])
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1_000_003, generator=g).to(cuda)
    x[5], x[77], x[-1] = 3.4028234e38, 1e-45, -3.4028234e38
    assert check(x) == 0.0
    for pos, bad in ((0, float("nan")), (123_457, float("inf")), (x.numel() - 1, float("-inf")), (x.numel() - 2, float("nan"))):
        y = x.clone()
        y[pos] = bad
        assert check(y) == 1.0, (pos, bad)
    assert check(x[:0]) == 0.0 and check(x[:3]) == 0.0
    from pcrcg_amd.trainer import GradientBucket
    p = torch.nn.Parameter(torch.zeros(1000, device=cuda))
    bucket = GradientBucket([p])
    assert bucket.finite() is True
    bucket.flat[17] = float("nan")
    assert bucket.finite() is False


def test_gather_jobs_maps_transposes_and_the_way_back(cuda):
    """pcrcg_gather_jobs (include/pcrcg_train.h): dst[i] = src[m1[i]] + s2 src[m2[i]] with -1 = nothing, plain transposes
    through LDS tiles (m1 = NULL), and the accumulate form that carries gradients back -- a table of jobs in one launch."""
    import ctypes
    from pcrcg_amd import _lib
    from pcrcg_amd.train_runner import GatherJob
    L = _lib.lib()
    g = torch.Generator().manual_seed(2)
    a = torch.randn(70, 33, generator=g).to(cuda)                       # transposed (odd sizes: partial tiles)
    b = torch.randn(1000, generator=g).to(cuda)                         # mapped with a second, subtracted term and holes
    m1 = torch.randint(-1, 1000, (1537,), generator=g, dtype=torch.int32).to(cuda)
    m2 = torch.randint(-1, 1000, (1537,), generator=g, dtype=torch.int32).to(cuda)
    out_t = torch.empty(33 * 70, device=cuda)
    out_m = torch.empty(1537, device=cuda)
    acc = torch.randn(1537, generator=g).to(cuda)
    acc0 = acc.clone()
    jobs = (GatherJob * 3)(GatherJob(a.data_ptr(), out_t.data_ptr(), None, None, a.numel(), -1.0, 0, 33),
                           GatherJob(b.data_ptr(), out_m.data_ptr(), m1.data_ptr(), m2.data_ptr(), 1537, -1.0, 0, 0),
                           GatherJob(b.data_ptr(), acc.data_ptr(), m1.data_ptr(), None, 1537, -1.0, 1, 0))
    table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(cuda)
    _lib.check(L.pcrcg_gather_jobs(table.data_ptr(), 3, a.numel(), ops._stream()), "pcrcg_gather_jobs")
    torch.cuda.synchronize()
    assert torch.equal(out_t.view(33, 70), a.t().contiguous())
    pick = lambda m: torch.where(m >= 0, b[m.clamp(min=0).long()], torch.zeros((), device=cuda))
    assert torch.equal(out_m, pick(m1) - pick(m2))
    assert torch.equal(acc, acc0 + pick(m1))
    _lib.check(L.pcrcg_gather_jobs(None, 0, 0, ops._stream()), "pcrcg_gather_jobs")          # an empty table is a no-op
