"""GPU: the BASELINE.json configurations at their real sizes.

  configs[1]  S30k pair, full-width KPFCNN+GCN forward, against outputs of the UNMODIFIED reference model on the
              reference's own collate (tests/golden/model_s30k.pt, scripts/make_golden_scale.py), 1e-4.
  configs[2]  3DLoMatch-shaped S30k pair: forward against the reference fixture, then one full train step
              (forward with tape + MetricLoss + backward + SGD) whose loss values and gradients are compared with the
              CPU oracle under torch autograd.
  configs[3]  stand-in on one GPU: eight S30k pairs through the pair engine equal the sequential results, and a
              train step whose gradient bucket goes through a one-rank RCCL all-reduce equals the step without it.
  configs[4]  K120k pair, KITTI architecture at full width, against outputs of the UNMODIFIED reference model
              (tests/golden/model_k120k.pt), 1e-4, and the GNN's kNN rows against the reference's; K120k / U30k front
              ends against raw digests of the reference C++ (tests/test_pairstream_gpu.py holds C1 / S30k / T8k).
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as MR
from pcrcg_amd import indoor_config, kitti_config, synthetic
from pcrcg_amd.architectures import KPFCNN
from pcrcg_amd.config import Config
from pcrcg_amd.correspondences import get_correspondences
from pcrcg_amd.loss import MetricLoss
from pcrcg_amd.pyramid import build_pyramid, collate_fn_descriptor
from pcrcg_amd.trainer import Trainer

pytestmark = pytest.mark.gpu
TOL = 1e-4
LOSS_CFG = Config(pos_margin=0.1, neg_margin=1.4, pos_radius=0.0375, safe_radius=0.1, matchability_radius=0.05,
                  max_points=256)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def _check_every_row(out, gold):
    """EVERY output row against the reference (round 5; the strided rows and means look at 1 % of them): feats_f through the
    fixture's seeded unit-vector projections (rows are unit vectors, so |p - p_ref| <= |f - f_ref|: the 1e-4 bar carries
    over), the two score vectors in full."""
    g = torch.Generator().manual_seed(123)                         # scripts/make_golden_scale.py: row_projections
    n_vec = gold["feats_proj"].shape[1]
    u = torch.randn(out["feats_f"].shape[1], n_vec, generator=g, dtype=torch.float64)
    u = u / u.norm(dim=0, keepdim=True)
    proj = out["feats_f"].double().cpu() @ u
    assert proj.shape == gold["feats_proj"].shape
    worst = float((proj - gold["feats_proj"].double()).abs().max())
    assert worst < TOL, ("feats_f, every row", worst)
    for k in ("scores_overlap", "scores_saliency"):
        if k + "_full" in gold:
            assert rel(out[k], gold[k + "_full"]) < TOL, (k, "every row")


def _seed0_model(dev):
    """The full-width model under the seeds the reference fixture used (bit-identical weights and kernel points:
    tests/test_host_logic.py::test_kernel_points)."""
    torch.manual_seed(0)
    np.random.seed(0)
    return KPFCNN(indoor_config())


def _stack(src, tgt, dev):
    return (torch.from_numpy(np.concatenate([src, tgt])).to(dev),
            torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev))


@pytest.fixture(scope="module")
def net(cuda):
    return _seed0_model(cuda).to(cuda).eval()


def test_s30k_full_width_outputs_vs_reference(cuda, golden_dir, net):
    gold = torch.load(os.path.join(golden_dir, "model_s30k.pt"))
    src, tgt = synthetic.pair("S30k", gold["seed"])
    batch = build_pyramid(*_stack(src, tgt, cuda), indoor_config(), gold["limits"])
    assert [int(p.shape[0]) for p in batch["points"]] == gold["levels"]
    with torch.no_grad():
        out = net(batch)
    torch.cuda.synchronize()
    s = gold["stride"]
    for k, want in gold["rows"].items():
        assert out[k][::s].shape == want.shape
        assert rel(out[k][::s], want) < TOL, k
        assert abs(float(out[k].double().mean()) - gold["means"][k]) < TOL * max(gold["absmax"][k], 1e-30), k
    _check_every_row(out, gold)
    # the op-by-op mirror agrees as well, and a few encoder activations (column means of the full tensors)
    with torch.no_grad():
        ops_out = net.forward_ops(batch)
    for k in gold["rows"]:
        assert rel(ops_out[k], out[k]) < TOL, k


def _knn_key_rows(coords, k):
    """The reference's kNN keys on the host, rounded as its CPU run rounds them (ref:models/gcn.py:15-34; the FMA chain
    of the [N,3]x[3,N] product emulated through float64): (d matrix, rows holding a tie across the k+1 cut)."""
    c = coords.cpu().numpy().astype(np.float32)
    x, y, z = c[:, 0], c[:, 1], c[:, 2]

    def fma(a, b, acc):
        return (a.astype(np.float64) * b.astype(np.float64) + acc.astype(np.float64)).astype(np.float32)

    dot = fma(z[:, None], z[None, :], fma(y[:, None], y[None, :], x[:, None] * x[None, :]))
    sa = (x * x + y * y) + z * z
    d = np.maximum((np.float32(-2) * dot + sa[:, None]) + sa[None, :], np.float32(1e-12))
    ds = np.sort(d, axis=1)
    return d, ds[:, k] == ds[:, k + 1]          # the (k+1)-th and (k+2)-th smallest are equal: the cut splits a tie group


def _check_knn(coords, want, k=10):
    """kNN index rows against the reference's get_graph_feature (ref:models/gcn.py:48-51).  Rows whose k+1 cut splits
    a group of EXACTLY equal fp32 distances are decided by torch.topk's partial sort in the reference (no defined
    order); there the distances of the kept points must agree, everywhere else the index sets."""
    from pcrcg_amd import ops
    got = ops.knn(coords, k).cpu().numpy()
    want = want.numpy()
    d, cut_tie = _knn_key_rows(coords, k)
    same_set = np.array([set(a.tolist()) == set(b.tolist()) for a, b in zip(got, want)])
    assert same_set[~cut_tie].all(), np.nonzero(~same_set & ~cut_tie)[0][:10]
    rows = np.arange(len(got))[:, None]
    assert np.array_equal(np.sort(d[rows, got], 1), np.sort(d[rows, want], 1))
    return int((~same_set).sum()), int(cut_tie.sum()), int((got == want).all(1).sum())


def test_s30k_knn_rows_vs_reference(cuda, golden_dir):
    """The GNN's neighbour graph on the ACTUAL S30k coarse clouds, against the index rows of the unmodified reference
    (tests/golden/model_s30k.pt, scripts/make_golden_scale.py): the same set in every row -- here even the same order."""
    gold = torch.load(os.path.join(golden_dir, "model_s30k.pt"))
    src, tgt = synthetic.pair("S30k", gold["seed"])
    batch = build_pyramid(*_stack(src, tgt, cuda), indoor_config(), gold["limits"])
    ns = int(batch["stack_lengths"][-1][0])
    for coords, want in ((batch["points"][-1][:ns], gold["knn_src"]), (batch["points"][-1][ns:], gold["knn_tgt"])):
        differ, cut_ties, exact = _check_knn(coords, want)
        assert differ == 0 and exact == len(want), (differ, cut_ties, exact)


def test_k120k_full_width_outputs_vs_reference(cuda, golden_dir):
    """configs[4]: the KITTI architecture (gnn_feats_dim 256, conv_radius 4.25, first_subsampling_dl 0.3) on the K120k
    pair against the UNMODIFIED reference model's own run (tests/golden/model_k120k.pt), 1e-4.  This workload is the one
    that takes the per-head GEMM attention path (1936 coarse points per cloud) and the streaming kNN kernel."""
    gold = torch.load(os.path.join(golden_dir, "model_k120k.pt"))
    cfg = kitti_config()
    src, tgt = synthetic.slab_pair(120000, gold["seed"])
    batch = build_pyramid(*_stack(src, tgt, cuda), cfg, gold["limits"])
    assert [int(p.shape[0]) for p in batch["points"]] == gold["levels"]
    ns = int(batch["stack_lengths"][-1][0])
    for coords, want in ((batch["points"][-1][:ns], gold["knn_src"]), (batch["points"][-1][ns:], gold["knn_tgt"])):
        differ, cut_ties, exact = _check_knn(coords, want)
        print("K120k kNN rows: %d of %d differ as sets, %d rows whose cut splits a tie, %d identical in order"
              % (differ, len(want), cut_ties, exact))
        # rows that hold a tie replay torch.topk's CPU selection (csrc/gnn.hip): the reference's rows, entry for entry
        assert differ == 0 and exact == len(want) and cut_ties >= 1
    torch.manual_seed(0)
    np.random.seed(0)
    model = KPFCNN(cfg).to(cuda).eval()
    with torch.no_grad():
        out = model(batch)
    torch.cuda.synchronize()
    s = gold["stride"]
    for k, want in gold["rows"].items():
        assert out[k][::s].shape == want.shape
        err = rel(out[k][::s], want)
        print("K120k", k, "max|a-b|/max|b| = %.2e" % err)
        assert err < TOL, k
        assert abs(float(out[k].double().mean()) - gold["means"][k]) < TOL * max(gold["absmax"][k], 1e-30), k
    _check_every_row(out, gold)


def test_tester_record_and_sampler_on_s30k(cuda, golden_dir, net):
    """SURVEY.md 8f rank 3 at size: the per-pair record IndoorTester.test dumps (ref:lib/tester.py:92-102) and the
    overlap x saliency sampler in front of RANSAC (:152-164), against what the reference produced for this pair."""
    from pcrcg_amd.tester import probabilistic_sample, test_record
    gold = torch.load(os.path.join(golden_dir, "model_s30k.pt"))
    src, tgt = synthetic.pair("S30k", gold["seed"])
    batch = build_pyramid(*_stack(src, tgt, cuda), indoor_config(), gold["limits"])
    batch["rot"], batch["trans"] = torch.eye(3), torch.zeros(3, 1)
    with torch.no_grad():
        out = net(batch)
    rec = test_record(batch, out)
    assert rec["len_src"] == len(src) and rec["pcd"].shape == (60000, 3) and not rec["feats"].is_cuda
    assert torch.equal(rec["pcd"], torch.from_numpy(np.concatenate([src, tgt])))
    assert rel(rec["overlaps"], gold["scores_overlap_full"]) < TOL and rel(rec["saliency"], gold["scores_saliency_full"]) < TOL
    n = len(src)
    # the reference's own scores through the sampler (device tensors in, host generator as in the reference): its draw
    sc_ref = (gold["scores_overlap_full"] * gold["scores_saliency_full"])[:n].to(cuda)
    np.random.seed(gold["sample_seed"])
    p, f, idx = probabilistic_sample(batch["points"][0][:n], out["feats_f"][:n], sc_ref, gold["sample_n"])
    assert np.array_equal(idx, gold["sample_idx_src"].numpy()) and p.shape == (gold["sample_n"], 3) and p.is_cuda
    # this path's scores: the same draw up to the few picks that sit on a cumulative-probability boundary
    np.random.seed(gold["sample_seed"])
    _, _, idx2 = probabilistic_sample(batch["points"][0][:n], out["feats_f"][:n],
                                      (out["scores_overlap"] * out["scores_saliency"])[:n], gold["sample_n"])
    same = len(set(idx2.tolist()) & set(gold["sample_idx_src"].tolist()))
    assert same >= 0.98 * gold["sample_n"], same


def test_s30k_lomatch_forward_and_train_step(cuda, golden_dir, monkeypatch):
    """configs[2] at its real size."""
    gold = torch.load(os.path.join(golden_dir, "model_s30k_lomatch.pt"))
    cfg = indoor_config()
    src, tgt, rot, trans = synthetic.lomatch_pair("S30k", gold["seed"], gold["overlap"])
    tsfm = np.eye(4)
    tsfm[:3, :3], tsfm[:3, 3] = rot, trans.flatten()
    corr = get_correspondences(torch.from_numpy(src).to(cuda), torch.from_numpy(tgt).to(cuda), tsfm, 0.0375)
    assert corr.shape[0] > 1000
    item = dict(src_pcd=src, tgt_pcd=tgt, src_feats=np.ones((len(src), 1), np.float32),
                tgt_feats=np.ones((len(tgt), 1), np.float32), rot=rot, trans=trans, correspondences=corr.cpu(), sample=0)
    inputs = collate_fn_descriptor([item], cfg, gold["limits"], device=cuda)
    assert [int(p.shape[0]) for p in inputs["points"]] == gold["levels"]
    model = _seed0_model(cuda)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(cuda)
    # forward (inference runner) against the unmodified reference model's outputs on its own collate
    model.eval()
    with torch.no_grad():
        out = model(inputs)
    for k, want in gold["rows"].items():
        assert rel(out[k][::gold["stride"]], want) < TOL, k
    _check_every_row(out, gold)
    # one train step; the CPU oracle (fp32, torch autograd) computes the same loss and gradients
    trainer = Trainer(model, MetricLoss(LOSS_CFG), lr=0.005, momentum=0.98)
    np.random.seed(5)
    stats = trainer.inference_one_batch(inputs, "train")
    grads = {n: p.grad.detach().clone().cpu() for n, p in model.named_parameters() if p.requires_grad}
    cpu_inputs = {k: ([t.cpu() if isinstance(t, torch.Tensor) else t for t in v] if isinstance(v, list)
                      else (v.cpu() if isinstance(v, torch.Tensor) else v)) for k, v in inputs.items()}
    n_src = len(src)
    # the loss on the CPU: the saliency labels' nearest-descriptor search as the reference writes it, a dense
    # matmul + arg-max (ref:lib/loss.py:207-219), in place of the fused HIP kernel
    from pcrcg_amd import loss as loss_mod
    monkeypatch.setattr(loss_mod.ops, "feature_argmax", lambda a, b, want_best=False: (a @ b.t()).argmax(1))

    def oracle(dt):
        """Network in `dt` on the CPU (torch autograd over oracle/model_ref.py), MetricLoss's torch formulation on its
        outputs (in fp32, as the reference computes it: its BCE takes float labels), gradients wrt every parameter."""
        sd = {k: (v.clone().to(dt).requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        b = {k: ([t.to(dt) if isinstance(t, torch.Tensor) and t.is_floating_point() else t for t in v] if isinstance(v, list)
                 else (v.to(dt) if isinstance(v, torch.Tensor) and v.is_floating_point() else v)) for k, v in cpu_inputs.items()}
        o = MR.kpfcnn_forward_with_grad(sd, dict(cfg), b)
        np.random.seed(5)
        res = MetricLoss(LOSS_CFG)({"src_feats": o["feats_f"][:n_src].float(), "tgt_feats": o["feats_f"][n_src:].float(),
                                    "rot": cpu_inputs["rot"], "trans": cpu_inputs["trans"],
                                    "scores_overlap": o["scores_overlap"].float(), "scores_saliency": o["scores_saliency"].float(),
                                    "src_pcd_raw": cpu_inputs["src_pcd_raw"], "tgt_pcd_raw": cpu_inputs["tgt_pcd_raw"],
                                    "correspondences": cpu_inputs["correspondences"]})
        sum(res[k] for k in res if k in ("circle_loss", "overlap_loss", "saliency_loss")).backward()
        return res, sd

    res, s32 = oracle(torch.float32)
    for k in ("circle_loss", "overlap_loss", "saliency_loss"):
        assert abs(stats[k] - float(res[k].detach())) < 2e-3 * max(abs(float(res[k].detach())), 1e-3), (k, stats[k], float(res[k].detach()))
    # Gradients: MEASURED against float64 at this size (round 3 asserted the band from the C1 measurement).  fp32 through
    # ~60 layers with data-dependent LeakyReLU / max-pool / neighbour-count decisions sits a few 1e-3 from float64 per
    # parameter tensor whoever computes it; the HIP path must be no further from the float64 oracle than 3x what the fp32
    # CPU oracle is, percentile by percentile (the form of tests/test_train_step_gpu.py::test_full_width_gradients_c1_vs_oracle)
    _, s64 = oracle(torch.float64)
    names = list(grads)
    floor = 1e-4 * max(float(s64[n].grad.abs().max()) for n in names)

    def errs(get):
        return np.array([float((get(n).double() - s64[n].grad).abs().max() / max(float(s64[n].grad.abs().max()), floor))
                         for n in names])
    e_hip, e_ref = errs(lambda n: grads[n]), errs(lambda n: s32[n].grad)
    print("S30k-lomatch gradient errors vs float64: HIP median %.2e p90 %.2e max %.2e | fp32 CPU oracle median %.2e p90 %.2e max %.2e"
          % (np.median(e_hip), np.percentile(e_hip, 90), e_hip.max(), np.median(e_ref), np.percentile(e_ref, 90), e_ref.max()))
    assert np.median(e_hip) <= 3 * np.median(e_ref), (np.median(e_hip), np.median(e_ref))
    assert np.percentile(e_hip, 90) <= 3 * np.percentile(e_ref, 90), (np.percentile(e_hip, 90), np.percentile(e_ref, 90))
    assert e_hip.max() <= 3 * e_ref.max() + 1e-2, (e_hip.max(), e_ref.max())
    assert trainer.optimizer_step() is True


def test_eight_pairs_and_one_rank_rccl_step(cuda, net):
    """configs[3] stand-in on one GPU."""
    from pcrcg_amd.pairstream import PairStreams
    cfg = indoor_config()
    limits = synthetic.LIMITS["S30k"]
    pairs = [_stack(*synthetic.pair("S30k", seed), cuda) for seed in range(8)]
    ref = []
    with torch.no_grad():
        for pts, lens in pairs:
            ref.append(net(build_pyramid(pts, lens, cfg, limits)))
    torch.cuda.synchronize()
    eng = PairStreams(net, cfg, limits, cuda, adaptive_jobs=False)    # grouping fixed by the configuration, not by timing
    for pts, lens in pairs:
        eng.submit(pts, lens)
    outs = [eng.result() for _ in pairs]
    eng.drain()
    eng.close()
    for i, (a, b) in enumerate(zip(outs, ref)):
        for k in ("feats_f", "scores_overlap", "scores_saliency"):
            assert rel(a[k], b[k]) < 1e-5, (i, k)
    # train step with the gradient bucket going through RCCL (one-rank group) == the same step without the exchange
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=cuda)
        created = True
    try:
        small = indoor_config(first_feats_dim=64, gnn_feats_dim=128)
        src, tgt, rot, trans = synthetic.lomatch_pair("C1", 3, 0.3)
        tsfm = np.eye(4)
        tsfm[:3, :3], tsfm[:3, 3] = rot, trans.flatten()
        corr = get_correspondences(torch.from_numpy(src).to(cuda), torch.from_numpy(tgt).to(cuda), tsfm, 0.0375)
        item = dict(src_pcd=src, tgt_pcd=tgt, src_feats=np.ones((len(src), 1), np.float32),
                    tgt_feats=np.ones((len(tgt), 1), np.float32), rot=rot, trans=trans, correspondences=corr.cpu(), sample=0)
        inputs = collate_fn_descriptor([item], small, synthetic.LIMITS["C1"], device=cuda)
        results = []
        for force in ("1", "0"):
            os.environ["PCRCG_FORCE_DIST"] = force
            torch.manual_seed(0)
            np.random.seed(0)
            m = KPFCNN(small).to(cuda)
            t = Trainer(m, MetricLoss(LOSS_CFG))
            assert t.bucket._forced() == (force == "1")
            np.random.seed(7)
            st = t.train_step(inputs)
            assert st["gradient_valid"] == 1.0
            results.append({k: v.detach().clone() for k, v in m.state_dict().items()})
        for k in results[0]:
            if results[0][k].is_floating_point():     # (backward scatters with fp32 atomics: equal up to summation order)
                # (parameters whose true gradient is zero -- biases in front of an InstanceNorm -- move by rounding
                # noise times the learning rate on both sides: absolute floor)
                d = float((results[0][k] - results[1][k]).abs().max())
                assert d <= 1e-4 * float(results[1][k].abs().max()) + 5e-6, (k, d)
    finally:
        os.environ.pop("PCRCG_FORCE_DIST", None)
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("recipe", ["U30k", "K120k"])
def test_front_end_digests_u30k_k120k(cuda, golden_dir, recipe):
    """configs[4] (and the uniform-cube shape): every level and every untruncated table of the reference C++ front
    end, by SHA-256."""
    dig = json.load(open(os.path.join(golden_dir, "frontend_digests.json")))[recipe]
    if recipe == "K120k":
        cfg, (src, tgt) = kitti_config(), synthetic.slab_pair(120000, 0)
    else:
        cfg, (src, tgt) = indoor_config(), synthetic.uniform_pair(30000, 1.07, 0)
    b = build_pyramid(*_stack(src, tgt, cuda), cfg, [400, 400, 400, 400])

    def sha(t, dtype):
        return hashlib.sha256(np.ascontiguousarray(t.cpu().numpy().astype(dtype)).tobytes()).hexdigest()

    for l in range(cfg.num_layers):
        assert sha(b["points"][l], np.float32) == dig[f"points{l}"]["sha256"], l
        assert sha(b["stack_lengths"][l], np.int32) == dig[f"lens{l}"]["sha256"], l
        for key, name in (("neighbors", "conv"), ("pools", "pool"), ("upsamples", "up")):
            if f"{name}{l}" in dig:
                t = b[key][l]
                assert list(t.shape) == dig[f"{name}{l}"]["shape"], (name, l, t.shape)
                assert sha(t, np.int32) == dig[f"{name}{l}"]["sha256"], (name, l)


@pytest.mark.parametrize("recipe", ["K120k", "U30k"])
def test_secondary_workloads_forward_runner_equals_mirror(cuda, recipe):
    """configs[4]-shaped pairs through the whole path at full width: the C++ runner (one-launch attention only below
    ~1000 x 1000 coarse points -- K120k has 1936 per cloud and takes the GEMM path --, column-sum statistics, fused decoder
    products) against the op-by-op mirror, which uses none of those."""
    cfg = kitti_config() if recipe == "K120k" else indoor_config()
    a, b = synthetic.slab_pair(120000, 0) if recipe == "K120k" else synthetic.uniform_pair(30000, 1.07, 0)
    pts, lens = _stack(a, b, cuda)
    from pcrcg_amd.pyramid import calibrate_neighbors
    limits = [int(v) for v in calibrate_neighbors([(pts, lens)], cfg, samples_threshold=0)]
    torch.manual_seed(0)
    np.random.seed(0)
    model = KPFCNN(cfg).to(cuda).eval()
    batch = build_pyramid(pts, lens, cfg, limits)
    with torch.no_grad():
        out = model(batch)
        ref = model.forward_ops(batch)
    torch.cuda.synchronize()
    for k in ref:
        assert torch.isfinite(out[k]).all() and rel(out[k], ref[k]) < 1e-5, (recipe, k)


def test_s30k_image129_shipped_configuration_vs_reference(cuda, golden_dir):
    """PCR-CG's SHIPPED configuration at full width and full size (ref:configs/test/indoor.yaml:21-34: image_feature True,
    img_num 2, in_feats_dim 129; ref:models/architectures.py:195-514): the reference model's outputs on the S30k pair with the
    synthetic 2-D inputs of pcrcg_amd.synthetic.image_inputs (tests/golden/model_s30k_img129.pt, scripts/
    make_golden_image_s30k.py) against this path -- the injected [N, 129] matrix bit for bit, the runner's outputs (first
    KPConv over rows of 132 floats) at 1e-4 on the strided rows and through every row's projections, and the pair engine
    carrying the same pair (submit(images=...)) equal to the direct forward."""
    from pcrcg_amd.pairstream import PairStreams
    gold = torch.load(os.path.join(golden_dir, "model_s30k_img129.pt"))
    cfg = indoor_config(image_feature=True, img_num=2, in_feats_dim=129)
    torch.manual_seed(0)
    np.random.seed(0)
    net = KPFCNN(cfg).eval()
    for k, v in gold["weights_check"].items():          # same seeds -> the reference's weights, bit for bit
        assert torch.equal(net.state_dict()[k], v), k
    net = net.to(cuda)
    src, tgt = synthetic.pair("S30k", 0)
    pts, lens = _stack(src, tgt, cuda)
    batch = build_pyramid(pts, lens, cfg, gold["limits"])
    assert [int(p.shape[0]) for p in batch["points"]] == gold["levels"]
    for k, v in synthetic.image_inputs(len(src), len(tgt), 0, img_num=2).items():
        batch[k] = torch.from_numpy(v).to(cuda)
    batch["src_pcd_raw"], batch["tgt_pcd_raw"] = pts[:len(src)], pts[len(src):]
    s = gold["stride"]
    x = net.image_features(batch)
    assert torch.equal(x[::s].cpu(), gold["x_rows"])
    assert int((x[:, :128] != 1).any(1).sum()) == gold["x_rows_with_image_features"]
    with torch.no_grad():
        out = net(batch)
    torch.cuda.synchronize()
    for k, want in gold["rows"].items():
        assert rel(out[k][::s], want) < TOL, k
        assert abs(float(out[k].double().mean()) - gold["means"][k]) < TOL, k
    _check_every_row(out, gold)
    _, _, images = net.image_list(batch)
    eng = PairStreams(net, cfg, gold["limits"], cuda)
    for _ in range(5):
        eng.submit(pts, lens, images=images)
    outs = [eng.result() for _ in range(5)]
    eng.drain()
    eng.close()
    for o in outs:
        for k in gold["rows"]:
            assert float((o[k] - out[k]).abs().max()) <= 1e-5 * float(out[k].abs().max()), k
