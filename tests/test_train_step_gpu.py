"""GPU: the differentiable forward (pcrcg_amd/train_forward.py) and the train step (pcrcg_amd/trainer.py)
against the CPU oracle (oracle/model_ref.py) under torch autograd, on the reduced-width reference model of
tests/golden/model_mini.pt."""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as MR
from pcrcg_amd import indoor_config, synthetic
from pcrcg_amd.architectures import KPFCNN
from pcrcg_amd.config import Config
from pcrcg_amd.correspondences import get_correspondences
from pcrcg_amd.loss import MetricLoss
from pcrcg_amd.pyramid import collate_fn_descriptor
from pcrcg_amd.train_forward import forward_train
from pcrcg_amd.trainer import Trainer

pytestmark = pytest.mark.gpu
LOSS_CFG = Config(pos_margin=0.1, neg_margin=1.4, pos_radius=0.0375, safe_radius=0.1, matchability_radius=0.05,
                  max_points=256)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def _to(batch, dev):
    out = {}
    for k, v in batch.items():
        if isinstance(v, list):
            out[k] = [t.to(dev) if isinstance(t, torch.Tensor) else t for t in v]
        elif isinstance(v, torch.Tensor):
            out[k] = v.to(dev)
        else:
            out[k] = v
    return out


def _train_forward(net, batch, path):
    """path "cpp": the C++ tape runner (csrc/train_runner.hip, one autograd node); "python": the op-by-op autograd
    composition of pcrcg_amd/train_forward.py, its mirror."""
    if path == "cpp":
        runner = net.train_runner()
        assert runner is not None
        return runner.forward(batch)
    return forward_train(net, batch)


@pytest.mark.parametrize("path", ["cpp", "python"])
def test_full_model_gradients_match_oracle_autograd(cuda, golden_dir, path):
    """d(scalar)/d(every parameter) through encoder, GNN, saliency head and decoder."""
    gold = torch.load(os.path.join(golden_dir, "model_mini.pt"))
    col = torch.load(os.path.join(golden_dir, "collate_mini.pt"))
    cfg = indoor_config(**{k: v for k, v in gold["config"].items() if k in ("first_feats_dim", "gnn_feats_dim")})
    net = KPFCNN(cfg)
    net.load_state_dict(gold["state_dict"])
    net = net.to(cuda).train()
    batch = _to(col["batch"], cuda)
    n = batch["points"][0].shape[0]
    g = torch.Generator().manual_seed(1)
    r1, r2, r3 = torch.randn(n, 32, generator=g), torch.randn(n, generator=g), torch.randn(n, generator=g)

    def scalar(out, dev):
        return (out["feats_f"] * r1.to(dev)).sum() + (out["scores_overlap"] * r2.to(dev)).sum() \
            + (out["scores_saliency"] * r3.to(dev)).sum()

    out = _train_forward(net, batch, path)
    for k in gold["outputs"]:                                     # the forward values still match the reference
        assert rel(out[k], gold["outputs"][k]) < 1e-4, k
    scalar(out, cuda).backward()

    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in gold["state_dict"].items()}
    out0 = MR.kpfcnn_forward_with_grad(sd, dict(gold["config"]), col["batch"])
    scalar(out0, "cpu").backward()
    worst = {}
    names = [n for n, p in net.named_parameters() if p.requires_grad]   # kernel points are fixed (ref:blocks.py:196-198)
    # biases in front of an InstanceNorm / inside the softmax have an exactly zero gradient: both sides return
    # rounding noise there, so errors are measured against max(|reference grad|, 1e-4 * largest gradient)
    floor = 1e-4 * max(float(sd[n].grad.abs().max()) for n in names)
    params = dict(net.named_parameters())
    for name in names:
        p, want = params[name], sd[name].grad
        assert p.grad is not None and want is not None, name
        worst[name] = float((p.grad.double().cpu() - want.double()).abs().max() / max(float(want.abs().max()), floor))
    bad = {k: v for k, v in worst.items() if v > 1e-3}
    assert not bad, bad
    assert np.median(list(worst.values())) < 1e-4, sorted(worst.items(), key=lambda kv: -kv[1])[:5]


def test_flat_sgd_is_torch_sgd(cuda):
    """trainer.FlatSGD (one launch over the flat parameter / gradient buffers, pcrcg_sgd_step) against torch.optim.SGD with
    the train loop's hyper-parameters over four steps; the fused gradient clearing; parameter slices 256-byte aligned."""
    from pcrcg_amd.trainer import FlatSGD, GradientBucket
    g = torch.Generator().manual_seed(3)
    shapes = [(7, 5), (1,), (33,), (64, 3, 2), (130,)]
    ref = [torch.nn.Parameter(torch.randn(*s, generator=g).to(cuda)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    bucket = GradientBucket(mine)
    flat = FlatSGD.flatten(mine, bucket.sizes)
    assert all(p.data_ptr() % 256 == 0 and p.grad.data_ptr() % 256 == 0 for p in mine)
    opt = FlatSGD(mine, flat, bucket.flat, lr=0.005, momentum=0.98, weight_decay=1e-6)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.95)
    topt = torch.optim.SGD(ref, lr=0.005, momentum=0.98, weight_decay=1e-6)
    tsched = torch.optim.lr_scheduler.ExponentialLR(topt, gamma=0.95)
    for step in range(4):
        for p, q in zip(ref, mine):
            gr = torch.randn(p.shape, generator=g).to(cuda)
            p.grad = gr.clone()
            q.grad.copy_(gr)
        topt.step()
        opt.step(zero_grad=(step % 2 == 0))
        if step % 2 == 0:
            assert float(bucket.flat.abs().sum()) == 0.0
        tsched.step()
        sched.step()
        for p, q in zip(ref, mine):
            assert float((p.data - q.data).abs().max()) <= 1e-6 * float(p.data.abs().max()), step
    # ---- checkpoints are interchangeable with torch.optim.SGD's (ref:lib/trainer.py:133,174 saves / loads the optimizer's
    # state_dict): FlatSGD -> torch, torch (mapped to the CPU, as a checkpoint is) -> FlatSGD, then both step on identically
    sd = opt.state_dict()
    assert set(sd["state"]) == set(range(len(mine))) and all(sd["state"][i]["momentum_buffer"].shape == mine[i].shape for i in sd["state"])
    topt2 = torch.optim.SGD([torch.nn.Parameter(q.detach().clone()) for q in mine], lr=1.0)
    topt2.load_state_dict(sd)                                  # torch accepts the format, hyper-parameters included
    assert topt2.param_groups[0]["momentum"] == 0.98 and abs(topt2.param_groups[0]["lr"] - opt.param_groups[0]["lr"]) < 1e-12
    tsd = {"state": {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in topt.state_dict()["state"].items()},
           "param_groups": topt.state_dict()["param_groups"]}
    mine2 = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    bucket2 = GradientBucket(mine2)
    opt2 = FlatSGD(mine2, FlatSGD.flatten(mine2, bucket2.sizes), bucket2.flat, lr=1.0, momentum=0.0, sizes=bucket2.sizes)
    opt2.load_state_dict(tsd)
    assert opt2.momentum_flat.is_cuda and opt2.param_groups[0]["momentum"] == 0.98          # copied INTO the device buffer
    for p, q in zip(ref, mine2):
        gr = torch.randn(p.shape, generator=g).to(cuda)
        p.grad = gr.clone()
        q.grad.copy_(gr)
    topt.step()
    opt2.step()
    for p, q in zip(ref, mine2):
        assert float((p.data - q.data).abs().max()) <= 1e-6 * float(p.data.abs().max())     # momentum was NOT reset
    # round-3 checkpoints (one flat buffer) still load; wrong sizes are refused
    opt2.load_state_dict({"state": {"flat": {"momentum_buffer": opt.momentum_flat.cpu()}}, "param_groups": sd["param_groups"]})
    assert torch.equal(opt2.momentum_flat, opt.momentum_flat)
    with pytest.raises(ValueError):
        opt2.load_state_dict({"state": {"flat": {"momentum_buffer": torch.zeros(3)}}, "param_groups": sd["param_groups"]})


def test_partly_frozen_model_trains_the_rest(cuda, golden_dir):
    """Frozen parameters (requires_grad = False) -- `epsilon`, the node's former only differentiable input, among them --
    get NO gradient tensor and no dW product (NULL gradient pointer in the C ABI), and every other parameter gets exactly
    the gradient it gets in the unfrozen model."""
    gold = torch.load(os.path.join(golden_dir, "model_mini.pt"))
    col = torch.load(os.path.join(golden_dir, "collate_mini.pt"))
    cfg = indoor_config(**{k: v for k, v in gold["config"].items() if k in ("first_feats_dim", "gnn_feats_dim")})
    batch = _to(col["batch"], cuda)
    grads = {}
    frozen = ("epsilon", "encoder_blocks.1.KPConv.weights", "decoder_blocks.1.mlp.weight", "gnn.layers.0.conv1.weight",
              "gnn.layers.1.attn.proj.0.weight", "bottle.bias")
    for freeze in (False, True):
        net = KPFCNN(cfg)
        net.load_state_dict(gold["state_dict"])
        net = net.to(cuda).train()
        params = dict(net.named_parameters())
        if freeze:
            for name in frozen:
                params[name].requires_grad_(False)
        out = net.train_runner().forward(batch)
        assert out["feats_f"].grad_fn is not None                  # (epsilon frozen: the node is still in the graph)
        (out["feats_f"].square().sum() + out["scores_overlap"].sum() + 2.0 * out["scores_saliency"].sum()).backward()
        grads[freeze] = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in params.items()}
    # (biases in front of an InstanceNorm have an exactly zero gradient: both runs return rounding noise there, and the
    # atomically accumulated products differ in their last bits from run to run -- errors are measured against a floor)
    floor = 1e-4 * max(float(v.abs().max()) for v in grads[False].values() if v is not None)
    for name, want in grads[False].items():
        got = grads[True][name]
        if name in frozen:
            assert got is None, name                               # no .grad allocated, nothing accumulated
        elif want is None:                                         # (kernel_points: never trainable)
            assert got is None, name
        else:
            assert got is not None, name
            assert float((got - want).abs().max()) <= 1e-5 * max(float(want.abs().max()), floor), name


def test_weight_gradients_on_the_second_stream_equal_one_stream(cuda, golden_dir):
    """train_side_stream (default on): the backward's weight-gradient products run on a second stream; same gradients as
    with everything on one stream, and complete when backward() returns to the caller's stream."""
    from pcrcg_amd import _lib
    gold = torch.load(os.path.join(golden_dir, "model_mini.pt"))
    col = torch.load(os.path.join(golden_dir, "collate_mini.pt"))
    cfg = indoor_config(**{k: v for k, v in gold["config"].items() if k in ("first_feats_dim", "gnn_feats_dim")})
    net = KPFCNN(cfg)
    net.load_state_dict(gold["state_dict"])
    net = net.to(cuda).train()
    batch = _to(col["batch"], cuda)
    runner = net.train_runner()
    grads = {}
    try:
        for side in (1, 0, 1):
            _lib.check(_lib.lib().pcrcg_debug_set(f"train_side_stream={side}".encode()), "pcrcg_debug_set")
            for p in net.parameters():
                p.grad = None
            out = runner.forward(batch)
            (out["feats_f"].sum() + (out["scores_overlap"] * 2.0).sum() + out["scores_saliency"].sum()).backward()
            # read on the caller's stream right away: the join inside backward orders the side stream before this
            grads[side] = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    finally:
        _lib.check(_lib.lib().pcrcg_debug_set(b"train_side_stream=1"), "pcrcg_debug_set")
    floor = 1e-5 * max(float(g.abs().max()) for g in grads[0].values())
    for n, ref in grads[0].items():
        assert float((grads[1][n] - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + floor, n


def test_two_outstanding_tapes_and_a_dropped_one(cuda, golden_dir):
    """The C++ runner's workspace discipline: a forward issued while an earlier one still awaits its backward gets its own
    workspace (both backwards are right: the gradients add up to twice one backward's), a forward that is dropped without a
    backward frees its tape, and a second backward through one node raises."""
    gold = torch.load(os.path.join(golden_dir, "model_mini.pt"))
    col = torch.load(os.path.join(golden_dir, "collate_mini.pt"))
    cfg = indoor_config(**{k: v for k, v in gold["config"].items() if k in ("first_feats_dim", "gnn_feats_dim")})
    net = KPFCNN(cfg)
    net.load_state_dict(gold["state_dict"])
    net = net.to(cuda).train()
    batch = _to(col["batch"], cuda)
    runner = net.train_runner()

    def scalar(out):
        return out["feats_f"].sum() + (out["scores_overlap"] * 2.0).sum() + out["scores_saliency"].sum()

    scalar(runner.forward(batch)).backward()
    once = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    for p in net.parameters():
        p.grad = None
    o1 = runner.forward(batch)
    o2 = runner.forward(batch)                    # o1's tape is outstanding: a workspace of its own
    assert o1["feats_f"].data_ptr() != o2["feats_f"].data_ptr()
    v1 = o1["feats_f"].clone()
    scalar(o2).backward()
    assert torch.equal(o1["feats_f"], v1)         # ... so o1's values are untouched
    scalar(o1).backward()
    # (parameters whose true gradient is zero -- biases in front of an InstanceNorm -- hold rounding noise on both sides:
    # the bar is relative to the largest gradient of the model)
    floor = 1e-4 * max(float(g.abs().max()) for g in once.values())
    for n, p in net.named_parameters():
        if n in once:
            ref = 2.0 * once[n]
            assert float((p.grad - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + floor, n
    dropped = runner.forward(batch)
    del dropped                                   # no backward: the tape goes with the autograd node
    import gc
    gc.collect()
    assert runner._lent is None
    out = runner.forward(batch)
    loss = scalar(out)
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError):
        loss.backward()


def _lomatch_inputs(cfg, dev, seed=2):
    src, tgt, rot, trans = synthetic.lomatch_pair("mini", seed, overlap=0.3)
    tsfm = np.eye(4)
    tsfm[:3, :3], tsfm[:3, 3] = rot, trans.flatten()
    corr = get_correspondences(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), tsfm, 0.0375)
    item = dict(src_pcd=src, tgt_pcd=tgt, src_feats=np.ones((len(src), 1), np.float32),
                tgt_feats=np.ones((len(tgt), 1), np.float32), rot=rot, trans=trans, correspondences=corr.cpu(), sample=0)
    return collate_fn_descriptor([item], cfg, [20, 26, 30, 32], device=dev)


def test_train_step_reduces_the_loss_and_updates_every_parameter(cuda):
    cfg = indoor_config(first_feats_dim=32, gnn_feats_dim=64)
    torch.manual_seed(0)
    np.random.seed(0)
    net = KPFCNN(cfg).to(cuda)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    trainer = Trainer(net, MetricLoss(LOSS_CFG), lr=0.005, momentum=0.98)
    inputs = _lomatch_inputs(cfg, cuda)
    losses = []
    for _ in range(12):
        np.random.seed(3)                      # same max_points subset every step: the loss is comparable
        stats = trainer.train_step(inputs)
        assert stats["gradient_valid"] == 1.0
        for k in ("circle_loss", "overlap_loss", "saliency_loss", "recall", "total_loss"):
            assert np.isfinite(stats[k]), k
        losses.append(stats["total_loss"])
    assert losses[-1] < losses[0], losses
    changed = [k for k, v in net.state_dict().items() if v.is_floating_point() and not torch.equal(v, before[k])]
    # (the scalar temperature parameter `epsilon` = -5 moves by less than one fp32 ulp per step at lr 0.005)
    trainable = [k for k, p in net.named_parameters() if p.requires_grad and k != "epsilon"]
    assert set(trainable) <= set(changed), set(trainable) - set(changed)
    # evaluation goes through the inference runner and agrees with the training forward
    np.random.seed(3)
    val = trainer.inference_one_batch(inputs, "val")
    assert "total_loss" not in val and np.isfinite(val["recall"])
    trainer.end_epoch()
    assert abs(trainer.optimizer.param_groups[0]["lr"] - 0.005 * 0.95) < 1e-12


def test_non_finite_gradients_skip_the_step(cuda):
    cfg = indoor_config(first_feats_dim=32, gnn_feats_dim=64)
    torch.manual_seed(0)
    net = KPFCNN(cfg).to(cuda)
    trainer = Trainer(net, MetricLoss(LOSS_CFG))
    before = {k: v.clone() for k, v in net.state_dict().items()}
    trainer.flat_grad[5] = float("nan")
    assert trainer.optimizer_step() is False and trainer.skipped_steps == 1
    assert all(torch.equal(v, before[k]) for k, v in net.state_dict().items())
    assert float(trainer.flat_grad.abs().sum()) == 0.0          # bucket cleared for the next pair


@pytest.mark.parametrize("path", ["cpp", "python"])
def test_full_width_gradients_c1_vs_oracle(cuda, path):
    """The gradient check at the real channel widths (29.7 M parameters, Cin up to 2048 in the 1x1 convolutions,
    KPConv widths up to 512) on a C1 pair: exercises the multi-block gather variants, the split-K A^T products
    with K = number of points and the 1538 / 769-wide decoder GEMMs.  At this depth fp32 itself is the limit:
    the fp32 CPU oracle differs from the fp64 CPU oracle by a median 3e-3 per parameter tensor (LeakyReLU /
    max-pool decisions flip, InstanceNorm amplifies), so the bar is relative to that noise: the HIP gradients
    must be no further from the fp64 oracle than 3x what the fp32 oracle is."""
    from pcrcg_amd.pyramid import build_pyramid
    cfg = indoor_config()
    torch.manual_seed(0)
    np.random.seed(0)
    net = KPFCNN(cfg)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(cuda).train()
    src, tgt = synthetic.pair("C1", 0)
    pts = torch.from_numpy(np.concatenate([src, tgt])).to(cuda)
    lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=cuda)
    batch = build_pyramid(pts, lens, cfg, synthetic.LIMITS["C1"])
    n = pts.shape[0]
    g = torch.Generator().manual_seed(1)
    r1, r2, r3 = torch.randn(n, 32, generator=g), torch.randn(n, generator=g), torch.randn(n, generator=g)

    def scalar(out, dev, dt=torch.float32):
        return (out["feats_f"] * r1.to(dev).to(dt)).sum() + (out["scores_overlap"] * r2.to(dev).to(dt)).sum() \
            + (out["scores_saliency"] * r3.to(dev).to(dt)).sum()

    out = _train_forward(net, batch, path)
    scalar(out, cuda).backward()
    cpu_batch = {k: ([t.cpu() if isinstance(t, torch.Tensor) else t for t in v] if isinstance(v, list)
                     else (v.cpu() if isinstance(v, torch.Tensor) else v)) for k, v in batch.items()}

    def oracle(dt):
        sd = {k: (v.clone().to(dt).requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        b = {k: ([t.to(dt) if isinstance(t, torch.Tensor) and t.is_floating_point() else t for t in v]
                 if isinstance(v, list) else (v.to(dt) if isinstance(v, torch.Tensor) and v.is_floating_point() else v))
             for k, v in cpu_batch.items()}
        o = MR.kpfcnn_forward_with_grad(sd, dict(cfg), b)
        scalar(o, "cpu", dt).backward()
        return o, sd

    o32, s32 = oracle(torch.float32)
    o64, s64 = oracle(torch.float64)
    for k in o32:
        assert rel(out[k], o32[k]) < 1e-4, k                      # forward parity at full width
    names = [nme for nme, p in net.named_parameters() if p.requires_grad]
    params = dict(net.named_parameters())
    floor = 1e-4 * max(float(s64[nme].grad.abs().max()) for nme in names)

    def errs(grads):
        return np.array([float((grads[nme].double().cpu() - s64[nme].grad).abs().max()
                               / max(float(s64[nme].grad.abs().max()), floor)) for nme in names])

    e_hip = errs({nme: params[nme].grad for nme in names})
    e_ref = errs({nme: s32[nme].grad for nme in names})
    assert np.median(e_hip) < 3 * np.median(e_ref) and np.median(e_hip) < 2e-2, (np.median(e_hip), np.median(e_ref))
    assert np.percentile(e_hip, 90) < 3 * np.percentile(e_ref, 90), (np.percentile(e_hip, 90), np.percentile(e_ref, 90))
    assert e_hip.max() < 3 * e_ref.max() + 1e-2, (e_hip.max(), e_ref.max())
