"""GPU: PCR-CG's image-feature injection (ref:models/architectures.py:195-514) -- the HIP gather /
scatter of per-pixel 2-D features into the [N,129] point features and the forward that consumes them -- against the
UNMODIFIED reference model run with a stand-in 2-D backbone (tests/golden/image_mini.pt)."""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as MR
from pcrcg_amd import indoor_config
from pcrcg_amd.architectures import KPFCNN

pytestmark = pytest.mark.gpu


def _to(v, dev):
    if isinstance(v, list):
        return [t.to(dev) if isinstance(t, torch.Tensor) else t for t in v]
    return v.to(dev) if isinstance(v, torch.Tensor) else v


@pytest.mark.parametrize("img_num", [1, 2, 3])
def test_image_feature_forward_vs_reference(cuda, golden_dir, img_num):
    gold = torch.load(os.path.join(golden_dir, "image_mini.pt"))[f"img{img_num}"]
    col = torch.load(os.path.join(golden_dir, "collate_mini.pt"))["batch"]
    cfg = indoor_config(first_feats_dim=32, gnn_feats_dim=64, image_feature=True, img_num=img_num, in_feats_dim=129)
    torch.manual_seed(gold["seed"])
    np.random.seed(gold["seed"])
    net = KPFCNN(cfg).eval()
    for k, v in gold["weights_check"].items():          # same seeds -> the reference's weights, bit for bit
        assert torch.equal(net.state_dict()[k], v), k
    net = net.to(cuda)
    n_src = int(col["stack_lengths"][0][0])
    batch = {k: _to(v, cuda) for k, v in col.items()}
    batch["src_pcd_raw"], batch["tgt_pcd_raw"] = batch["points"][0][:n_src], batch["points"][0][n_src:]
    for k, v in gold["inputs"].items():                  # precomputed 2-D feature maps, indices, valid maps
        batch[k] = v.to(cuda)
    x = net.image_features(batch)
    assert x.shape == (batch["points"][0].shape[0], 129)
    assert torch.equal(x[::gold["x_stride"]].cpu(), gold["x_rows"])          # copies and one multiply: bit-exact
    with torch.no_grad():
        out = net(batch)
        out_ops = net.forward_ops({**batch, "features": x})
    for k, want in gold["outputs"].items():
        assert MR.rel_err(out[k].cpu(), want) < 1e-4, k
        assert MR.rel_err(out_ops[k].cpu(), want) < 1e-4, k
    # the same maps through a `backbone2d` callable, as the reference passes them (forward(batch, backbone2d))
    batch2 = {k: v for k, v in batch.items() if not k.endswith("_feature2d")}
    lookup = {}
    for s in ("src", "tgt"):
        for i in range(1, img_num + 1):
            color = torch.full((3, 2, 2), float(len(lookup)), device=cuda)       # a tag the stand-in backbone recognises
            batch2[f"{s}_color{i}"] = color
            lookup[float(len(lookup))] = batch[f"{s}{i}_feature2d"]
    with torch.no_grad():
        out2 = net(batch2, backbone2d=lambda c: lookup[float(c.flatten()[0])].unsqueeze(0))
    for k in gold["outputs"]:
        assert torch.equal(out2[k], out[k]) or MR.rel_err(out2[k].cpu(), out[k].cpu()) < 1e-5, k


def _mini_image_case(cuda, golden_dir, img_num=2, train=False):
    gold = torch.load(os.path.join(golden_dir, "image_mini.pt"))[f"img{img_num}"]
    col = torch.load(os.path.join(golden_dir, "collate_mini.pt"))["batch"]
    cfg = indoor_config(first_feats_dim=32, gnn_feats_dim=64, image_feature=True, img_num=img_num, in_feats_dim=129)
    torch.manual_seed(gold["seed"])
    np.random.seed(gold["seed"])
    net = KPFCNN(cfg).to(cuda)
    net = net.train() if train else net.eval()
    n_src = int(col["stack_lengths"][0][0])
    batch = {k: _to(v, cuda) for k, v in col.items()}
    batch["src_pcd_raw"], batch["tgt_pcd_raw"] = batch["points"][0][:n_src], batch["points"][0][n_src:]
    for k, v in gold["inputs"].items():
        batch[k] = v.to(cuda)
    return gold, cfg, net, batch


def test_runners_take_the_input_in_rows_of_132_floats(cuda, golden_dir):
    """Round 6: the C++ runners are handed the [N, 129] input in rows of KPFCNN.IMAGE_WIDTH = 132 floats (three zero columns)
    against zero-padded first-layer weights -- no pad copy per forward.  The padded matrix carries the reference's 129 columns
    bit for bit, zeros beside them, and the forward's outputs are what the [N, 129] form gives."""
    gold, cfg, net, batch = _mini_image_case(cuda, golden_dir)
    x129 = net.image_features(batch)
    x132 = net.image_features(batch, width=net.IMAGE_WIDTH)
    assert x132.shape == (x129.shape[0], 132) and torch.equal(x132[:, :129], x129) and not x132[:, 129:].any()
    with torch.no_grad():
        out = net(batch)                                                   # the runner, fed the 132-wide rows
        out129 = net.runner().forward({**batch, "features": x129})         # the runner's own pad copy of [N, 129]
    for k, want in gold["outputs"].items():
        assert MR.rel_err(out[k].cpu(), want) < 1e-4, k
        assert MR.rel_err(out[k].cpu(), out129[k].cpu()) < 1e-5, k      # (equal up to the order of the products' fp32 atomics)


def test_pair_engine_carries_image_features(cuda, golden_dir):
    """PairStreams.submit(points, lengths, images=...): PCR-CG's shipped configuration through the engine -- pyramid build,
    injection on the model stream, grouped forward -- equals the sequential forward of the same pairs; and the engine refuses
    a pair without images."""
    from pcrcg_amd import synthetic
    from pcrcg_amd.pairstream import PairStreams
    from pcrcg_amd.pyramid import build_pyramid
    cfg = indoor_config(first_feats_dim=32, gnn_feats_dim=64, image_feature=True, img_num=2, in_feats_dim=129)
    torch.manual_seed(3)
    np.random.seed(3)
    net = KPFCNN(cfg).to(cuda).eval()
    limits = [20, 26, 30, 32]
    cases, ref = [], []
    for seed, recipe in enumerate(("mini", "C1", "mini", "mini", "C1")):
        src, tgt = synthetic.pair(recipe, seed)
        pts = torch.from_numpy(np.concatenate([src, tgt])).to(cuda)
        lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=cuda)
        im = {k: torch.from_numpy(v).to(cuda) for k, v in synthetic.image_inputs(len(src), len(tgt), seed, img_num=2, h=12, w=16).items()}
        batch = build_pyramid(pts, lens, cfg, limits)
        batch.update(im)
        batch["src_pcd_raw"], batch["tgt_pcd_raw"] = pts[:len(src)], pts[len(src):]
        _, _, images = net.image_list(batch)
        cases.append((pts, lens, images))
        with torch.no_grad():
            ref.append(net(batch))
    for per_build, per_fwd in ((1, 1), (4, 4), (3, 2)):
        eng = PairStreams(net, cfg, limits, cuda, model_streams=2, pairs_per_build=per_build, pairs_per_forward=per_fwd)
        for pts, lens, images in cases * 2:
            eng.submit(pts, lens, images=images)
        outs = [eng.result() for _ in range(2 * len(cases))]
        eng.drain()
        for i, o in enumerate(outs):
            for k in ("feats_f", "scores_overlap", "scores_saliency"):
                r = ref[i % len(cases)][k]
                assert float((o[k] - r).abs().max()) <= 1e-5 * float(r.abs().max()), (per_build, i, k)
        eng.submit(cases[0][0], cases[0][1])                # no images for a network that takes them
        with pytest.raises(RuntimeError, match="image"):
            eng.result()
        eng.close()


def test_train_runner_covers_the_129_channel_input(cuda, golden_dir):
    """The C++ train-step runner on PCR-CG's shipped configuration (round 5 sent it down the op-by-op path): same forward
    values, and every parameter's gradient equals the op-by-op autograd composition's -- the first KPConv's [15, 129, cout]
    gradient included, which the runner accumulates in the padded [15, 132, cout] layout and folds back."""
    from pcrcg_amd.train_forward import forward_train
    gold, cfg, net, batch = _mini_image_case(cuda, golden_dir, train=True)
    assert net.train_runner() is not None
    n = batch["points"][0].shape[0]
    g = torch.Generator().manual_seed(1)
    r1, r2, r3 = (torch.randn(n, 32, generator=g).to(cuda), torch.randn(n, generator=g).to(cuda), torch.randn(n, generator=g).to(cuda))

    def scalar(out):
        return (out["feats_f"] * r1).sum() + (out["scores_overlap"] * r2).sum() + (out["scores_saliency"] * r3).sum()
    out = net(batch)                                     # training mode: KPFCNN.forward -> train runner, 132-wide input
    assert out["feats_f"].grad_fn is not None
    for k, want in gold["outputs"].items():
        assert MR.rel_err(out[k].detach().cpu(), want) < 1e-4, k
    scalar(out).backward()
    got = {name: p.grad.clone() for name, p in net.named_parameters() if p.requires_grad}
    net.zero_grad(set_to_none=True)
    out2 = forward_train(net, {**batch, "features": net.image_features(batch)})      # the op-by-op mirror on [N, 129]
    scalar(out2).backward()
    floor = 1e-4 * max(float(p.grad.abs().max()) for p in net.parameters() if p.grad is not None)
    worst = {}
    for name, p in net.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None and name in got, name
        worst[name] = float((got[name] - p.grad).abs().max() / max(float(p.grad.abs().max()), floor))
    assert got["encoder_blocks.0.KPConv.weights"].shape == (15, 129, 16)
    bad = {k: v for k, v in worst.items() if v > 1e-3}
    assert not bad, bad
