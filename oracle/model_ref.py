"""oracle/model_ref.py -- TEST INFRASTRUCTURE ONLY (CPU oracle), not product code.

A functional, plain-PyTorch fp32 restatement of the reference's floating-point model path:
KPConv / blocks (ref:models/blocks.py), the GNN head (ref:models/gcn.py) and the KPFCNN wiring
(ref:models/architectures.py:37-174, 181-191, 516-610, geometry-only branch).  It works directly on
a reference ``state_dict`` (same key names and shapes) and on the reference's batch dict
(ref:datasets/dataloader.py:363-380), so that outputs of the unmodified reference, of this oracle
and of the HIP path can be compared tensor by tensor.

Parity status: PINNED against the imported reference model in the build container
(scripts/make_golden_model.py; fixtures tests/golden/model_*.pt; tests/test_oracle_model.py).
The tolerance for this floating-point path is the one BASELINE.json states: 1e-4 relative
(`max|a-b| <= 1e-4 * max|ref|` per tensor).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# architecture plan (ref:models/architectures.py:37-174, ref:configs/models.py)
# ------------------------------------------------------------------------------------------------
def plan(cfg):
    """Walk the block list exactly as KPFCNN.__init__ does and return per-block descriptors."""
    arch = list(cfg["architecture"])
    layer = 0
    r = cfg["first_subsampling_dl"] * cfg["conv_radius"]
    in_dim = cfg["in_feats_dim"]
    out_dim = cfg["first_feats_dim"]
    enc, skips, skip_dims = [], [], []
    for i, blk in enumerate(arch):
        if any(t in blk for t in ("pool", "strided", "upsample", "global")):
            skips.append(i)
            skip_dims.append(in_dim)
        if "upsample" in blk:
            break
        enc.append(dict(name=blk, radius=r, extent=r * cfg["KP_extent"] / cfg["conv_radius"],
                        in_dim=in_dim, out_dim=out_dim, layer=layer, strided="strided" in blk))
        in_dim = out_dim // 2 if "simple" in blk else out_dim
        if "pool" in blk or "strided" in blk:
            layer += 1
            r *= 2
            out_dim *= 2
    enc_out_dim = in_dim
    gnn_dim = cfg["gnn_feats_dim"]
    out_dim = gnn_dim + 2
    in_dim = out_dim  # after the GNN the decoder input is [score, saliency, feats]
    start = next(i for i, b in enumerate(arch) if "upsample" in b)
    dec, concats = [], []
    for j, blk in enumerate(arch[start:]):
        if j > 0 and "upsample" in arch[start + j - 1]:
            in_dim += skip_dims[layer]
            concats.append(j)
        dec.append(dict(name=blk, in_dim=in_dim, out_dim=out_dim, layer=layer))
        in_dim = out_dim
        if "upsample" in blk:
            layer -= 1
            r *= 0.5
            out_dim = out_dim // 2
    return dict(encoder=enc, encoder_skips=skips, skip_dims=skip_dims, enc_out_dim=enc_out_dim,
                decoder=dec, decoder_concats=concats)


# ------------------------------------------------------------------------------------------------
# point ops (ref:models/blocks.py)
# ------------------------------------------------------------------------------------------------
def kpconv(q_pts, s_pts, inds, x, kernel_points, weights, extent, chunk=4096):
    """Rigid KPConv, linear influence, sum aggregation (ref:models/blocks.py:229-374).

    out[q] = (1/n_q) * sum_k ( sum_h max(0, 1 - |s_h - q - kp_k| / extent) * x[h] ) @ W_k,
    n_q = max(1, #{h : sum_c x[h, c] > 0}); shadow index Ns -> point (1e6,1e6,1e6), feature 0."""
    s_pad = torch.cat([s_pts, torch.full_like(s_pts[:1], 1e6)], 0)        # :269
    x_pad = torch.cat([x, torch.zeros_like(x[:1])], 0)                    # :348
    outs = []
    for a in range(0, inds.shape[0], chunk):
        idx = inds[a:a + chunk]
        nb = s_pad[idx] - q_pts[a:a + chunk].unsqueeze(1)                 # :272-275  [n,H,3]
        diff = nb.unsqueeze(2) - kernel_points                            # :285-286  [n,H,K,3]
        d2 = (diff ** 2).sum(3)                                           # :289
        w = torch.clamp(1 - torch.sqrt(d2) / extent, min=0.0)             # :328      [n,H,K]
        nx = x_pad[idx]                                                   # :351      [n,H,Cin]
        wf = torch.matmul(w.transpose(1, 2), nx)                          # :354      [n,K,Cin]
        out = torch.matmul(wf.permute(1, 0, 2), weights).sum(0)           # :360-366  [n,Cout]
        n = (nx.sum(-1) > 0).sum(-1).clamp(min=1)                         # :369-371
        outs.append(out / n.unsqueeze(1))                                 # :372
    return torch.cat(outs, 0)


def max_pool(x, inds):
    """ref:models/blocks.py:86-102 (shadow row is zero, so it takes part in the max)."""
    x_pad = torch.cat([x, torch.zeros_like(x[:1])], 0)
    return x_pad[inds].max(1)[0]


def closest_pool(x, inds):
    """ref:models/blocks.py:71-83."""
    x_pad = torch.cat([x, torch.zeros_like(x[:1])], 0)
    return x_pad[inds[:, 0]]


def instance_norm_rows(x, eps=1e-5):
    """BatchNormBlock with use_bn=True is nn.InstanceNorm1d over ALL stacked points
    (ref:models/blocks.py:448,456-463): per channel, biased variance, eps 1e-5, no affine."""
    mean = x.mean(0, keepdim=True)
    var = x.var(0, unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


def _norm(sd, prefix, x, use_bn):
    return instance_norm_rows(x) if use_bn else x + sd[prefix + ".bias"]


def unary(sd, prefix, x, use_bn, relu=True):
    """UnaryBlock (ref:models/blocks.py:473-501)."""
    x = x @ sd[prefix + ".mlp.weight"].t()
    x = _norm(sd, prefix + ".batch_norm", x, use_bn)
    return F.leaky_relu(x, 0.1) if relu else x


def _geometry(batch, blk):
    l = blk["layer"]
    if blk["strided"]:
        return batch["points"][l + 1], batch["points"][l], batch["pools"][l]
    return batch["points"][l], batch["points"][l], batch["neighbors"][l]


def simple_block(sd, prefix, blk, x, batch, use_bn):
    """ref:models/blocks.py:536-590."""
    q, s, inds = _geometry(batch, blk)
    x = kpconv(q, s, inds, x, sd[prefix + ".KPConv.kernel_points"], sd[prefix + ".KPConv.weights"],
               blk["extent"])
    return F.leaky_relu(_norm(sd, prefix + ".batch_norm", x, use_bn), 0.1)


def resnetb_block(sd, prefix, blk, feats, batch, use_bn):
    """ref:models/blocks.py:593-678."""
    q, s, inds = _geometry(batch, blk)
    x = feats
    if prefix + ".unary1.mlp.weight" in sd:
        x = unary(sd, prefix + ".unary1", x, use_bn)
    x = kpconv(q, s, inds, x, sd[prefix + ".KPConv.kernel_points"], sd[prefix + ".KPConv.weights"],
               blk["extent"])
    x = F.leaky_relu(_norm(sd, prefix + ".batch_norm_conv", x, use_bn), 0.1)
    x = unary(sd, prefix + ".unary2", x, use_bn, relu=False)
    sc = max_pool(feats, inds) if blk["strided"] else feats
    if prefix + ".unary_shortcut.mlp.weight" in sd:
        sc = unary(sd, prefix + ".unary_shortcut", sc, use_bn, relu=False)
    return F.leaky_relu(x + sc, 0.1)


# ------------------------------------------------------------------------------------------------
# GNN head (ref:models/gcn.py)
# ------------------------------------------------------------------------------------------------
def knn_indices(coords, k):
    """ref:models/gcn.py:15-34,48-51: dist = -2ab + a^2 + b^2 clamped at 1e-12, k+1 smallest, drop
    the first.  coords [N,3] -> [N,k]."""
    d = -2 * coords @ coords.t()
    d = d + (coords ** 2).sum(-1)[:, None]
    d = d + (coords ** 2).sum(-1)[None, :]
    d = torch.clamp(d, min=1e-12)
    return d.topk(k + 1, dim=-1, largest=False, sorted=True)[1][:, 1:]


def _inorm(x, dims, eps=1e-5):
    mean = x.mean(dims, keepdim=True)
    var = x.var(dims, unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


def _edge_conv(feats, idx, w):
    """get_graph_feature + 1x1 conv + InstanceNorm2d + LeakyReLU(0.2) + max over k
    (ref:models/gcn.py:37-64,123-129).  feats [N,C], idx [N,k], w [Cout,2C] -> [N,Cout]."""
    nb = feats[idx]                                        # [N,k,C]
    ctr = feats.unsqueeze(1).expand_as(nb)
    e = torch.cat([ctr, nb - ctr], -1) @ w.t()             # [N,k,Cout]
    e = F.leaky_relu(_inorm(e, (0, 1)), 0.2)
    return e.max(1)[0]


def self_attention(sd, prefix, coords, feats, k):
    """SelfAttention.forward (ref:models/gcn.py:110-134).  coords [N,3], feats [N,C] -> [N,C]."""
    k = min(k, coords.shape[0] - 1)
    idx = knn_indices(coords, k)
    x0 = feats
    x1 = _edge_conv(x0, idx, sd[prefix + ".conv1.weight"].flatten(1))
    x2 = _edge_conv(x1, idx, sd[prefix + ".conv2.weight"].flatten(1))
    x3 = torch.cat([x0, x1, x2], 1) @ sd[prefix + ".conv3.weight"].flatten(1).t()
    return F.leaky_relu(_inorm(x3, (0,)), 0.2)


def _conv1d(sd, prefix, x):
    return x @ sd[prefix + ".weight"].squeeze(-1).t() + sd[prefix + ".bias"]


def cross_attention(sd, prefix, x, src, heads):
    """AttentionalPropagation.forward (ref:models/gcn.py:151-185).  x [N,C], src [M,C] -> [N,C].
    The reference views the projected [B, C, N] tensor as [B, dim, heads, N] (:170), i.e. channel
    c belongs to head c % heads, and merges back with the same interleaving (:173)."""
    n, c = x.shape
    dim = c // heads
    q = _conv1d(sd, prefix + ".attn.proj.0", x).view(n, dim, heads)
    kk = _conv1d(sd, prefix + ".attn.proj.1", src).view(-1, dim, heads)
    v = _conv1d(sd, prefix + ".attn.proj.2", src).view(-1, dim, heads)
    scores = torch.einsum("ndh,mdh->hnm", q, kk) / dim ** 0.5
    prob = torch.softmax(scores, -1)
    msg = torch.einsum("hnm,mdh->ndh", prob, v).reshape(n, c)
    msg = _conv1d(sd, prefix + ".attn.merge", msg)
    y = _conv1d(sd, prefix + ".mlp.0", torch.cat([x, msg], 1))
    y = F.relu(_inorm(y, (0,)))                                       # MLP: IN1d + ReLU (:137-148)
    return _conv1d(sd, prefix + ".mlp.3", y)


def gcn(sd, prefix, names, c0, c1, d0, d1, k, heads):
    """GCN.forward (ref:models/gcn.py:208-217).  desc1's cross update sees the UPDATED desc0."""
    for i, name in enumerate(names):
        p = f"{prefix}.layers.{i}"
        if name == "cross":
            d0 = d0 + cross_attention(sd, p, d0, d1, heads)
            d1 = d1 + cross_attention(sd, p, d1, d0, heads)
        else:
            d0 = self_attention(sd, p, c0, d0, k)
            d1 = self_attention(sd, p, c1, d1, k)
    return d0, d1


def inject_image_features(n_points, len_src, images, channels=128):
    """ref:models/architectures.py:195-514 (all three img_num branches reduce to this): x = ones [N, C+1]; for every
    image in the reference's write order  x[inds3d (+ len_src for the target cloud), :] =
    cat(fmap[:, inds2d[:,1], inds2d[:,0]].T, ones)  with fmap pre-multiplied by valid.T when a valid map exists."""
    x = torch.ones(n_points, channels + 1)
    for im in images:
        fmap = im["fmap"]
        if im.get("valid") is not None:
            fmap = fmap * im["valid"].transpose(0, 1).unsqueeze(0)                      # :272-284
        feats = fmap[:, im["inds2d"][:, 1], im["inds2d"][:, 0]]                         # :227-232
        rows = im["inds3d"] + (len_src if im.get("target") else 0)                      # :239-241
        x[rows, :] = torch.cat((feats.transpose(1, 0), torch.ones(feats.shape[1], 1)), -1)
    return x


# ------------------------------------------------------------------------------------------------
# KPFCNN.forward, geometry-only branch (ref:models/architectures.py:181-191, 516-610)
# ------------------------------------------------------------------------------------------------
def kpfcnn_forward_with_grad(sd, cfg, batch, return_intermediates=False):
    """The forward below with autograd recording (reference gradients for the training-row tests)."""
    return _kpfcnn_forward(sd, cfg, batch, return_intermediates)


@torch.no_grad()
def kpfcnn_forward(sd, cfg, batch, return_intermediates=False):
    return _kpfcnn_forward(sd, cfg, batch, return_intermediates)


def _kpfcnn_forward(sd, cfg, batch, return_intermediates=False):
    pl = plan(cfg)
    use_bn = cfg.get("use_batch_norm", True)
    x = batch["features"].clone()
    len_src_c = int(batch["stack_lengths"][-1][0])
    pcd_c = batch["points"][-1]
    inter = {}
    skip_x = []
    for i, blk in enumerate(pl["encoder"]):
        if i in pl["encoder_skips"]:
            skip_x.append(x)
        prefix = f"encoder_blocks.{i}"
        if "simple" in blk["name"]:
            x = simple_block(sd, prefix, blk, x, batch, use_bn)
        else:
            x = resnetb_block(sd, prefix, blk, x, batch, use_bn)
        inter[f"enc{i}"] = x
    feats_c = _conv1d(sd, "bottle", x)                                       # :527-528
    inter["bottle"] = feats_c
    s_f, t_f = gcn(sd, "gnn", cfg["nets"], pcd_c[:len_src_c], pcd_c[len_src_c:], feats_c[:len_src_c],
                   feats_c[len_src_c:], cfg["dgcnn_k"], cfg["num_head"])     # :532-535
    feats_c = torch.cat([s_f, t_f], 0)
    inter["gnn"] = feats_c
    feats_c = _conv1d(sd, "proj_gnn", feats_c)                               # :538
    scores_c = _conv1d(sd, "proj_score", feats_c)                            # :539
    fn = F.normalize(feats_c, p=2, dim=1)                                    # :541
    inner = fn[:len_src_c] @ fn[len_src_c:].t()                              # :556-557
    temp = torch.exp(sd["epsilon"]) + 0.03                                   # :561
    s1 = torch.softmax(inner / temp, 1) @ scores_c[len_src_c:]
    s2 = torch.softmax(inner.t() / temp, 1) @ scores_c[:len_src_c]
    x = torch.cat([scores_c, torch.cat([s1, s2], 0), feats_c], 1)            # :565
    inter["coarse"] = x
    for j, blk in enumerate(pl["decoder"]):
        if j in pl["decoder_concats"]:
            x = torch.cat([x, skip_x.pop()], 1)
        prefix = f"decoder_blocks.{j}"
        if "upsample" in blk["name"]:
            x = closest_pool(x, batch["upsamples"][blk["layer"] - 1])        # blocks.py:704-705
        elif blk["name"] == "last_unary":
            x = x @ sd[prefix + ".mlp.weight"].t()
        else:
            x = unary(sd, prefix, x, use_bn)
    fd = cfg["final_feats_dim"]
    feats_f = x[:, :fd]
    so = torch.clamp(torch.sigmoid(x[:, fd]), 0, 1)                          # :576-577
    ss = torch.clamp(torch.sigmoid(x[:, fd + 1]), 0, 1)
    so = torch.nan_to_num(so, nan=0.0, posinf=0.0, neginf=0.0)               # regular_score :176-179
    ss = torch.nan_to_num(ss, nan=0.0, posinf=0.0, neginf=0.0)
    out = {"feats_f": F.normalize(feats_f, p=2, dim=1), "scores_overlap": so, "scores_saliency": ss}
    if return_intermediates:
        out["_inter"] = inter
    return out


def rel_err(a, b):
    """Parity metric for this path: max|a-b| / max|ref|."""
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
