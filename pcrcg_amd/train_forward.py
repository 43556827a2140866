"""Differentiable forward of KPFCNN: the same module tree and parameters as
pcrcg_amd.architectures.KPFCNN, composed from the autograd-aware HIP ops of pcrcg_amd.autograd so that
`loss.backward()` reaches every parameter.  Mirrors ref:models/architectures.py:516-610 and the block
forwards of ref:models/blocks.py / ref:models/gcn.py line by line; only cheap element-wise glue
(concatenation, residual add + LeakyReLU, sigmoid, L2 normalisation, the temperature division) is left to
torch -- every gather, scatter, normalisation, GEMM and attention kernel, forward and backward, is HIP.

Forward in train() mode equals eval() mode (no dropout, InstanceNorm without running statistics: SURVEY
appendix C), so the values agree with the inference runner; tests/test_train_step_gpu.py checks both that
and the gradients against torch autograd of the CPU oracle."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import autograd as AG
from . import ops
from .blocks import (LastUnaryBlock, NearestUpsampleBlock, ResnetBottleneckBlock, SimpleBlock, UnaryBlock, _geometry)
from .gcn import AttentionalPropagation, SelfAttention


def _unary(block, x):
    """UnaryBlock (ref:models/blocks.py:473-507): Linear (no bias) -> InstanceNorm -> LeakyReLU(0.1) unless no_relu."""
    y = AG.linear(x, block.mlp.weight)
    if block.use_bn:
        return AG.instnorm_lrelu(y, 1.0 if block.no_relu else 0.1)
    y = y + block.batch_norm.bias
    return y if block.no_relu else F.leaky_relu(y, 0.1)


def _norm(bn_block, x, slope):
    if bn_block.use_bn:
        return AG.instnorm_lrelu(x, slope)
    y = x + bn_block.bias
    return y if slope == 1.0 else F.leaky_relu(y, slope)


def _kpconv(mod, q_pts, s_pts, inds, x):
    return AG.kpconv(x, mod.weights, q_pts, s_pts, inds, mod.kernel_points.data, mod.KP_extent)


def _block(block, x, batch):
    if isinstance(block, SimpleBlock):                                   # ref:models/blocks.py:579-590
        q_pts, s_pts, inds = _geometry(block.block_name, block.layer_ind, batch)
        return _norm(block.batch_norm, _kpconv(block.KPConv, q_pts, s_pts, inds, x), 0.1)
    if isinstance(block, ResnetBottleneckBlock):                         # :650-678
        q_pts, s_pts, inds = _geometry(block.block_name, block.layer_ind, batch)
        y = x if isinstance(block.unary1, nn.Identity) else _unary(block.unary1, x)
        y = _norm(block.batch_norm_conv, _kpconv(block.KPConv, q_pts, s_pts, inds, y), 0.1)
        y = _unary(block.unary2, y)
        shortcut = AG.max_pool(x, inds) if "strided" in block.block_name else x
        if not isinstance(block.unary_shortcut, nn.Identity):
            shortcut = _unary(block.unary_shortcut, shortcut)
        return F.leaky_relu(y + shortcut, 0.1)
    if isinstance(block, UnaryBlock):
        return _unary(block, x)
    if isinstance(block, LastUnaryBlock):                                # :527-529
        return AG.linear(x, block.mlp.weight)
    if isinstance(block, NearestUpsampleBlock):                          # :704-705
        return AG.closest_pool(x, batch["upsamples"][block.layer_ind - 1])
    raise NotImplementedError(f"pcrcg_amd.train_forward: block {type(block).__name__}")


def _conv1x1(layer, x):
    return AG.linear(x, layer.weight.squeeze(-1), layer.bias)


def _edge_layer(feats, idx, conv):
    """max_k LeakyReLU(IN2d(conv(cat(f_i, f_j - f_i))), 0.2) with the 1x1 conv split into a centre and a
    neighbour product: W [f_i ; f_j - f_i] = (Wa - Wb) f_i + Wb f_j   (ref:models/gcn.py:37-64,121-129)."""
    w = conv.weight.flatten(1)                       # [Cout, 2*Cin]
    cin = feats.shape[1]
    wa, wb = w[:, :cin], w[:, cin:]
    ctr = AG.linear(feats, wa - wb)
    nbr = AG.linear(feats, wb)
    return AG.edge_conv(ctr, nbr, idx, 0.2)


def _self_attention(layer, coords, feats):
    """ref:models/gcn.py:109-134."""
    n = feats.shape[0]
    idx = ops.knn(coords.contiguous(), min(layer.k, n - 1))
    x1 = _edge_layer(feats, idx, layer.conv1)
    x2 = _edge_layer(x1, idx, layer.conv2)
    x3 = AG.linear(torch.cat([feats, x1, x2], 1), layer.conv3.weight.flatten(1))
    return AG.instnorm_lrelu(x3, 0.2)


def _attention(att, query, key, value):
    """MultiHeadedAttention (ref:models/gcn.py:151-173); channel c of a projection belongs to head c % heads."""
    h, d = att.num_heads, att.dim
    q, k, v = [_conv1x1(layer, x) for layer, x in zip(att.proj, (query, key, value))]
    msgs = []
    for i in range(h):
        qi, ki, vi = q[:, i::h], k[:, i::h], v[:, i::h]                 # [N, d] strided column views
        prob = AG.softmax_rows(AG.matmul(qi.contiguous(), ki.contiguous().t()), 1.0 / d ** 0.5)
        msgs.append(AG.matmul(prob, vi.contiguous()))                    # [N, d]
    msg = torch.stack(msgs, 2).reshape(q.shape[0], h * d)               # channel = dim * heads + head
    return _conv1x1(att.merge, msg)


def _cross_attention(layer, x, source):
    """AttentionalPropagation (ref:models/gcn.py:176-185)."""
    message = _attention(layer.attn, x, source, source)
    y = _conv1x1(layer.mlp[0], torch.cat([x, message], 1))
    y = AG.instnorm_lrelu(y, 0.0)                                        # InstanceNorm1d + ReLU
    return _conv1x1(layer.mlp[3], y)


def _gnn(gnn, coords0, coords1, desc0, desc1):
    """ref:models/gcn.py:208-217."""
    for layer, name in zip(gnn.layers, gnn.names):
        if isinstance(layer, AttentionalPropagation):
            desc0 = desc0 + _cross_attention(layer, desc0, desc1)
            desc1 = desc1 + _cross_attention(layer, desc1, desc0)
        elif isinstance(layer, SelfAttention):
            desc0 = _self_attention(layer, coords0, desc0)
            desc1 = _self_attention(layer, coords1, desc1)
    return desc0, desc1


def forward_train(net, batch):
    """KPFCNN.forward with gradients (ref:models/architectures.py:181-191,516-610)."""
    x = batch["features"].clone().detach()
    if "stack_lengths_host" in batch:
        len_src_c = int(batch["stack_lengths_host"][-1][0])
    else:
        len_src_c = int(batch["stack_lengths"][-1][0])
    pcd_c = batch["points"][-1]
    src_pcd_c, tgt_pcd_c = pcd_c[:len_src_c], pcd_c[len_src_c:]

    skip_x = []
    for block_i, block_op in enumerate(net.encoder_blocks):
        if block_i in net.encoder_skips:
            skip_x.append(x)
        x = _block(block_op, x, batch)

    feats_c = _conv1x1(net.bottle, x)
    src_f, tgt_f = _gnn(net.gnn, src_pcd_c, tgt_pcd_c, feats_c[:len_src_c], feats_c[len_src_c:])
    feats_c = _conv1x1(net.proj_gnn, torch.cat([src_f, tgt_f], 0))
    scores_c = _conv1x1(net.proj_score, feats_c)                        # [N, 1]
    feats_norm = F.normalize(feats_c, p=2, dim=1)

    src_n, tgt_n = feats_norm[:len_src_c], feats_norm[len_src_c:]
    inner = AG.matmul(src_n.contiguous(), tgt_n.contiguous().t())
    temperature = torch.exp(net.epsilon) + 0.03
    p_st = AG.softmax_rows(inner / temperature, 1.0)
    p_ts = AG.softmax_rows(inner.t() / temperature, 1.0)
    s1 = AG.matmul(p_st, scores_c[len_src_c:].contiguous())
    s2 = AG.matmul(p_ts, scores_c[:len_src_c].contiguous())
    x = torch.cat([scores_c, torch.cat([s1, s2], 0), feats_c], 1)

    for block_i, block_op in enumerate(net.decoder_blocks):
        if block_i in net.decoder_concats:
            x = torch.cat([x, skip_x.pop()], 1)
        x = _block(block_op, x, batch)
    fd = net.final_feats_dim
    feats_f = F.normalize(x[:, :fd], p=2, dim=1)
    scores_overlap = net.regular_score(torch.clamp(torch.sigmoid(x[:, fd].view(-1)), min=0, max=1))
    scores_saliency = net.regular_score(torch.clamp(torch.sigmoid(x[:, fd + 1].view(-1)), min=0, max=1))
    return {"feats_f": feats_f, "scores_overlap": scores_overlap, "scores_saliency": scores_saliency}
