"""GNN / cross-attention overlap head on the MI355X -- host-side mirror of ref:models/gcn.py with the
same module tree and parameter names (``layers.{i}.conv{1,2,3}.weight``, ``layers.{i}.attn.*``,
``layers.{i}.mlp.*``), working on row-major [N, C] feature matrices.

Differences in HOW (not WHAT):
  * get_graph_feature (:37-64) never builds the [1, C, N, N] tensor nor the [1, 2C, N, k] edge tensor:
    the 1x1 conv over cat(f_i, f_j - f_i) is split into a centre and a neighbour GEMM on [N, C] and the
    edge max / InstanceNorm2d statistics are reduced on the fly (pcrcg_edgeconv_reduce);
  * attention heads are made contiguous by permuting the projection weights once per call (the
    reference's `.view(B, dim, heads, N)` interleaves heads over channels, :170)."""
from copy import deepcopy

import torch
import torch.nn as nn

from . import ops


def _w2d(conv):
    return conv.weight.data.flatten(1)  # [Cout, Cin] of a 1x1 conv


class SelfAttention(nn.Module):
    """ref:models/gcn.py:96-134 (DGCNN edge-conv x2 + fuse)."""

    def __init__(self, feature_dim, k=10):
        super().__init__()
        self.conv1 = nn.Conv2d(feature_dim * 2, feature_dim, kernel_size=1, bias=False)
        self.in1 = nn.InstanceNorm2d(feature_dim)
        self.conv2 = nn.Conv2d(feature_dim * 2, feature_dim * 2, kernel_size=1, bias=False)
        self.in2 = nn.InstanceNorm2d(feature_dim * 2)
        self.conv3 = nn.Conv2d(feature_dim * 4, feature_dim, kernel_size=1, bias=False)
        self.in3 = nn.InstanceNorm2d(feature_dim)
        self.k = k

    @staticmethod
    def _edge_layer(feats, idx, conv, out):
        """feats [N, Cin] -> out [N, Cout] = max_k LeakyReLU(IN2d(conv(cat(f_i, f_j - f_i))), 0.2)."""
        w = _w2d(conv)                                   # [Cout, 2*Cin]
        cin = feats.shape[1]
        wa, wb = w[:, :cin], w[:, cin:]
        both = torch.cat([(wa - wb).t(), wb.t()], 1).contiguous()     # [Cin, 2*Cout]: centre | neighbour
        cn = ops.gemm(feats, both)                       # [N, 2*Cout]
        cout = w.shape[0]
        emax, stats = ops.edgeconv_reduce(cn[:, :cout], cn[:, cout:], idx)
        return ops.instnorm_apply(emax, stats, 0.2, out=out)

    def forward(self, coords, features):
        """coords [N, 3], features [N, C] -> [N, C]."""
        n, c = features.shape
        k = min(self.k, n - 1)
        idx = ops.knn(coords, k)                         # :48-51
        cat = torch.empty((n, 4 * c), dtype=torch.float32, device=features.device)
        cat[:, :c].copy_(features)                       # x0
        self._edge_layer(features, idx, self.conv1, cat[:, c:2 * c])          # x1  :121-125
        self._edge_layer(cat[:, c:2 * c], idx, self.conv2, cat[:, 2 * c:])    # x2  :127-129
        x3 = ops.gemm(cat, _w2d(self.conv3).t())         # :131-132
        return ops.instnorm_lrelu(x3, 0.2)


def MLP(channels, do_bn=True):
    """ref:models/gcn.py:137-148 (module tree only; AttentionalPropagation.forward runs it on HIP)."""
    n = len(channels)
    layers = []
    for i in range(1, n):
        layers.append(nn.Conv1d(channels[i - 1], channels[i], kernel_size=1, bias=True))
        if i < (n - 1):
            if do_bn:
                layers.append(nn.InstanceNorm1d(channels[i]))
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class MultiHeadedAttention(nn.Module):
    """ref:models/gcn.py:158-173."""

    def __init__(self, num_heads, d_model):
        super().__init__()
        assert d_model % num_heads == 0
        self.dim = d_model // num_heads
        self.num_heads = num_heads
        self.merge = nn.Conv1d(d_model, d_model, kernel_size=1)
        self.proj = nn.ModuleList([deepcopy(self.merge) for _ in range(3)])

    def forward(self, query, key, value):
        """query [N, C]; key, value [M, C] -> [N, C]."""
        h, d = self.num_heads, self.dim
        c = h * d
        dev = query.device
        # channel d*h + head  ->  head-major position head*dim + d
        perm = (torch.arange(h, device=dev)[:, None] + h * torch.arange(d, device=dev)[None, :]).reshape(-1)
        proj = []
        for layer, x in zip(self.proj, (query, key, value)):
            w = layer.weight.data.squeeze(-1)[perm]      # rows = output channels, head-major
            proj.append(ops.gemm(x, w.t(), bias=layer.bias.data[perm].contiguous()))
        q, kk, v = proj
        n, m = q.shape[0], kk.shape[0]
        msg = torch.empty((n, c), dtype=torch.float32, device=dev)
        scores = torch.empty((n, m), dtype=torch.float32, device=dev)
        for i in range(h):
            ops.gemm(q[:, i * d:(i + 1) * d], kk[:, i * d:(i + 1) * d].t(), out=scores)   # :152
            ops.softmax_rows_(scores, 1.0 / d ** 0.5)                                 # :153
            ops.gemm(scores, v[:, i * d:(i + 1) * d], out=msg[:, i * d:(i + 1) * d])  # :154
        wm = self.merge.weight.data.squeeze(-1)[:, perm]
        return ops.gemm(msg, wm.t(), bias=self.merge.bias.data)                       # :173


class AttentionalPropagation(nn.Module):
    """ref:models/gcn.py:176-185."""

    def __init__(self, feature_dim, num_heads):
        super().__init__()
        self.attn = MultiHeadedAttention(num_heads, feature_dim)
        self.mlp = MLP([feature_dim * 2, feature_dim * 2, feature_dim])
        nn.init.constant_(self.mlp[-1].bias, 0.0)

    def forward(self, x, source):
        message = self.attn(x, source, source)
        y = torch.cat([x, message], 1)
        l0, l3 = self.mlp[0], self.mlp[3]
        y = ops.gemm(y, l0.weight.data.squeeze(-1).t(), bias=l0.bias.data)
        y = ops.instnorm_lrelu(y, 0.0)                   # InstanceNorm1d + ReLU  (:146-147)
        return ops.gemm(y, l3.weight.data.squeeze(-1).t(), bias=l3.bias.data)


class GCN(nn.Module):
    """ref:models/gcn.py:188-217.  coords [N, 3], descriptors [N, C] (row-major)."""

    def __init__(self, num_head, feature_dim, k, layer_names):
        super().__init__()
        layers = []
        for atten_type in layer_names:
            if atten_type == "cross":
                layers.append(AttentionalPropagation(feature_dim, num_head))
            elif atten_type == "self":
                layers.append(SelfAttention(feature_dim, k))
        self.layers = nn.ModuleList(layers)
        self.names = layer_names

    def forward(self, coords0, coords1, desc0, desc1):
        for layer, name in zip(self.layers, self.names):
            if name == "cross":
                desc0 = desc0 + layer(desc0, desc1)      # :213
                desc1 = desc1 + layer(desc1, desc0)      # :214 (sees the updated desc0)
            elif name == "self":
                desc0 = layer(coords0, desc0)
                desc1 = layer(coords1, desc1)
        return desc0, desc1
