"""KPConv network blocks on the MI355X -- the host-side mirror of ref:models/blocks.py.

Same class names, constructor arguments, parameter names and shapes (so a reference ``state_dict``
loads unchanged); every forward runs in the hand-written HIP kernels behind ``pcrcg_amd.ops``.  The
module forwards here are the inference path (forward under no_grad); training goes through
``KPFCNN.forward`` with autograd enabled, i.e. pcrcg_amd/train_forward.py over the backward kernels of
include/pcrcg_train.h."""
import math

import torch
import torch.nn as nn
from torch.nn.init import kaiming_uniform_
from torch.nn.parameter import Parameter

from . import ops
from .kernel_points import load_kernels


def _no_autograd(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise RuntimeError("pcrcg_amd: these module forwards are the inference path; wrap the call in "
                           "torch.no_grad(), or train through KPFCNN.forward / pcrcg_amd.train_forward "
                           "(differentiable composition of the same kernels)")


_PADDED = {}


def aligned_weight(weight):
    """[out, in] weight whose rows start on 16-byte boundaries: a cached copy with the leading
    dimension padded to a multiple of 4 when `in` is not one (decoder widths 1538 and 769), so the GEMM
    can use float4 loads.  The returned tensor is a [out, in] view."""
    w = weight.data
    k = w.shape[1]
    if k % 4 == 0:
        return w
    key = (w.data_ptr(), weight._version, tuple(w.shape))
    hit = _PADDED.get(id(weight))
    if hit is None or hit[0] != key:
        buf = torch.zeros((w.shape[0], (k + 3) // 4 * 4), dtype=w.dtype, device=w.device)
        buf[:, :k].copy_(w)
        hit = (key, buf)
        _PADDED[id(weight)] = hit
    return hit[1][:, :k]


def gather(x, idx, method=2):
    """ref:models/blocks.py:27-58 (row gather); kept for API parity."""
    return x[idx]


def closest_pool(x, inds):
    """ref:models/blocks.py:71-83."""
    return ops.gather_first(x, inds)


def max_pool(x, inds):
    """ref:models/blocks.py:86-102."""
    return ops.gather_max(x, inds)


def global_average(x, batch_lengths):
    """ref:models/blocks.py:105-126."""
    out, i0 = [], 0
    for length in batch_lengths:
        length = int(length)
        out.append(torch.mean(x[i0:i0 + length], dim=0))
        i0 += length
    return torch.stack(out)


class KPConv(nn.Module):
    """ref:models/blocks.py:135-379 (rigid kernel, 'linear' influence, 'sum' aggregation)."""

    def __init__(self, kernel_size, p_dim, in_channels, out_channels, KP_extent, radius,
                 fixed_kernel_points="center", KP_influence="linear", aggregation_mode="sum", deformable=False,
                 modulated=False):
        super().__init__()
        if deformable or KP_influence != "linear" or aggregation_mode != "sum":
            raise NotImplementedError("pcrcg_amd.KPConv: only the rigid / linear / sum configuration of the "
                                      "shipped architectures is implemented")
        self.K = kernel_size
        self.p_dim = p_dim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.radius = radius
        self.KP_extent = KP_extent
        self.fixed_kernel_points = fixed_kernel_points
        self.KP_influence = KP_influence
        self.aggregation_mode = aggregation_mode
        self.deformable = deformable
        self.modulated = modulated
        self.weights = Parameter(torch.zeros((self.K, in_channels, out_channels), dtype=torch.float32),
                                 requires_grad=True)
        kaiming_uniform_(self.weights, a=math.sqrt(5))
        self.kernel_points = Parameter(
            torch.tensor(load_kernels(self.radius, self.K, dimension=self.p_dim, fixed=self.fixed_kernel_points),
                         dtype=torch.float32), requires_grad=False)

    def forward(self, q_pts, s_pts, neighb_inds, x):
        _no_autograd(x, self.weights)
        return ops.kpconv(q_pts, s_pts, neighb_inds, x, self.kernel_points.data, self.weights.data,
                          self.KP_extent)

    def __repr__(self):
        return "KPConv(radius: {:.2f}, extent: {:.2f}, in_feat: {:d}, out_feat: {:d})".format(
            self.radius, self.KP_extent, self.in_channels, self.out_channels)


class BatchNormBlock(nn.Module):
    """ref:models/blocks.py:433-470: InstanceNorm1d over all stacked points (no affine, no running
    stats) when use_bn, else a learned bias."""

    def __init__(self, in_dim, use_bn, bn_momentum):
        super().__init__()
        self.bn_momentum = bn_momentum
        self.use_bn = use_bn
        self.in_dim = in_dim
        if not self.use_bn:
            self.bias = Parameter(torch.zeros(in_dim, dtype=torch.float32), requires_grad=True)

    def forward(self, x, slope=1.0):
        """`slope` fuses the LeakyReLU that always follows (1.0 = none)."""
        _no_autograd(x)
        if self.use_bn:
            return ops.instnorm_lrelu(x, slope)
        y = x + self.bias
        return y if slope == 1.0 else torch.nn.functional.leaky_relu(y, slope)

    def __repr__(self):
        return "BatchNormBlock(in_feat: {:d}, momentum: {:.3f}, only_bias: {:s})".format(
            self.in_dim, self.bn_momentum, str(not self.use_bn))


class UnaryBlock(nn.Module):
    """ref:models/blocks.py:473-507."""

    def __init__(self, in_dim, out_dim, use_bn, bn_momentum, no_relu=False):
        super().__init__()
        self.bn_momentum = bn_momentum
        self.use_bn = use_bn
        self.no_relu = no_relu
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.mlp = nn.Linear(in_dim, out_dim, bias=False)
        self.batch_norm = BatchNormBlock(out_dim, self.use_bn, self.bn_momentum)

    def linear(self, x):
        _no_autograd(x, self.mlp.weight)
        return ops.gemm(x, aligned_weight(self.mlp.weight).t())

    def forward(self, x, batch=None):
        return self.batch_norm(self.linear(x), 1.0 if self.no_relu else 0.1)

    def __repr__(self):
        return "UnaryBlock(in_feat: {:d}, out_feat: {:d}, BN: {:s}, ReLU: {:s})".format(
            self.in_dim, self.out_dim, str(self.use_bn), str(not self.no_relu))


class LastUnaryBlock(nn.Module):
    """ref:models/blocks.py:510-533."""

    def __init__(self, in_dim, out_dim, use_bn, bn_momentum, no_relu=False):
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.mlp = nn.Linear(in_dim, out_dim, bias=False)

    def forward(self, x, batch=None):
        _no_autograd(x, self.mlp.weight)
        return ops.gemm(x, aligned_weight(self.mlp.weight).t())

    def __repr__(self):
        return "LastUnaryBlock(in_feat: {:d}, out_feat: {:d})".format(self.in_dim, self.out_dim)


def _geometry(block_name, layer_ind, batch):
    """ref:models/blocks.py:580-587, 652-659."""
    if "strided" in block_name:
        return batch["points"][layer_ind + 1], batch["points"][layer_ind], batch["pools"][layer_ind]
    return batch["points"][layer_ind], batch["points"][layer_ind], batch["neighbors"][layer_ind]


class SimpleBlock(nn.Module):
    """ref:models/blocks.py:536-590."""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super().__init__()
        current_extent = radius * config.KP_extent / config.conv_radius
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.layer_ind = layer_ind
        self.block_name = block_name
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.KPConv = KPConv(config.num_kernel_points, config.in_points_dim, in_dim, out_dim // 2, current_extent,
                             radius, fixed_kernel_points=config.fixed_kernel_points,
                             KP_influence=config.KP_influence, aggregation_mode=config.aggregation_mode,
                             deformable="deform" in block_name, modulated=config.modulated)
        self.batch_norm = BatchNormBlock(out_dim // 2, self.use_bn, self.bn_momentum)

    def forward(self, x, batch):
        q_pts, s_pts, neighb_inds = _geometry(self.block_name, self.layer_ind, batch)
        x = self.KPConv(q_pts, s_pts, neighb_inds, x)
        return self.batch_norm(x, 0.1)


class ResnetBottleneckBlock(nn.Module):
    """ref:models/blocks.py:593-678."""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super().__init__()
        current_extent = radius * config.KP_extent / config.conv_radius
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.block_name = block_name
        self.layer_ind = layer_ind
        self.in_dim = in_dim
        self.out_dim = out_dim
        if in_dim != out_dim // 4:
            self.unary1 = UnaryBlock(in_dim, out_dim // 4, self.use_bn, self.bn_momentum)
        else:
            self.unary1 = nn.Identity()
        self.KPConv = KPConv(config.num_kernel_points, config.in_points_dim, out_dim // 4, out_dim // 4,
                             current_extent, radius, fixed_kernel_points=config.fixed_kernel_points,
                             KP_influence=config.KP_influence, aggregation_mode=config.aggregation_mode,
                             deformable="deform" in block_name, modulated=config.modulated)
        self.batch_norm_conv = BatchNormBlock(out_dim // 4, self.use_bn, self.bn_momentum)
        self.unary2 = UnaryBlock(out_dim // 4, out_dim, self.use_bn, self.bn_momentum, no_relu=True)
        if in_dim != out_dim:
            self.unary_shortcut = UnaryBlock(in_dim, out_dim, self.use_bn, self.bn_momentum, no_relu=True)
        else:
            self.unary_shortcut = nn.Identity()

    def forward(self, features, batch):
        q_pts, s_pts, neighb_inds = _geometry(self.block_name, self.layer_ind, batch)
        x = self.unary1(features)
        x = self.KPConv(q_pts, s_pts, neighb_inds, x)
        x = self.batch_norm_conv(x, 0.1)
        shortcut = max_pool(features, neighb_inds) if "strided" in self.block_name else features
        if not self.use_bn:
            x = self.unary2(x)
            shortcut = self.unary_shortcut(shortcut)
            return torch.nn.functional.leaky_relu(x + shortcut, 0.1)
        # fused tail: LeakyReLU( IN(unary2.mlp(x)) + [IN(unary_shortcut.mlp(sc)) | sc] )   (:667-678)
        y = self.unary2.linear(x)
        y_stats = ops.instnorm_stats(y)
        if isinstance(self.unary_shortcut, nn.Identity):
            return ops.instnorm_apply(y, y_stats, 0.1, res=shortcut)
        sc = self.unary_shortcut.linear(shortcut)
        return ops.instnorm_apply(y, y_stats, 0.1, res=sc, res_stats=ops.instnorm_stats(sc))


class GlobalAverageBlock(nn.Module):
    """ref:models/blocks.py:681-691."""

    def forward(self, x, batch):
        return global_average(x, batch["stack_lengths"][-1])


class NearestUpsampleBlock(nn.Module):
    """ref:models/blocks.py:694-709."""

    def __init__(self, layer_ind):
        super().__init__()
        self.layer_ind = layer_ind

    def forward(self, x, batch, skip=None):
        """With `skip`, returns cat([upsampled x, skip], 1) built in place in a buffer whose rows are
        16-byte aligned (the concat of ref:models/architectures.py:568-569 fused with the upsample)."""
        inds = batch["upsamples"][self.layer_ind - 1]
        if skip is None:
            return closest_pool(x, inds)
        c, cs = x.shape[1], skip.shape[1]
        buf = torch.empty((inds.shape[0], (c + cs + 3) // 4 * 4), dtype=x.dtype, device=x.device)
        ops.gather_first(x, inds, out=buf[:, :c])
        buf[:, c:c + cs].copy_(skip)
        return buf[:, :c + cs]

    def __repr__(self):
        return "NearestUpsampleBlock(layer: {:d} -> {:d})".format(self.layer_ind, self.layer_ind - 1)


class MaxPoolBlock(nn.Module):
    """ref:models/blocks.py:712-723."""

    def __init__(self, layer_ind):
        super().__init__()
        self.layer_ind = layer_ind

    def forward(self, x, batch):
        return max_pool(x, batch["pools"][self.layer_ind + 1])


def block_decider(block_name, radius, in_dim, out_dim, layer_ind, config):
    """ref:models/blocks.py:387-430."""
    if block_name == "unary":
        return UnaryBlock(in_dim, out_dim, config.use_batch_norm, config.batch_norm_momentum)
    if block_name == "last_unary":
        return LastUnaryBlock(in_dim, config.final_feats_dim + 2, config.use_batch_norm, config.batch_norm_momentum)
    if block_name in ("simple", "simple_strided"):
        return SimpleBlock(block_name, in_dim, out_dim, radius, layer_ind, config)
    if block_name in ("resnetb", "resnetb_strided"):
        return ResnetBottleneckBlock(block_name, in_dim, out_dim, radius, layer_ind, config)
    if block_name in ("max_pool", "max_pool_wide"):
        return MaxPoolBlock(layer_ind)
    if block_name == "global_average":
        return GlobalAverageBlock()
    if block_name == "nearest_upsample":
        return NearestUpsampleBlock(layer_ind)
    if any(t in block_name for t in ("deformable", "invariant", "equivariant")):
        raise NotImplementedError("pcrcg_amd: block '%s' is not used by the shipped architectures" % block_name)
    raise ValueError("Unknown block name in the architecture definition : " + block_name)
