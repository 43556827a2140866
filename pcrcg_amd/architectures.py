"""KPFCNN on the MI355X -- host-side mirror of ref:models/architectures.py:37-174 (constructor) and
:181-191, :516-610 (forward, geometry-only branch).  Same constructor argument, same forward
signature and result dict, same state_dict keys and shapes; the forward pass runs in the HIP kernels
behind pcrcg_amd.ops."""
import os
import threading

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .blocks import NearestUpsampleBlock, block_decider
from .config import as_config
from .gcn import GCN

_RUNNER_LOCK = threading.Lock()   # module-level: a lock attribute would make the nn.Module uncopyable


class KPFCNN(nn.Module):

    def __init__(self, config):
        super().__init__()
        config = as_config(config)
        if config.get("node_overlap", False) or config.get("quaternion", False):
            raise NotImplementedError("pcrcg_amd.KPFCNN: the node-overlap / quaternion heads are outside the "
                                      "accelerated path (SURVEY.md 8f)")
        # PCR-CG's image-feature injection (ref:models/architectures.py:48-50,195-514): the 2-D backbone is the caller's
        # (forward(batch, backbone2d)); the gather of its per-pixel features into the [N, 129] point features is ours
        self.image_feature = bool(config.get("image_feature", False))
        self.img_num = int(config.get("img_num", 0) or 0)
        if self.image_feature and (self.img_num not in (1, 2, 3) or config.in_feats_dim != 129):
            raise ValueError("pcrcg_amd.KPFCNN: image_feature needs img_num in {1, 2, 3} and in_feats_dim = 129")
        # bf16 feature-storage VARIANT of the inference forward (include/pcrcg.h pcrcg_model.feature_bf16): off by
        # default, not in the reference; outside the fp32 parity bound (tests/test_bf16_gpu.py states its error)
        self.feature_bf16 = bool(config.get("feature_bf16", False))
        layer = 0
        r = config.first_subsampling_dl * config.conv_radius
        in_dim = config.in_feats_dim
        out_dim = config.first_feats_dim
        self.K = config.num_kernel_points
        self.epsilon = torch.nn.Parameter(torch.tensor(-5.0))
        self.final_feats_dim = config.final_feats_dim

        # encoder (:58-100)
        self.encoder_blocks = nn.ModuleList()
        self.encoder_skip_dims = []
        self.encoder_skips = []
        for block_i, block in enumerate(config.architecture):
            if ("equivariant" in block) and (not out_dim % 3 == 0):
                raise ValueError("Equivariant block but features dimension is not a factor of 3")
            if np.any([tmp in block for tmp in ["pool", "strided", "upsample", "global"]]):
                self.encoder_skips.append(block_i)
                self.encoder_skip_dims.append(in_dim)
            if "upsample" in block:
                break
            self.encoder_blocks.append(block_decider(block, r, in_dim, out_dim, layer, config))
            in_dim = out_dim // 2 if "simple" in block else out_dim
            if "pool" in block or "strided" in block:
                layer += 1
                r *= 2
                out_dim *= 2

        # bottleneck + GNN (:102-111)
        gnn_feats_dim = config.gnn_feats_dim
        self.bottle = nn.Conv1d(in_dim, gnn_feats_dim, kernel_size=1, bias=True)
        self.gnn = GCN(config.num_head, gnn_feats_dim, config.dgcnn_k, config.nets)
        self.proj_gnn = nn.Conv1d(gnn_feats_dim, gnn_feats_dim, kernel_size=1, bias=True)
        self.proj_score = nn.Conv1d(gnn_feats_dim, 1, kernel_size=1, bias=True)

        # decoder (:113-153)
        out_dim = gnn_feats_dim + 2
        self.decoder_blocks = nn.ModuleList()
        self.decoder_concats = []
        start_i = 0
        for block_i, block in enumerate(config.architecture):
            if "upsample" in block:
                start_i = block_i
                break
        for block_i, block in enumerate(config.architecture[start_i:]):
            if block_i > 0 and "upsample" in config.architecture[start_i + block_i - 1]:
                in_dim += self.encoder_skip_dims[layer]
                self.decoder_concats.append(block_i)
            self.decoder_blocks.append(block_decider(block, r, in_dim, out_dim, layer, config))
            in_dim = out_dim
            if "upsample" in block:
                layer -= 1
                r *= 0.5
                out_dim = out_dim // 2
        self._eps_cache = None
        self._runner = None
        # the C++ runner enqueues the forward (one FFI call); set use_runner = False for the op-by-op mirror forward_ops
        self.use_runner = bool(config.use_batch_norm)

    def regular_score(self, score):
        """ref:models/architectures.py:176-179."""
        score = torch.where(torch.isnan(score), torch.zeros_like(score), score)
        score = torch.where(torch.isinf(score), torch.zeros_like(score), score)
        return score

    def _temperature(self):
        key = (self.epsilon._version, self.epsilon.data_ptr())
        if self._eps_cache is None or self._eps_cache[0] != key:
            self._eps_cache = (key, float(torch.exp(self.epsilon.detach()).item()) + 0.03)
        return self._eps_cache[1]

    @staticmethod
    def _conv1x1(layer, x):
        return ops.gemm(x, layer.weight.data.squeeze(-1).t(), bias=layer.bias.data)

    IMAGE_WIDTH = 132      # the 129-channel input as the runners take it: rows of whole float4s, three zero columns

    def image_list(self, batch, backbone2d=None):
        """The projections of a batch as ops.inject_image_features takes them, in the reference's write order."""
        n = int(batch["points"][0].shape[0])
        len_src = int(batch["src_pcd_raw"].shape[0])
        dev = batch["points"][0].device
        images = []
        for side in ("src", "tgt"):
            for i in range(self.img_num, 0, -1):              # the reference writes image 3, 2, 1: image 1 wins (:242-247)
                key = f"{side}{i}_feature2d"
                if key in batch:
                    fmap = batch[key]
                elif backbone2d is not None:
                    with torch.no_grad():
                        fmap = backbone2d(batch[f"{side}_color{i}"].unsqueeze(0).to(dev)).squeeze(0)
                else:
                    raise RuntimeError(f"pcrcg_amd.KPFCNN: image_feature needs backbone2d or batch['{key}']")
                valid = batch.get(f"{side}_valid_map{i}") if self.img_num < 3 else None   # :196-252 has no valid maps
                images.append(dict(fmap=fmap.detach().to(dev, torch.float32), inds2d=batch[f"{side}{i}_inds2d"].to(dev),
                                   inds3d=batch[f"{side}{i}_inds3d"].to(dev), target=side == "tgt",
                                   valid=None if valid is None else valid.to(dev)))
        return n, len_src, images

    def image_features(self, batch, backbone2d=None, width=None):
        """ref:models/architectures.py:195-514: x = ones [N, 129] with the 2-D features of the projected points
        scattered in (pcrcg_inject_image_features).  The 2-D feature maps come from `backbone2d` applied to
        batch['{src,tgt}_color{i}'] as in the reference, or -- precomputed -- from batch['{src,tgt}{i}_feature2d'].
        width=IMAGE_WIDTH: rows padded with zero columns (what the C++ runners take)."""
        n, len_src, images = self.image_list(batch, backbone2d)
        return ops.inject_image_features(n, len_src, images, channels=128, width=width)

    def forward(self, batch, backbone2d=None):
        training = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if self.image_feature:
            # the C++ runners (inference and train step) take the 129 channels in rows of 132 floats -- the first KPConv's
            # gather kernel reads whole float4s -- against weights padded with zero rows; the op-by-op paths take [N, 129]
            wide = self.use_runner
            batch = dict(batch)
            batch["features"] = self.image_features(batch, backbone2d, width=self.IMAGE_WIDTH if wide else None)
        if training:
            # training: forward with a tape + backward in C++ (pcrcg_amd/train_runner.py, one autograd node whose backward
            # accumulates every parameter's gradient into p.grad); configurations it does not cover take the op-by-op
            # autograd composition of the same kernels (pcrcg_amd/train_forward.py)
            tr = self.train_runner()
            if tr is not None:
                return tr.forward(batch)
            from .train_forward import forward_train
            return forward_train(self, batch)
        if self.use_runner:
            # the whole forward below, enqueued by one call into the C++ runner (csrc/runner.hip)
            return self.runner().forward(batch)
        return self.forward_ops(batch)

    def runner(self):
        """The (lazily created) descriptor cache of the C++ runner; creation is serialised because forwards may
        be enqueued from several host threads (pcrcg_amd/pairstream.py)."""
        if self._runner is None:
            with _RUNNER_LOCK:
                if self._runner is None:
                    from .runner import Runner
                    self._runner = Runner(self)
        return self._runner

    def train_runner(self):
        """The (lazily created) C++ train-step runner (pcrcg_amd/train_runner.py): forward with a tape + backward, one
        library call each.  None when this configuration needs the op-by-op autograd path (pcrcg_amd/train_forward.py):
        no InstanceNorm (use_batch_norm False).  (Round 6: PCR-CG's shipped configuration -- the 129-channel
        image-feature input -- goes through it too, its first KPConv against zero-padded weights.)"""
        if not self.use_runner:
            return None
        if getattr(self, "_train_runner", None) is None:
            with _RUNNER_LOCK:
                if getattr(self, "_train_runner", None) is None:
                    from .train_runner import TrainRunner
                    self._train_runner = TrainRunner(self)
        return self._train_runner

    def forward_ops(self, batch):
        """Op-by-op forward through pcrcg_amd.ops (one FFI call per kernel); same kernels, same results
        as the runner -- kept as the readable mirror of the reference's forward and for debugging."""
        x = batch["features"].clone().detach()                                   # :183
        if "stack_lengths_host" in batch:
            len_src_c = int(batch["stack_lengths_host"][-1][0])
        else:
            len_src_c = int(batch["stack_lengths"][-1][0])                       # :187
        pcd_c = batch["points"][-1]
        src_pcd_c, tgt_pcd_c = pcd_c[:len_src_c], pcd_c[len_src_c:]

        # 1. joint encoder (:519-524)
        skip_x = []
        for block_i, block_op in enumerate(self.encoder_blocks):
            if block_i in self.encoder_skips:
                skip_x.append(x)
            x = block_op(x, batch)

        # 2. bottleneck projection (:527-528), row-major [N, C]
        feats_c = self._conv1x1(self.bottle, x)

        # 3. GNN (:532-536)
        src_f, tgt_f = self.gnn(src_pcd_c, tgt_pcd_c, feats_c[:len_src_c], feats_c[len_src_c:])
        feats_c = torch.cat([src_f, tgt_f], 0)
        feats_c = self._conv1x1(self.proj_gnn, feats_c)                          # :538
        scores_c = self._conv1x1(self.proj_score, feats_c)                       # :539  [N, 1]
        feats_norm = F.normalize(feats_c, p=2, dim=1)                            # :541

        # 4. cross-cloud saliency (:556-565)
        src_n, tgt_n = feats_norm[:len_src_c], feats_norm[len_src_c:]
        inv_t = 1.0 / self._temperature()
        p_st = ops.softmax_rows_(ops.gemm(src_n, tgt_n.t()), inv_t)
        p_ts = ops.softmax_rows_(ops.gemm(tgt_n, src_n.t()), inv_t)
        s1 = ops.gemm(p_st, scores_c[len_src_c:])
        s2 = ops.gemm(p_ts, scores_c[:len_src_c])
        x = torch.cat([scores_c, torch.cat([s1, s2], 0), feats_c], 1)

        # decoder (:567-570)
        fused_concat = False
        for block_i, block_op in enumerate(self.decoder_blocks):
            if block_i in self.decoder_concats and not fused_concat:
                x = torch.cat([x, skip_x.pop()], 1)
            fused_concat = isinstance(block_op, NearestUpsampleBlock) and (block_i + 1) in self.decoder_concats
            x = block_op(x, batch, skip_x.pop()) if fused_concat else block_op(x, batch)
        fd = self.final_feats_dim
        feats_f = x[:, :fd]
        scores_overlap = x[:, fd]
        scores_saliency = x[:, fd + 1]
        scores_overlap = torch.clamp(torch.sigmoid(scores_overlap.view(-1)), min=0, max=1)     # :576-577
        scores_saliency = torch.clamp(torch.sigmoid(scores_saliency.view(-1)), min=0, max=1)
        scores_overlap = self.regular_score(scores_overlap)
        scores_saliency = self.regular_score(scores_saliency)
        feats_f = F.normalize(feats_f, p=2, dim=1)                                             # :582
        return {"feats_f": feats_f, "scores_overlap": scores_overlap, "scores_saliency": scores_saliency}
