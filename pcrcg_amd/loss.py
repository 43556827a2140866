"""MetricLoss of the train / test step on the device (mirror of
ref:lib/loss.py:46-252 -- same constructor, same method names, same keys in the returned dict).

What differs from the reference, on purpose:
  * device-agnostic: every tensor lives where the inputs live (the reference hard-codes
    torch.device('cuda') for one label vector and builds two on the CPU, :194-198);
  * the saliency labels need arg-max over the src x tgt descriptor similarity of ALL points in the overlap
    region (:209-213); the reference materialises that matrix with torch.matmul, here the fused HIP kernel
    pcrcg_feature_argmax produces the arg-max without it;
  * precision / recall are computed on the device with the definition sklearn's
    precision_recall_fscore_support(average='binary') uses (:131-133), and returned as 0-dim tensors
    (no host round trip inside the step);
  * `set(...)` of the correspondence columns (:159-160) becomes torch.unique -- the set's iteration order is
    irrelevant to every output (all consumers are order-invariant sums or means).
The dense math -- the circle loss with the feature-match recall on the <= max_points^2 matrices, the class-weighted
BCE over N points -- runs in two fused HIP kernels with their gradients (csrc/lossops.hip, round 3: ~130 small torch ops
per step before) when the tensors are on the device; the torch formulation below stays as their mirror (`fused=False`,
and on the CPU).  The data-dependent selections around them (overlap region, max_points draw) are plain torch."""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops


def square_distance(src, dst, normalised=False):
    """ref:lib/utils.py:78-97."""
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    if normalised:
        dist += 2
    else:
        dist += torch.sum(src ** 2, dim=-1)[:, :, None]
        dist += torch.sum(dst ** 2, dim=-1)[:, None, :]
    return torch.clamp(dist, min=1e-12, max=None)


class _CircleLoss(torch.autograd.Function):
    """get_circle_loss + get_recall in one launch (pcrcg_circle_loss), gradients wrt both descriptor sets included."""

    @staticmethod
    def forward(ctx, src_feats, tgt_feats, coords_dist, cfg):
        from . import _lib
        a, b, cd = src_feats.detach().float().contiguous(), tgt_feats.detach().float().contiguous(), coords_dist.float().contiguous()
        n, c = a.shape
        out = torch.empty(2, dtype=torch.float32, device=a.device)
        da, db = torch.empty_like(a), torch.empty_like(b)
        ws = torch.empty(16 * n, dtype=torch.float32, device=a.device)          # pcrcg_circle_loss_ws_bytes(n)
        _lib.check(_lib.lib().pcrcg_circle_loss(a.data_ptr(), c, b.data_ptr(), c, cd.data_ptr(), n, n, c, *cfg, out.data_ptr(),
                                                da.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                                torch.cuda.current_stream().cuda_stream), "pcrcg_circle_loss")
        ctx.save_for_backward(da, db)
        loss, recall = out[0], out[1]              # bind the views ONCE: the marking applies to the object that is returned
        ctx.mark_non_differentiable(recall)
        return loss, recall

    @staticmethod
    def backward(ctx, g_loss, g_recall):
        da, db = ctx.saved_tensors
        return g_loss * da, g_loss * db, None, None


class _WeightedBCE(torch.autograd.Function):
    """get_weighted_bce_loss in three launches (pcrcg_weighted_bce), gradient wrt the prediction included."""

    @staticmethod
    def forward(ctx, prediction, gt):
        from . import _lib
        L = _lib.lib()
        p, g = prediction.detach().float().contiguous(), gt.float().contiguous()
        n = p.shape[0]
        out = torch.empty(3, dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        ws = torch.empty(int(L.pcrcg_weighted_bce_ws_bytes()), dtype=torch.uint8, device=p.device)
        _lib.check(L.pcrcg_weighted_bce(p.data_ptr(), g.data_ptr(), n, out.data_ptr(), grad.data_ptr(), ws.data_ptr(), ws.numel(),
                                        torch.cuda.current_stream().cuda_stream), "pcrcg_weighted_bce")
        ctx.save_for_backward(grad)
        loss, precision, recall = out[0], out[1], out[2]     # (bound once: see _CircleLoss)
        ctx.mark_non_differentiable(precision, recall)
        return loss, precision, recall

    @staticmethod
    def backward(ctx, g_loss, g_p, g_r):
        (grad,) = ctx.saved_tensors
        return g_loss * grad, None


class MetricLoss(torch.nn.Module):
    """Circle loss + overlap / saliency weighted BCE + feature-match recall (ref:lib/loss.py:46-70)."""

    def __init__(self, configs, log_scale=16, pos_optimal=0.1, neg_optimal=1.4, fused=True):
        super().__init__()
        self.fused = bool(fused)          # the dense parts in the fused HIP kernels when the tensors are on the device
        # NB the yaml's `log_scale` is NOT read by the reference either (ref:main.py:100)
        self.log_scale, self.pos_optimal, self.neg_optimal = log_scale, pos_optimal, neg_optimal
        self.image_feature = configs.get("image_feature", False)
        self.node_overlap = configs.get("node_overlap", False)
        self.quaternion = configs.get("quaternion", False)
        self.pos_margin, self.neg_margin = configs["pos_margin"], configs["neg_margin"]
        self.max_points = configs["max_points"]
        self.safe_radius = configs["safe_radius"]
        self.matchability_radius = configs["matchability_radius"]
        self.pos_radius = configs["pos_radius"]

    def get_circle_loss(self, coords_dist, feats_dist):
        """ref:lib/loss.py:71-104."""
        pos_mask = coords_dist < self.pos_radius
        neg_mask = coords_dist > self.safe_radius
        row_sel = ((pos_mask.sum(-1) > 0) * (neg_mask.sum(-1) > 0)).detach()
        col_sel = ((pos_mask.sum(-2) > 0) * (neg_mask.sum(-2) > 0)).detach()

        pos_weight = feats_dist - 1e5 * (~pos_mask).float()
        pos_weight = pos_weight - self.pos_optimal
        pos_weight = torch.max(torch.zeros_like(pos_weight), pos_weight).detach()
        neg_weight = feats_dist + 1e5 * (~neg_mask).float()
        neg_weight = self.neg_optimal - neg_weight
        neg_weight = torch.max(torch.zeros_like(neg_weight), neg_weight).detach()

        lse_pos_row = torch.logsumexp(self.log_scale * (feats_dist - self.pos_margin) * pos_weight, dim=-1)
        lse_pos_col = torch.logsumexp(self.log_scale * (feats_dist - self.pos_margin) * pos_weight, dim=-2)
        lse_neg_row = torch.logsumexp(self.log_scale * (self.neg_margin - feats_dist) * neg_weight, dim=-1)
        lse_neg_col = torch.logsumexp(self.log_scale * (self.neg_margin - feats_dist) * neg_weight, dim=-2)

        loss_row = F.softplus(lse_pos_row + lse_neg_row) / self.log_scale
        loss_col = F.softplus(lse_pos_col + lse_neg_col) / self.log_scale
        return (loss_row[row_sel].mean() + loss_col[col_sel].mean()) / 2

    def get_recall(self, coords_dist, feats_dist):
        """ref:lib/loss.py:106-116."""
        pos_mask = coords_dist < self.pos_radius
        has_pos = pos_mask.sum(-1) > 0
        n_gt_pos = has_pos.float().sum() + 1e-12
        _, sel_idx = torch.min(feats_dist, -1)
        sel_dist = torch.gather(coords_dist, dim=-1, index=sel_idx[:, None])[has_pos]
        n_pred_pos = (sel_dist < self.pos_radius).float().sum()
        return n_pred_pos / n_gt_pos

    def get_weighted_bce_loss(self, prediction, gt):
        """ref:lib/loss.py:118-135 -> (loss, precision, recall)."""
        if self.fused and prediction.is_cuda and prediction.dim() == 1 and prediction.numel() >= 1:
            return _WeightedBCE.apply(prediction, gt)
        class_loss = F.binary_cross_entropy(prediction, gt, reduction="none")
        w_negative = gt.sum() / gt.size(0)
        w_positive = 1 - w_negative
        weights = torch.where(gt >= 0.5, w_positive, w_negative)
        w_class_loss = torch.mean(weights * class_loss)
        # binary precision / recall of the rounded prediction (0/0 -> 0, as sklearn reports it)
        pred = prediction.detach().round() > 0.5
        true = gt.round() > 0.5
        tp = (pred & true).sum().float()
        n_pred, n_true = pred.sum().float(), true.sum().float()
        zero = torch.zeros((), device=prediction.device)
        precision = torch.where(n_pred > 0, tp / n_pred.clamp(min=1), zero)
        recall = torch.where(n_true > 0, tp / n_true.clamp(min=1), zero)
        return w_class_loss, precision, recall

    def prepare(self, inputs):
        """The part of forward() that reads no network output (ref:lib/loss.py:139-203,227-235: the moved source cloud,
        which points lie in the overlap, the overlap labels, the <= max_points correspondences drawn with the HOST numpy
        generator exactly as the reference does -- same draws for the same np.random state -- and their coordinate
        distances).  forward() calls it itself; a trainer may call it on a second stream while the network's forward runs
        (its data-dependent shapes cost host round trips that then wait for a few small kernels, not for the forward) and
        hand the result to forward(inputs, prepared=...).  Call order = draw order: one prepare() per forward()."""
        rot, trans = inputs["rot"], inputs["trans"]
        src_pcd, tgt_pcd = inputs["src_pcd_raw"], inputs["tgt_pcd_raw"]
        dev = src_pcd.device
        correspondence = inputs["correspondences"].to(dev).long()
        p = dict()
        src_pcd = (torch.matmul(rot, src_pcd.transpose(0, 1)) + trans).transpose(0, 1)
        p["src_idx"] = src_idx = torch.unique(correspondence[:, 0])
        p["tgt_idx"] = tgt_idx = torch.unique(correspondence[:, 1])
        # overlap labels: a point is "in the overlap" iff it appears in a correspondence (:193-203)
        src_gt = torch.zeros(src_pcd.size(0), device=dev)
        src_gt[src_idx] = 1.
        tgt_gt = torch.zeros(tgt_pcd.size(0), device=dev)
        tgt_gt[tgt_idx] = 1.
        p["overlap_labels"] = torch.cat((src_gt, tgt_gt))
        # rows of the stacked score vector the saliency loss reads (unique by construction: one gather forward, one
        # index_add backward instead of two indexing nodes whose backward sorts its indices)
        p["saliency_rows"] = torch.cat((src_idx, tgt_idx + src_pcd.size(0)))
        p["src_pcd_sel"], p["tgt_pcd_sel"] = src_pcd[src_idx], tgt_pcd[tgt_idx]
        # correspondences closer than pos_radius, capped to max_points (:227-233)
        c_dist = torch.norm(src_pcd[correspondence[:, 0]] - tgt_pcd[correspondence[:, 1]], dim=1)
        correspondence = correspondence[c_dist < self.pos_radius - 0.001]
        if correspondence.size(0) > self.max_points:
            choice = np.random.permutation(correspondence.size(0))[:self.max_points]
            correspondence = correspondence[torch.from_numpy(choice).to(dev)]
        p["src_sel"], p["tgt_sel"] = correspondence[:, 0], correspondence[:, 1]
        p["coords_dist"] = torch.sqrt(square_distance(src_pcd[p["src_sel"]][None], tgt_pcd[p["tgt_sel"]][None]).squeeze(0))
        return p

    def forward(self, inputs, prepared=None):
        """ref:lib/loss.py:139-252.  inputs: rot [3,3], trans [3,1], src_feats [N,C], tgt_feats [M,C],
        src_pcd_raw [N,3], tgt_pcd_raw [M,3], correspondences [K,2] int64, scores_overlap / scores_saliency
        [N+M] -- all on one device.  prepared: the result of prepare(inputs), when the caller ran it ahead."""
        src_feats, tgt_feats = inputs["src_feats"], inputs["tgt_feats"]
        scores_overlap, scores_saliency = inputs["scores_overlap"], inputs["scores_saliency"]
        p = prepared if prepared is not None else self.prepare(inputs)
        src_idx, tgt_idx = p["src_idx"], p["tgt_idx"]
        n_src = inputs["src_pcd_raw"].size(0)
        stats = dict()

        if self.node_overlap:
            loss, a, b = self.get_weighted_bce_loss(inputs["node_overlap_score_pred"], inputs["node_overlap_gt"])
            stats["node_overlap_loss"], stats["node_overlap_recall"], stats["node_overlap_precision"] = loss, a, b
        if self.quaternion:
            q = F.mse_loss(inputs["quaternion_pred"], inputs["quaternion_gt"], reduction="sum")
            t = F.mse_loss(inputs["trans_pred"], inputs["trans_gt"], reduction="sum")
            stats["pose_loss"] = q + t

        # overlap BCE (:193-203)
        class_loss, cls_precision, cls_recall = self.get_weighted_bce_loss(scores_overlap, p["overlap_labels"])
        stats["overlap_loss"], stats["overlap_recall"], stats["overlap_precision"] = class_loss, cls_recall, cls_precision

        # saliency BCE, supervised in the overlap region only (:205-225): a point is matchable iff its nearest
        # descriptor on the other side lies within matchability_radius
        src_feats_sel, src_pcd_sel = src_feats[src_idx], p["src_pcd_sel"]
        tgt_feats_sel, tgt_pcd_sel = tgt_feats[tgt_idx], p["tgt_pcd_sel"]
        with torch.no_grad():
            idx12 = ops.feature_argmax(src_feats_sel.detach().float(), tgt_feats_sel.detach().float())
            idx21 = ops.feature_argmax(tgt_feats_sel.detach().float(), src_feats_sel.detach().float())
        distance_1 = torch.norm(src_pcd_sel - tgt_pcd_sel[idx12], p=2, dim=1)
        distance_2 = torch.norm(tgt_pcd_sel - src_pcd_sel[idx21], p=2, dim=1)
        gt_labels = torch.cat(((distance_1 < self.matchability_radius).float(),
                               (distance_2 < self.matchability_radius).float()))
        saliency = scores_saliency.index_select(0, p["saliency_rows"]) if "saliency_rows" in p else \
            torch.cat((scores_saliency[:n_src][src_idx], scores_saliency[n_src:][tgt_idx]))
        class_loss, cls_precision, cls_recall = self.get_weighted_bce_loss(saliency, gt_labels)
        stats["saliency_loss"], stats["saliency_recall"], stats["saliency_precision"] = class_loss, cls_recall, cls_precision

        # circle loss + feature-match recall on the drawn correspondences (:227-252)
        src_feats, tgt_feats = src_feats[p["src_sel"]], tgt_feats[p["tgt_sel"]]
        coords_dist = p["coords_dist"]
        n_sel = src_feats.shape[0]
        if self.fused and src_feats.is_cuda and 1 <= n_sel <= 512 and src_feats.shape[1] <= 64:
            cfg = (float(self.pos_radius), float(self.safe_radius), float(self.pos_optimal), float(self.neg_optimal),
                   float(self.pos_margin), float(self.neg_margin), float(self.log_scale))
            stats["circle_loss"], stats["recall"] = _CircleLoss.apply(src_feats, tgt_feats, coords_dist, cfg)
            return stats
        feats_dist = torch.sqrt(square_distance(src_feats[None], tgt_feats[None], normalised=True)).squeeze(0)
        stats["circle_loss"] = self.get_circle_loss(coords_dist, feats_dist)
        stats["recall"] = self.get_recall(coords_dist, feats_dist)
        return stats
