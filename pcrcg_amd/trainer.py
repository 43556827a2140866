"""The train step of the hot path: host-side mirror of
Trainer.inference_one_batch / the optimisation block of inference_one_epoch (ref:lib/trainer.py:216-300,
354-361) for ONE pair per rank, plus the data-parallel piece the reference lacks.

  forward (train_forward, HIP kernels)  ->  MetricLoss  ->  total = unweighted sum of the loss keys
  (ref:lib/trainer.py:255-260)  ->  backward (HIP backward kernels through torch.autograd's graph)
  ->  ONE all-reduce(sum) of a flat fp32 gradient bucket over the ranks (RCCL; every parameter's .grad is
  a view into the bucket, so there is no pack / unpack copy), scaled by 1/world  ->  the reference's
  NaN/Inf gradient check (ref:lib/utils.py:100-111) evaluated on the REDUCED bucket, which makes the
  skip decision identical on every rank without a second collective  ->  SGD step.

Pairs shard over ranks exactly as in the forward benchmark (pair i -> rank i mod world); the all-reduce is
the only data-path collective of the step.  Optimiser and scheduler objects are torch.optim (plumbing):
SGD lr 0.005, momentum 0.98, weight decay 1e-6, ExponentialLR 0.95 per epoch
(ref:configs/train/indoor.yaml:65-74, ref:main.py:59-78)."""
import torch

from .train_forward import forward_train

LOSS_KEYS = ("circle_loss", "overlap_loss", "saliency_loss", "node_overlap_loss", "pose_loss")   # ref:lib/trainer.py:255


class Trainer:
    def __init__(self, model, desc_loss, lr=0.005, momentum=0.98, weight_decay=1e-6, scheduler_gamma=0.95,
                 iter_size=1, process_group=None):
        self.model, self.desc_loss = model, desc_loss
        self.iter_size = iter_size
        self.group = process_group
        self.params = [p for p in model.parameters() if p.requires_grad]
        # one flat gradient bucket; p.grad are views into it
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.optimizer = torch.optim.SGD(self.params, lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, gamma=scheduler_gamma)
        self.skipped_steps = 0
        self._iter = 0

    # ---- one pair ---------------------------------------------------------------------------------
    def inference_one_batch(self, inputs, phase):
        """-> dict of python floats (the reference detaches every stat, ref:lib/trainer.py:306-316)."""
        assert phase in ("train", "val", "test")
        train = phase == "train"
        self.model.train(train)
        with torch.set_grad_enabled(train):
            output = forward_train(self.model, inputs) if train else self.model(inputs)
            len_src = int(inputs["stack_lengths_host"][0][0]) if "stack_lengths_host" in inputs \
                else int(inputs["stack_lengths"][0][0])
            feats = output["feats_f"]
            loss_input = {
                "src_feats": feats[:len_src], "tgt_feats": feats[len_src:],
                "rot": inputs["rot"], "trans": inputs["trans"],
                "scores_overlap": output["scores_overlap"], "scores_saliency": output["scores_saliency"],
                "src_pcd_raw": inputs["src_pcd_raw"], "tgt_pcd_raw": inputs["tgt_pcd_raw"],
                "correspondences": inputs["correspondences"],
            }
            res = self.desc_loss(loss_input)
            if train:
                c_loss = sum(res[k] for k in res if k in LOSS_KEYS)
                c_loss.backward()               # accumulates into the flat bucket (iter_size > 1 sums pairs)
                res["total_loss"] = c_loss
        return {k: float(v.detach()) if isinstance(v, torch.Tensor) else float(v) for k, v in res.items()}

    # ---- optimisation block -------------------------------------------------------------------------
    def all_reduce_gradients(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size(self.group)
            if world > 1:
                dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
                self.flat_grad.div_(world)

    def gradient_valid(self):
        """validate_gradient (ref:lib/utils.py:100-111) on the reduced bucket: same answer on every rank."""
        return bool(torch.isfinite(self.flat_grad).all().item())

    def optimizer_step(self):
        self.all_reduce_gradients()
        ok = self.gradient_valid()
        if ok:
            self.optimizer.step()
        else:
            self.skipped_steps += 1
        self.flat_grad.zero_()                   # optimizer.zero_grad() would detach the views
        return ok

    def train_step(self, inputs):
        """One iteration of the reference's epoch loop for phase 'train' (ref:lib/trainer.py:340-361)."""
        stats = self.inference_one_batch(inputs, "train")
        self._iter += 1
        if self._iter % self.iter_size == 0:
            stats["gradient_valid"] = float(self.optimizer_step())
        return stats

    def end_epoch(self):
        self.scheduler.step()
