"""The train step of the hot path: host-side mirror of
Trainer.inference_one_batch / the optimisation block of inference_one_epoch (ref:lib/trainer.py:216-300,
354-361) for ONE pair per rank, plus the data-parallel piece the reference lacks.

  forward (train_forward, HIP kernels)  ->  MetricLoss  ->  total = unweighted sum of the loss keys
  (ref:lib/trainer.py:255-260)  ->  backward (HIP backward kernels through torch.autograd's graph)
  ->  all-reduce(sum) of a flat fp32 gradient bucket over the ranks (RCCL; every parameter's .grad is a view
  into the bucket, so there is no pack / unpack copy), issued as a few large contiguous slices that start as
  soon as backward has produced them (the exchange overlaps the rest of backward), scaled by 1/world  ->  the reference's
  NaN/Inf gradient check (ref:lib/utils.py:100-111) evaluated on the REDUCED bucket, which makes the
  skip decision identical on every rank without a second collective  ->  SGD step.

Pairs shard over ranks exactly as in the forward benchmark (pair i -> rank i mod world); the all-reduce is
the only data-path collective of the step.  Optimiser and scheduler objects are torch.optim (plumbing):
SGD lr 0.005, momentum 0.98, weight decay 1e-6, ExponentialLR 0.95 per epoch
(ref:configs/train/indoor.yaml:65-74, ref:main.py:59-78)."""
import torch

from .train_forward import forward_train

LOSS_KEYS = ("circle_loss", "overlap_loss", "saliency_loss", "node_overlap_loss", "pose_loss")   # ref:lib/trainer.py:255


class GradientBucket:
    """One flat fp32 buffer holding every parameter's gradient (p.grad are views into it): the whole
    data-parallel exchange of a step is a single all-reduce, and the NaN/Inf check reads one tensor."""

    ALIGN = 64        # floats: every parameter's slice starts on a 256-byte boundary (the kernels' vector loads want 16)

    @classmethod
    def padded(cls, n):
        return (n + cls.ALIGN - 1) // cls.ALIGN * cls.ALIGN

    def __init__(self, params, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.sizes = [self.padded(p.numel()) for p in self.params]      # slice lengths in the flat buffer (zero padding)
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=self.params[0].device)
        off = 0
        for p, n in zip(self.params, self.sizes):
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += n

        self._chunks = None       # overlap mode: [(start, end, n_params)] slices of the flat buffer
        self._armed = False
        # PCRCG_FORCE_DIST=1: run the collective even in a one-rank group (RCCL all-reduce of the bucket with itself:
        # the identity) so that the exchange path can be exercised on a single GPU
        import os
        self.force = os.environ.get("PCRCG_FORCE_DIST") == "1"

    # ---- overlap of the exchange with backward -----------------------------------------------------
    def enable_overlap(self, n_chunks=4):
        """Split the bucket into `n_chunks` contiguous slices (parameters are laid out in forward order, so
        backward completes the LAST slice first) and all-reduce each slice asynchronously as soon as every
        parameter in it has received its gradient, while backward is still running on the earlier layers.
        xGMI is point-to-point and a ring all-reduce is per-link bound (~1.4 ms for the whole
        bucket), so a few large slices, not many small ones."""
        sizes = self.sizes
        total, target = sum(sizes), sum(sizes) / float(n_chunks)
        self._chunks, self._chunk_of = [], {}
        start = acc = count = 0
        for i, n in enumerate(sizes):
            self._chunk_of[id(self.params[i])] = len(self._chunks)
            acc += n
            count += 1
            if acc >= target * (len(self._chunks) + 1) - 1e-9 or i == len(sizes) - 1:
                self._chunks.append((start, acc, count))
                start, count = acc, 0
        assert self._chunks[-1][1] == total
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    def arm(self, on=True):
        """Call before the backward pass whose gradients complete the optimiser step (the last of iter_size)."""
        self._armed = bool(on) and self._chunks is not None and (self._world() > 1 or self._forced())
        self._seen = [0] * len(self._chunks or [])
        self._handles = {}

    def _forced(self):
        import torch.distributed as dist
        return self.force and dist.is_available() and dist.is_initialized()

    def _world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def _on_grad(self, p):
        if not self._armed:
            return
        import torch.distributed as dist
        k = self._chunk_of[id(p)]
        self._seen[k] += 1
        a, b, n = self._chunks[k]
        if self._seen[k] == n and k not in self._handles:
            self._handles[k] = dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def all_reduce_mean(self):
        """Finish the exchange: wait for the slices already in flight, reduce the rest (slices holding a
        parameter that received no gradient never complete on their own), divide by the world size."""
        import torch.distributed as dist
        world = self._world()
        if world <= 1 and not self._forced():
            self._armed = False
            return
        if self._armed:
            for k, (a, b, _) in enumerate(self._chunks):
                h = self._handles.get(k)
                if h is None:
                    h = dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                h.wait()
            self._armed = False
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(world)

    def finite(self):
        return bool(self.finite_tensor().item())

    def finite_tensor(self):
        """The check as a 0-dim tensor on the bucket's device, non-zero = all finite (no host round trip: Trainer.train_step
        reads it together with the step's statistics).  On a HIP device ONE pass over the bucket (pcrcg_nonfinite_flag)
        instead of torch's abs / compare / compare / all chain."""
        if self.flat.is_cuda and self.flat.dtype == torch.float32 and self.flat.data_ptr() % 16 == 0:
            from . import _lib
            flag = torch.empty(1, dtype=torch.float32, device=self.flat.device)
            _lib.check(_lib.lib().pcrcg_nonfinite_flag(self.flat.data_ptr(), self.flat.numel(), flag.data_ptr(),
                                                       torch.cuda.current_stream().cuda_stream), "pcrcg_nonfinite_flag")
            return flag[0] == 0
        return torch.isfinite(self.flat).all()

    def zero(self):
        self.flat.zero_()                        # optimizer.zero_grad() would drop the views


class FlatSGD(torch.optim.Optimizer):
    """torch.optim.SGD(lr, momentum, weight_decay) over a model whose parameters and gradients are views of two flat
    device buffers: the whole step is ONE launch (pcrcg_sgd_step, csrc/lossops.hip) instead of the multi-tensor kernels
    of torch.optim over 170 tensors, and it clears the gradient bucket on its way.  An ordinary Optimizer otherwise
    (param_groups carry lr / momentum / weight_decay, so lr schedulers work on it; state_dict / load_state_dict speak
    torch.optim.SGD's per-parameter format)."""

    def __init__(self, params, flat_param, flat_grad, lr, momentum=0.0, weight_decay=0.0, sizes=None):
        params = list(params)
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.flat_param, self.flat_grad = flat_param, flat_grad
        self._flat_params = params
        self._sizes = list(sizes) if sizes is not None else [GradientBucket.padded(p.numel()) for p in params]
        if sum(self._sizes) != flat_param.numel():
            raise ValueError("FlatSGD: slice sizes do not add up to the flat parameter buffer")
        self.momentum_flat = torch.zeros_like(flat_param)      # the only state: one flat momentum buffer (device)

    # ---- checkpoints: torch.optim.SGD's format, so that they are interchangeable with the reference's
    # (ref:lib/trainer.py:133,174 saves / loads optimizer.state_dict()) ---------------------------------------------
    def _slices(self):
        off = 0
        for p, size in zip(self._flat_params, self._sizes):
            yield p, off, p.numel()
            off += size

    def state_dict(self):
        """Per-parameter momentum_buffer slices (clones, shaped like the parameters) under integer parameter ids and one
        param_group with torch.optim.SGD's keys: torch.optim.SGD(model.parameters(), ...).load_state_dict() accepts it."""
        g = self.param_groups[0]
        state = {i: {"momentum_buffer": self.momentum_flat[off:off + n].view_as(p).clone()}
                 for i, (p, off, n) in enumerate(self._slices())}
        group = {"lr": g["lr"], "momentum": g["momentum"], "dampening": 0, "weight_decay": g["weight_decay"],
                 "nesterov": False, "maximize": False, "foreach": None, "differentiable": False, "fused": None,
                 "params": list(range(len(self._flat_params)))}
        for k, v in g.items():                                  # (scheduler bookkeeping such as initial_lr)
            if k not in group and k != "params":
                group[k] = v
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, state_dict):
        """Accepts torch.optim.SGD's / this class's state_dict (per-parameter momentum buffers, any device; missing or
        None = zero, as before SGD's first step) and the round-3 form (one 'flat' buffer).  Values are COPIED into the
        existing device buffer: nothing loaded from a checkpoint is ever handed to the kernel directly."""
        groups = state_dict["param_groups"]
        if len(groups) != 1:
            raise ValueError("FlatSGD.load_state_dict: exactly one param_group expected")
        st = state_dict.get("state", {})
        if "flat" in st:                                        # round-3 checkpoints of this class
            buf = st["flat"]["momentum_buffer"]
            if buf.numel() != self.momentum_flat.numel():
                raise ValueError("FlatSGD.load_state_dict: flat momentum buffer of %d elements, %d expected"
                                 % (buf.numel(), self.momentum_flat.numel()))
            self.momentum_flat.copy_(buf.to(self.momentum_flat.device, torch.float32).reshape(-1))
        else:
            ids = groups[0]["params"]
            if len(ids) != len(self._flat_params):
                raise ValueError("FlatSGD.load_state_dict: %d parameters in the checkpoint, %d in the model"
                                 % (len(ids), len(self._flat_params)))
            self.momentum_flat.zero_()
            for pid, (p, off, n) in zip(ids, self._slices()):
                entry = st.get(pid, st.get(str(pid)))
                buf = None if entry is None else entry.get("momentum_buffer")
                if buf is None:
                    continue
                if buf.numel() != n:
                    raise ValueError("FlatSGD.load_state_dict: momentum buffer of parameter %s has %d elements, %d expected"
                                     % (pid, buf.numel(), n))
                self.momentum_flat[off:off + n].copy_(buf.to(self.momentum_flat.device, torch.float32).reshape(-1))
        g = self.param_groups[0]
        if groups[0].get("nesterov") or groups[0].get("dampening", 0) not in (0, 0.0) or groups[0].get("maximize"):
            raise ValueError("FlatSGD.load_state_dict: nesterov / dampening / maximize are not implemented by pcrcg_sgd_step")
        for k, v in groups[0].items():
            if k != "params" and k not in ("dampening", "nesterov", "maximize", "foreach", "differentiable", "fused"):
                g[k] = v

    @staticmethod
    def flatten(params, sizes):
        """Move the parameters into one flat fp32 buffer laid out like the gradient bucket (slice lengths `sizes`, zero
        padding between them; p.data become views of it) -> the buffer."""
        params = list(params)
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=params[0].device)
        off = 0
        for p, size in zip(params, sizes):
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view_as(p)
            off += size
        return flat

    @torch.no_grad()
    def step(self, closure=None, zero_grad=False):
        from . import _lib
        g = self.param_groups[0]
        _lib.check(_lib.lib().pcrcg_sgd_step(self.flat_param.data_ptr(), self.flat_grad.data_ptr(),
                                             self.momentum_flat.data_ptr(), self.flat_param.numel(),
                                             float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]), int(bool(zero_grad)),
                                             torch.cuda.current_stream().cuda_stream), "pcrcg_sgd_step")
        return None


class Trainer:
    def __init__(self, model, desc_loss, lr=0.005, momentum=0.98, weight_decay=1e-6, scheduler_gamma=0.95,
                 iter_size=1, process_group=None, overlap_chunks=4, use_cpp_runner=True):
        self.model, self.desc_loss = model, desc_loss
        self.use_cpp_runner = bool(use_cpp_runner)
        self.iter_size = iter_size
        self.bucket = GradientBucket(model.parameters(), process_group)
        if overlap_chunks and overlap_chunks > 1:
            self.bucket.enable_overlap(overlap_chunks)
        self.params = self.bucket.params
        self.flat_grad = self.bucket.flat
        # one-launch SGD over flat parameter / gradient buffers on the GPU (every requires_grad parameter of the model is in
        # the bucket); torch.optim.SGD elsewhere (CPU runs of the host logic) and for models with frozen parameters mixed in
        self.flat_param = None
        if self.params[0].is_cuda and all(p.dtype == torch.float32 and p.is_cuda for p in self.params):
            self.flat_param = FlatSGD.flatten(self.params, self.bucket.sizes)
            self.optimizer = FlatSGD(self.params, self.flat_param, self.flat_grad, lr=lr, momentum=momentum,
                                     weight_decay=weight_decay, sizes=self.bucket.sizes)
        else:
            self.optimizer = torch.optim.SGD(self.params, lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, gamma=scheduler_gamma)
        self.skipped_steps = 0
        self._iter = 0

    # ---- one pair ---------------------------------------------------------------------------------
    def inference_one_batch(self, inputs, phase, as_tensors=False):
        """-> dict of python floats (the reference detaches every stat, ref:lib/trainer.py:306-316); as_tensors: the
        statistics as they are (0-dim device tensors mostly), for a caller that reads them back in one copy."""
        assert phase in ("train", "val", "test")
        train = phase == "train"
        self.model.train(train)
        with torch.set_grad_enabled(train):
            ahead = self._prepare_ahead(inputs)
            if train:
                # the network's forward + backward in C++ (one autograd node); the op-by-op autograd composition of
                # pcrcg_amd/train_forward.py remains as its mirror (use_cpp_runner = False, or a configuration the runner
                # does not cover)
                tr = self.model.train_runner() if self.use_cpp_runner else None
                output = tr.forward(inputs) if tr is not None else forward_train(self.model, inputs)
            else:
                output = self.model(inputs)
            prepared = ahead() if ahead is not None else None
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
            res = self.desc_loss(loss_input, prepared=prepared) if prepared is not None else self.desc_loss(loss_input)
            if train:
                c_loss = sum(res[k] for k in res if k in LOSS_KEYS)
                self.bucket.arm((self._iter + 1) % self.iter_size == 0)   # exchange overlaps the step's last backward
                c_loss.backward()               # accumulates into the flat bucket (iter_size > 1 sums pairs)
                res["total_loss"] = c_loss
        if as_tensors:
            return res
        return {k: float(v.detach()) if isinstance(v, torch.Tensor) else float(v) for k, v in res.items()}

    @staticmethod
    def _read_back(res, extra=None):
        """The statistics as python floats with ONE device-to-host copy (each float(tensor) is a copy and a wait of its own:
        a dozen of them at the end of a step, while the GPU has nothing left to do).  extra: further 0-dim tensors to bring
        along, returned as a list of floats behind the dict."""
        keys = [k for k, v in res.items() if isinstance(v, torch.Tensor)]
        tensors = [res[k].detach().reshape(()).to(torch.float32) for k in keys] + \
                  [e.detach().reshape(()).to(torch.float32) for e in (extra or [])]
        vals = torch.stack(tensors).tolist() if tensors else []
        out = {k: float(v) for k, v in res.items() if not isinstance(v, torch.Tensor)}
        out.update(zip(keys, vals[:len(keys)]))
        return {k: out[k] for k in res}, vals[len(keys):]

    def _prepare_ahead(self, inputs):
        """The loss's geometry-only part (MetricLoss.prepare: which points overlap, the max_points draw, the coordinate
        distances) on a second stream, beside the network's forward: its data-dependent shapes cost host round trips, which
        then wait for a handful of small kernels instead of for the forward.  -> a callable that runs it (after the
        forward has been enqueued) and returns its result, made safe to use on the current stream; None when the loss
        has no prepare() or the inputs are not on a GPU."""
        if not hasattr(self.desc_loss, "prepare") or not inputs["src_pcd_raw"].is_cuda:
            return None
        main = torch.cuda.current_stream()
        if getattr(self, "_prep_stream", None) is None:
            self._prep_stream = torch.cuda.Stream()
        side = self._prep_stream
        ready = torch.cuda.Event()
        ready.record(main)                      # the inputs are complete here; the forward is enqueued after this point

        def run():
            with torch.cuda.stream(side), torch.no_grad():
                side.wait_event(ready)
                prepared = self.desc_loss.prepare({k: inputs[k] for k in ("rot", "trans", "src_pcd_raw", "tgt_pcd_raw",
                                                                         "correspondences")})
            main.wait_stream(side)
            for v in prepared.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(main)
            return prepared
        return run

    # ---- optimisation block -------------------------------------------------------------------------
    def gradient_valid(self):
        """validate_gradient (ref:lib/utils.py:100-111) on the reduced bucket: same answer on every rank."""
        return self.bucket.finite()

    def optimizer_step(self):
        self.bucket.all_reduce_mean()
        return self._apply_step(self.gradient_valid())

    def _apply_step(self, ok):
        """The optimiser step behind the (already reduced and checked) gradients: SGD when they are finite, else skip."""
        if ok and self.flat_param is not None:
            self.optimizer.step(zero_grad=True)          # the launch clears the bucket as well
            self._bump_versions()
            return ok
        if ok:
            self.optimizer.step()
        else:
            self.skipped_steps += 1
        self.bucket.zero()
        return ok

    def _bump_versions(self):
        """The flat step changed every parameter under its view: tell the caches that are keyed by tensor versions (the
        inference runner's weight signature, KPFCNN's temperature cache)."""
        r = getattr(self.model, "_runner", None)
        if r is not None and hasattr(r, "invalidate"):
            r.invalidate()
        if hasattr(self.model, "_eps_cache"):
            self.model._eps_cache = None

    def train_step(self, inputs):
        """One iteration of the reference's epoch loop for phase 'train' (ref:lib/trainer.py:340-361)."""
        res = self.inference_one_batch(inputs, "train", as_tensors=True)
        self._iter += 1
        if self._iter % self.iter_size != 0:
            return self._read_back(res)[0]
        # the exchange is finished and the finite-check enqueued BEFORE anything is read back: statistics and the check's
        # answer come to the host in one copy, one wait for the GPU instead of thirteen
        self.bucket.all_reduce_mean()
        stats, (finite,) = self._read_back(res, [self.bucket.finite_tensor()])
        stats["gradient_valid"] = float(self._apply_step(finite > 0.5))
        return stats

    def end_epoch(self):
        self.scheduler.step()
