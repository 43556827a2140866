"""Python side of the C++ train-step runner (csrc/train_runner.hip; include/pcrcg_train.h: pcrcg_kpfcnn_train_*):
KPFCNN.forward with a tape and its backward, each ONE call into libpcrcg_hip.so, wrapped as a single
torch.autograd.Function so that `loss.backward()` works unchanged (ref:lib/trainer.py:216-265).

What torch still does around it: the loss (pcrcg_amd/loss.py), a handful of tiny tensor ops per step that hand derived
weight layouts to the library and fold their gradients back (the DGCNN edge convolutions' [Wa - Wb ; Wb] split, the
head-major attention projections, two decoder weights with padded rows), and the optimiser.

Parameter gradients are ACCUMULATED by the library straight into `p.grad` (pcrcg_amd/trainer.py makes those views of the
flat all-reduce bucket); the autograd graph sees one differentiable input, `net.epsilon`, whose gradient it gets back
-- so parameter hooks (the bucket's overlap of the exchange with backward) do not fire per parameter: the trainer
reduces the bucket after the backward call, which is already complete when it returns."""
import ctypes

import torch

from . import _lib
from .blocks import LastUnaryBlock, NearestUpsampleBlock, ResnetBottleneckBlock, SimpleBlock, UnaryBlock
from .gcn import AttentionalPropagation, SelfAttention
from .runner import (BLK_LAST_UNARY, BLK_RESNETB, BLK_SIMPLE, BLK_UNARY, BLK_UPSAMPLE, MAX_BLOCKS, MAX_GNN, Model, Runner)


class TrainOutputs(ctypes.Structure):
    """pcrcg_train_outputs (include/pcrcg_train.h)."""
    _fields_ = [("feats_f", ctypes.c_void_p), ("scores_overlap", ctypes.c_void_p), ("scores_saliency", ctypes.c_void_p),
                ("d_inv_temperature", ctypes.c_void_p), ("n_points", ctypes.c_int), ("final_dim", ctypes.c_int)]


class GatherJob(ctypes.Structure):
    """pcrcg_gather_job (include/pcrcg_train.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("m1", ctypes.c_void_p), ("m2", ctypes.c_void_p),
                ("n", ctypes.c_int), ("s2", ctypes.c_float), ("accumulate", ctypes.c_int), ("cols", ctypes.c_int)]


class _Tape:
    """Owner of one C++ tape: frees it exactly once (after the backward, or when the autograd node is dropped without
    one) and tells the runner that its cached workspace is free again."""

    def __init__(self, handle, runner, ws):
        self.handle, self.runner, self.ws = handle, runner, ws

    def release(self):
        if self.handle is not None:
            _lib.lib().pcrcg_kpfcnn_train_free(self.handle)
            self.handle = None
            if self.runner._lent is self.ws:
                self.runner._lent = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def _grad(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


class TrainRunner:
    def __init__(self, model):
        self.model = model
        self.ws = None                            # cached workspace (values | gradients | scratch of one step)
        self._lent = None                         # ... while a tape that lives in it is outstanding
        self._batch_helper = Runner(model)        # batch dict -> pcrcg_batch

    # a copied / pickled model starts without a runner (KPFCNN.train_runner() re-creates it lazily)
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())

    # ---- descriptors ---------------------------------------------------------------------------------------------
    def _pair(self, keep, value, grad):
        """(value pointer, gradient pointer) of one weight; both tensors are kept alive."""
        if value.dtype != torch.float32 or not value.is_cuda or not value.is_contiguous():
            raise RuntimeError("pcrcg_amd.train_runner: weights must be contiguous float32 tensors on a HIP device")
        if grad is not None and not grad.is_contiguous():
            raise RuntimeError("pcrcg_amd.train_runner: parameter gradients must be contiguous")
        keep.append(value)
        keep.append(grad)
        return value.data_ptr(), (grad.data_ptr() if grad is not None else None)

    def _plan_key(self):
        """What the cached plan depends on: where every parameter and its gradient live, and which of them train."""
        key = []
        for p in self.model.parameters():
            g = _grad(p) if p.requires_grad else None
            key.append((p.data_ptr(), g.data_ptr() if g is not None else 0))
        return tuple(key)

    def _plan(self):
        """-> the step's descriptors, built once and reused while no parameter or gradient has moved: the two Model
        structs (values, gradient twins), the arena of re-packed weight layouts with the device job table that refreshes it
        from the parameters (ONE launch per step, pcrcg_gather_jobs), and the arena their gradients land in with the table
        that adds those back to the parameters' gradients (one launch after the backward)."""
        key = self._plan_key()
        if self._lent is not None:                # an earlier forward still awaits its backward: it owns the cached arenas
            return self._build_plan()
        plan = getattr(self, "_cached_plan", None)
        if plan is None or plan["key"] != key:
            plan = self._cached_plan = self._build_plan()
            plan["key"] = key
        return plan

    def _build_plan(self):
        m = self.model
        v, g = Model(), Model()
        keep, specs = [], []
        dev = m.epsilon.device
        v.n_enc, v.n_dec, v.n_gnn = len(m.encoder_blocks), len(m.decoder_blocks), len(m.gnn.layers)
        if v.n_enc > MAX_BLOCKS or v.n_dec > MAX_BLOCKS or v.n_gnn > MAX_GNN:
            raise RuntimeError("pcrcg_amd.train_runner: architecture too deep for the descriptor")

        def direct(dst_v, dst_g, name, param, view=None):
            """A parameter used as stored (optionally through a reshaping VIEW of it and of its gradient)."""
            val = param.data if view is None else view(param.data)
            if param.requires_grad:
                grd = _grad(param) if view is None else view(_grad(param))
            else:
                grd = None                          # frozen: a NULL gradient pointer, the backward skips its dW product
            pv, pg = self._pair(keep, val, grd)
            setattr(dst_v, name, pv)
            setattr(dst_g, name, pg)

        def index_of(param):
            return torch.arange(param.numel(), dtype=torch.int32, device=dev).view(param.shape)

        def derived(dst_v, dst_g, name, param, m1, m2=None, with_grad=True, transpose_cols=0):
            """A re-packed copy of `param`:  derived.flat[i] = param.flat[m1[i]] - param.flat[m2[i]]  (-1: nothing).  It
            lives in the plan's arena and follows the parameter every step; its gradient (with_grad, and the parameter
            trains) lands in the twin arena and is added back through the inverse maps.  transpose_cols: the copy is the
            plain transpose of the parameter seen as [numel / cols, cols] (value only, m1 unused)."""
            specs.append(dict(dst_v=dst_v, dst_g=dst_g, name=name, param=param, cols=int(transpose_cols),
                              n=param.numel() if transpose_cols else m1.numel(),
                              m1=None if transpose_cols else m1.reshape(-1).contiguous(),
                              m2=None if m2 is None else m2.reshape(-1).contiguous(),
                              grad=bool(with_grad and param.requires_grad and not transpose_cols)))

        def kp_block(bv, bg, kp):
            bv.extent = float(kp.KP_extent)
            keep.append(kp.kernel_points.data)
            bv.kp = kp.kernel_points.data.contiguous().data_ptr()
            if kp.in_channels != 1 and kp.in_channels % 4 != 0:
                # PCR-CG's 129-channel first layer (round 6): the runner is handed the input in rows of cp = 132 floats (three
                # zero columns, KPFCNN.IMAGE_WIDTH) and the weights as [15, cp, cout] with zero rows for them -- a derived
                # layout like the others: refreshed from the parameter every step, its gradient folded back through the
                # inverse map (the zero rows' gradients, products with zero columns, are dropped)
                kk, cin, cout = kp.weights.shape
                cp = (cin + 3) // 4 * 4
                ix = torch.full((kk, cp, cout), -1, dtype=torch.int32, device=dev)
                ix[:, :cin] = index_of(kp.weights)
                derived(bv, bg, "kp_w", kp.weights, ix)
                derived(bv, bg, "kp_wt", kp.weights, ix.reshape(kk * cp, cout).t().contiguous(), with_grad=False)
                return
            direct(bv, bg, "kp_w", kp.weights, lambda t: t.reshape(-1, t.shape[-1]))
            # a K-contiguous copy [cout, 15 cin] for the FORWARD contraction (value only: the gradient belongs to kp_w): the
            # k-contiguous product instead of the k-major one
            if (kp.weights.shape[0] * kp.weights.shape[1]) % 4 == 0:
                derived(bv, bg, "kp_wt", kp.weights, None, with_grad=False, transpose_cols=kp.weights.shape[-1])

        for i, mod in enumerate(m.encoder_blocks):
            bv, bg = v.enc[i], g.enc[i]
            v.enc_skip[i] = int(i in m.encoder_skips)
            if not getattr(mod, "use_bn", True):
                raise RuntimeError("pcrcg_amd.train_runner: use_batch_norm=False is handled by the op-by-op path only")
            if isinstance(mod, SimpleBlock):
                bv.type, bv.layer, bv.strided = BLK_SIMPLE, mod.layer_ind, int("strided" in mod.block_name)
                kp = mod.KPConv
                bv.in_dim, bv.out_dim, bv.mid_dim = kp.in_channels, kp.out_channels, kp.out_channels
                kp_block(bv, bg, kp)
            elif isinstance(mod, ResnetBottleneckBlock):
                bv.type, bv.layer, bv.strided = BLK_RESNETB, mod.layer_ind, int("strided" in mod.block_name)
                kp = mod.KPConv
                bv.in_dim, bv.out_dim, bv.mid_dim = mod.in_dim, mod.out_dim, kp.out_channels
                kp_block(bv, bg, kp)
                if isinstance(mod.unary1, UnaryBlock):
                    direct(bv, bg, "unary1", mod.unary1.mlp.weight)
                direct(bv, bg, "unary2", mod.unary2.mlp.weight)
                if isinstance(mod.unary_shortcut, UnaryBlock):
                    direct(bv, bg, "shortcut", mod.unary_shortcut.mlp.weight)
            else:
                raise RuntimeError(f"pcrcg_amd.train_runner: unsupported encoder block {type(mod).__name__}")
        for j, mod in enumerate(m.decoder_blocks):
            bv, bg = v.dec[j], g.dec[j]
            v.dec_concat[j] = int(j in m.decoder_concats)
            if isinstance(mod, (UnaryBlock, LastUnaryBlock)):
                if not getattr(mod, "use_bn", True):
                    raise RuntimeError("pcrcg_amd.train_runner: use_batch_norm=False is handled by the op-by-op path only")
                bv.type = BLK_UNARY if isinstance(mod, UnaryBlock) else BLK_LAST_UNARY
                bv.in_dim, bv.out_dim = mod.in_dim, mod.out_dim
                w = mod.mlp.weight
                k = w.shape[1]
                if k % 4 == 0:
                    direct(bv, bg, "mlp", w)
                    bv.mlp_ld = k
                else:                               # rows padded to 16 bytes (decoder widths 1538 and 769)
                    kp4 = (k + 3) // 4 * 4
                    padded = torch.full((w.shape[0], kp4), -1, dtype=torch.int32, device=dev)
                    padded[:, :k] = index_of(w)
                    derived(bv, bg, "mlp", w, padded)
                    bv.mlp_ld = kp4
            elif isinstance(mod, NearestUpsampleBlock):
                bv.type, bv.layer = BLK_UPSAMPLE, mod.layer_ind
            else:
                raise RuntimeError(f"pcrcg_amd.train_runner: unsupported decoder block {type(mod).__name__}")
        heads = None
        for i, layer in enumerate(m.gnn.layers):
            lv, lg = v.gnn[i], g.gnn[i]
            if isinstance(layer, SelfAttention):
                lv.cross = 0
                v.knn_k = layer.k
                for name, conv in (("edge1", layer.conv1), ("edge2", layer.conv2)):
                    # conv(cat(x_i, x_j - x_i)) = (Wa - Wb) x_i + Wb x_j: the packed rows [Wa - Wb ; Wb]
                    w = conv.weight
                    ix = index_of(w).flatten(1)
                    cin = ix.shape[1] // 2
                    ia, ib = ix[:, :cin], ix[:, cin:]
                    derived(lv, lg, name, w, torch.cat([ia, ib], 0), torch.cat([ib, torch.full_like(ib, -1)], 0))
                direct(lv, lg, "conv3", layer.conv3.weight, lambda t: t.flatten(1))
            elif isinstance(layer, AttentionalPropagation):
                lv.cross = 1
                att = layer.attn
                h, d = att.num_heads, att.dim
                heads = h
                perm = (torch.arange(h, device=dev)[:, None] + h * torch.arange(d, device=dev)[None, :]).reshape(-1)
                for name, proj in zip("qkv", att.proj):
                    derived(lv, lg, "w" + name, proj.weight, index_of(proj.weight).squeeze(-1)[perm])
                    derived(lv, lg, "b" + name, proj.bias, index_of(proj.bias)[perm])
                derived(lv, lg, "wm", att.merge.weight, index_of(att.merge.weight).squeeze(-1)[:, perm])
                direct(lv, lg, "bm", att.merge.bias)
                direct(lv, lg, "w0", layer.mlp[0].weight, lambda t: t.squeeze(-1))
                direct(lv, lg, "b0", layer.mlp[0].bias)
                direct(lv, lg, "w3", layer.mlp[3].weight, lambda t: t.squeeze(-1))
                direct(lv, lg, "b3", layer.mlp[3].bias)
            else:
                raise RuntimeError(f"pcrcg_amd.train_runner: unsupported GNN layer {type(layer).__name__}")
        v.heads = heads or 1
        v.enc_out_dim, v.gnn_dim, v.final_dim = m.bottle.in_channels, m.bottle.out_channels, m.final_feats_dim
        for name, conv in (("bottle", m.bottle), ("proj_gnn", m.proj_gnn), ("proj_score", m.proj_score)):
            direct(v, g, name + "_w", conv.weight, lambda t: t.squeeze(-1))
            direct(v, g, name + "_b", conv.bias)
        # ---- the arenas of the re-packed layouts and of their gradients, and the two job tables
        def rounded(n):
            return (n + 63) // 64 * 64
        total = sum(rounded(sp["n"]) for sp in specs)
        total_g = sum(rounded(sp["n"]) for sp in specs if sp["grad"])
        values = torch.zeros(max(total, 1), dtype=torch.float32, device=dev)
        grads = torch.zeros(max(total_g, 1), dtype=torch.float32, device=dev)
        derive, fold = [], []
        off = off_g = 0
        for sp in specs:
            p, n = sp["param"], sp["n"]
            if p.dtype != torch.float32 or not p.is_cuda or not p.data.is_contiguous():
                raise RuntimeError("pcrcg_amd.train_runner: weights must be contiguous float32 tensors on a HIP device")
            setattr(sp["dst_v"], sp["name"], values.data_ptr() + 4 * off)
            keep += [sp["m1"], sp["m2"], p.data]
            derive.append(GatherJob(p.data_ptr(), values.data_ptr() + 4 * off, sp["m1"].data_ptr() if sp["m1"] is not None else None,
                                    sp["m2"].data_ptr() if sp["m2"] is not None else None, n, -1.0, 0, sp["cols"]))
            off += rounded(n)
            if not sp["grad"]:
                setattr(sp["dst_g"], sp["name"], None)
                continue
            setattr(sp["dst_g"], sp["name"], grads.data_ptr() + 4 * off_g)
            # the way back: parameter element q collects the gradient of every packed element it went into
            ar = torch.arange(n, dtype=torch.int32, device=dev)
            inv = []
            for mp in (sp["m1"], sp["m2"]):
                if mp is None:
                    inv.append(None)
                    continue
                back = torch.full((p.numel(),), -1, dtype=torch.int32, device=dev)
                ok = mp >= 0
                back[mp[ok].long()] = ar[ok]
                inv.append(back)
            gp = _grad(p)
            if not gp.is_contiguous():
                raise RuntimeError("pcrcg_amd.train_runner: parameter gradients must be contiguous")
            keep += inv + [gp]
            fold.append(GatherJob(grads.data_ptr() + 4 * off_g, gp.data_ptr(), inv[0].data_ptr(),
                                  inv[1].data_ptr() if inv[1] is not None else None, p.numel(), -1.0, 1, 0))
            off_g += rounded(n)

        def table(jobs):
            if not jobs:
                return None, 0, 0
            arr = (GatherJob * len(jobs))(*jobs)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            return host.to(dev), len(jobs), max(j.n for j in jobs)
        return {"v": v, "g": g, "keep": keep, "values": values, "grads": grads, "derive": table(derive), "fold": table(fold),
                "has_grads": total_g > 0}

    def _descriptors(self):
        """-> (values Model, gradients Model, keep-alive list, plan) for this step: the cached plan with the re-packed
        layouts refreshed from the parameters (one launch), their gradient arena cleared (one more) and the temperature
        read back from epsilon."""
        plan = self._plan()
        v, g = plan["v"], plan["g"]
        stream = torch.cuda.current_stream().cuda_stream
        tab, n_jobs, max_n = plan["derive"]
        if n_jobs:
            _lib.check(_lib.lib().pcrcg_gather_jobs(tab.data_ptr(), n_jobs, max_n, stream), "pcrcg_gather_jobs")
        if plan["has_grads"]:
            plan["grads"].zero_()
        v.temperature = float(torch.exp(self.model.epsilon.detach()).item()) + 0.03
        # the gradient twin only needs its pointers; copy the integer layout so that validation sees the same model
        for f in ("n_enc", "n_dec", "n_gnn", "enc_out_dim", "gnn_dim", "heads", "knn_k", "final_dim", "temperature"):
            setattr(g, f, getattr(v, f))
        return v, g, plan["keep"], plan

    # ---- forward / backward --------------------------------------------------------------------------------------
    def forward(self, batch):
        """-> {'feats_f', 'scores_overlap', 'scores_saliency'} with a grad_fn (one autograd node for the whole network).
        The three tensors are views into the step's workspace, which the NEXT training forward reuses once this one's
        backward has run: clone what must outlive the step.  (A forward issued while an earlier one still awaits its
        backward gets a workspace of its own.)"""
        # the node's differentiable input: epsilon when it trains; otherwise a dummy that only anchors the node in the
        # graph (with epsilon frozen and no anchor the outputs would carry no grad_fn and NO parameter would get a gradient)
        eps = self.model.epsilon
        if eps.requires_grad:
            anchor = eps
        else:
            if getattr(self, "_anchor", None) is None or self._anchor.device != eps.device:
                self._anchor = torch.zeros((), dtype=torch.float32, device=eps.device, requires_grad=True)
            anchor = self._anchor
        outs = _TrainNet.apply(anchor, self, batch)
        return {"feats_f": outs[0], "scores_overlap": outs[1], "scores_saliency": outs[2]}

    def _forward(self, batch):
        L = _lib.lib()
        v, g, keep, plan = self._descriptors()
        b, bkeep, dev = self._batch_helper.batch_struct(batch)
        sizes = [ctypes.c_size_t() for _ in range(3)]
        _lib.check(L.pcrcg_kpfcnn_train_ws_bytes(ctypes.byref(v), ctypes.byref(g), ctypes.byref(b), *[ctypes.byref(s) for s in sizes]),
                   "pcrcg_kpfcnn_train_ws_bytes")
        vb, gb, sb = (int(s.value) for s in sizes)
        total = vb + gb + sb
        if self._lent is not None:                # an earlier tape still lives in the cached workspace
            ws = torch.empty(int(total), dtype=torch.uint8, device=dev)
        else:
            if self.ws is None or self.ws.numel() < total or self.ws.device != dev:
                self.ws = None
                self.ws = torch.empty(int(total * 1.05), dtype=torch.uint8, device=dev)
            ws = self._lent = self.ws
        out = TrainOutputs()
        tape = ctypes.c_void_p()
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(L.pcrcg_kpfcnn_train_forward(ctypes.byref(v), ctypes.byref(g), ctypes.byref(b), ws.data_ptr(), vb, gb, sb,
                                                ctypes.byref(out), ctypes.byref(tape), stream), "pcrcg_kpfcnn_train_forward")
        base = ws.data_ptr()

        def view(ptr, *shape):
            n = 1
            for s in shape:
                n *= s
            off = ptr - base
            return ws[off:off + 4 * n].view(torch.float32).view(*shape)
        n0, fd = out.n_points, out.final_dim
        outs = (view(out.feats_f, n0, fd), view(out.scores_overlap, n0), view(out.scores_saliency, n0))
        state = {"tape": _Tape(tape, self, ws), "keep": (keep, bkeep, v, g, b, plan), "plan": plan,
                 "d_inv_t": view(out.d_inv_temperature, 1), "inv_t": 1.0 / v.temperature, "ws": ws}
        return outs, state

    def _backward(self, state, d_f, d_so, d_ss):
        L = _lib.lib()
        tape = state["tape"]
        if tape.handle is None:
            raise RuntimeError("pcrcg_amd.train_runner: backward called twice (the tape is released after the first)")

        def ptr(t, shape):
            if t is None:
                return None
            t = t.to(torch.float32).contiguous()
            assert tuple(t.shape) == tuple(shape)
            state.setdefault("dkeep", []).append(t)
            return t.data_ptr()
        n0, fd = state["shape"]
        try:
            _lib.check(L.pcrcg_kpfcnn_train_backward(tape.handle, ptr(d_f, (n0, fd)), ptr(d_so, (n0,)), ptr(d_ss, (n0,)),
                                                     torch.cuda.current_stream().cuda_stream), "pcrcg_kpfcnn_train_backward")
        finally:
            tape.release()
        tab, n_jobs, max_n = state["plan"]["fold"]
        if n_jobs:       # the re-packed layouts' gradients, added to the parameters' (one launch)
            _lib.check(L.pcrcg_gather_jobs(tab.data_ptr(), n_jobs, max_n, torch.cuda.current_stream().cuda_stream),
                       "pcrcg_gather_jobs")
        # temperature = exp(epsilon) + 0.03 and the library reports dL/d(1/temperature)
        eps = self.model.epsilon.detach()
        return state["d_inv_t"].reshape(()) * (-(state["inv_t"] ** 2)) * torch.exp(eps)


class _TrainNet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, epsilon, runner, batch):
        outs, state = runner._forward(batch)
        state["shape"] = tuple(outs[0].shape)
        ctx.runner, ctx.state = runner, state
        return outs

    @staticmethod
    def backward(ctx, d_f, d_so, d_ss):
        d_eps = ctx.runner._backward(ctx.state, d_f, d_so, d_ss)
        eps = ctx.runner.model.epsilon
        if not eps.requires_grad:                  # the input was the anchor: its "gradient" is discarded
            return torch.zeros((), dtype=torch.float32, device=eps.device), None, None
        return d_eps.reshape(eps.shape), None, None
