"""Python side of the whole-network runner (pcrcg_kpfcnn_forward in include/pcrcg.h): builds the C
descriptors (device pointers + sizes) from a pcrcg_amd.architectures.KPFCNN module and a batch dict,
and enqueues the complete forward pass with ONE call into libpcrcg_hip.so.

Weights are used in place as stored in the (reference-compatible) state_dict.  Three re-packed copies
are made once per weight version and cached:
  * DGCNN edge convs: [Cout, 2Cin] -> [2Cout, Cin] = [Wa-Wb ; Wb]   (centre term rows, then neighbour term rows;
    k-contiguous like every other weight: the C = A @ B^T form the split-bf16 GEMM is built for)
  * attention projections / merge: channels permuted head-major (the reference interleaves heads,
    ref:models/gcn.py:170)
  * decoder unary weights with 1538 / 769 input channels: rows padded to a multiple of 4 floats."""
import ctypes
import threading

import torch

from . import _lib
from .blocks import (LastUnaryBlock, NearestUpsampleBlock, ResnetBottleneckBlock, SimpleBlock, UnaryBlock,
                     aligned_weight)
from .gcn import AttentionalPropagation, SelfAttention

MAX_LEVELS, MAX_BLOCKS, MAX_GNN = 8, 32, 8
BLK_SIMPLE, BLK_RESNETB, BLK_UNARY, BLK_LAST_UNARY, BLK_UPSAMPLE = range(5)
_fp = ctypes.c_void_p


class Block(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("layer", ctypes.c_int), ("strided", ctypes.c_int),
                ("in_dim", ctypes.c_int), ("out_dim", ctypes.c_int), ("mid_dim", ctypes.c_int),
                ("extent", ctypes.c_float), ("kp", _fp), ("kp_w", _fp), ("kp_wt", _fp), ("kp_w_pad", _fp),
                ("cin_pad", ctypes.c_int), ("unary1", _fp), ("unary2", _fp),
                ("shortcut", _fp), ("mlp", _fp), ("mlp_ld", ctypes.c_int),
                ("mlp_skip", _fp), ("mlp_skip_ld", ctypes.c_int), ("skip_dim", ctypes.c_int)]


class GnnLayer(ctypes.Structure):
    _fields_ = [("cross", ctypes.c_int), ("edge1", _fp), ("edge2", _fp), ("conv3", _fp),
                ("wq", _fp), ("bq", _fp), ("wk", _fp), ("bk", _fp), ("wv", _fp), ("bv", _fp),
                ("wm", _fp), ("bm", _fp), ("w0", _fp), ("b0", _fp), ("w3", _fp), ("b3", _fp)]


class Model(ctypes.Structure):
    _fields_ = [("n_enc", ctypes.c_int), ("n_dec", ctypes.c_int), ("n_gnn", ctypes.c_int),
                ("enc", Block * MAX_BLOCKS), ("dec", Block * MAX_BLOCKS), ("gnn", GnnLayer * MAX_GNN),
                ("enc_skip", ctypes.c_int * MAX_BLOCKS), ("dec_concat", ctypes.c_int * MAX_BLOCKS),
                ("enc_out_dim", ctypes.c_int), ("gnn_dim", ctypes.c_int), ("heads", ctypes.c_int),
                ("knn_k", ctypes.c_int), ("final_dim", ctypes.c_int),
                ("bottle_w", _fp), ("bottle_b", _fp), ("proj_gnn_w", _fp), ("proj_gnn_b", _fp),
                ("proj_score_w", _fp), ("proj_score_b", _fp), ("temperature", ctypes.c_float),
                ("feature_bf16", ctypes.c_int)]


class Table(ctypes.Structure):
    _fields_ = [("idx", _fp), ("rows", ctypes.c_int), ("cols", ctypes.c_int), ("ld", ctypes.c_int)]


class Batch(ctypes.Structure):
    _fields_ = [("n_levels", ctypes.c_int), ("points", _fp * MAX_LEVELS), ("n_points", ctypes.c_int * MAX_LEVELS),
                ("neighbors", Table * MAX_LEVELS), ("pools", Table * MAX_LEVELS), ("upsamples", Table * MAX_LEVELS),
                ("features", _fp), ("feat_dim", ctypes.c_int), ("len_src_c", ctypes.c_int),
                ("stack_lengths", _fp * MAX_LEVELS)]


class PyramidCfg(ctypes.Structure):
    """pcrcg_pyramid_cfg (include/pcrcg.h)."""
    _fields_ = [("n_levels", ctypes.c_int), ("r_conv", ctypes.c_float * MAX_LEVELS), ("r_pool", ctypes.c_float * MAX_LEVELS),
                ("dl", ctypes.c_float * MAX_LEVELS), ("has_conv", ctypes.c_int * MAX_LEVELS),
                ("pooled", ctypes.c_int * MAX_LEVELS), ("limit", ctypes.c_int * MAX_LEVELS), ("tie_order", ctypes.c_int),
                ("group", ctypes.c_int), ("up_nearest", ctypes.c_int), ("shrink", ctypes.c_double), ("side_stream", _fp),
                ("side_stream2", _fp)]


from .ops import ReorderJob as ReorderJobC   # pcrcg_reorder_job (include/pcrcg.h): ONE mirror of the struct


class PyramidRestore(ctypes.Structure):
    """pcrcg_pyramid_restore (include/pcrcg.h)."""
    _fields_ = [("njobs", ctypes.c_int), ("jobs", ReorderJobC * 12), ("tie_status", _fp)]


class Outputs(ctypes.Structure):
    _fields_ = [("feats_f", _fp), ("scores_overlap", _fp), ("scores_saliency", _fp)]


_bound = False


def _bind():
    global _bound
    L = _lib.lib()
    if not _bound:
        _bound = True   # signatures are declared in _lib.SIGNATURES (struct pointers as void*)
    return L


def _dense(t):
    t = t.detach()
    if t.dtype != torch.float32 or not t.is_cuda:
        raise RuntimeError("pcrcg_amd.runner: parameters must be float32 tensors on a HIP device")
    return t.contiguous()


class Runner:
    """Descriptor cache for one KPFCNN module."""

    def __init__(self, model):
        self.model = model
        self.sig = None
        self.desc = None
        self.keep = []      # tensors the descriptor points into
        self.ws = {}
        self._lock = threading.Lock()   # descriptor() is called from one forward-worker thread per model stream
        self._params = None             # the module tree's parameters, listed once (see _signature)
        self._built = None              # event behind the re-packing kernels of the current descriptor
        self._seen = set()              # launch streams that have been ordered behind _built
        self._retired = []              # [(tensors of an older descriptor, events after which they may go)]

    # a copied / pickled model starts without a runner (KPFCNN.runner() re-creates it lazily)
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())

    def _signature(self):
        """Weight version: (storage address, in-place version counter) of every parameter.  Walking the module tree costs
        ~1 ms of interpreter time, and every launch of every model thread asks -- so the parameter OBJECTS are listed
        once; optimiser steps, load_state_dict and .to() all keep the objects and change what is compared here.
        A caller that replaces a Parameter object calls invalidate()."""
        if self._params is None:
            self._params = list(self.model.parameters())
        return (bool(getattr(self.model, "feature_bf16", False)),) + tuple(
            (p.data_ptr(), p._version) for p in self._params)

    def invalidate(self):
        """Forget the cached parameter list and descriptor (after replacing Parameter objects of the module)."""
        with self._lock:
            self._params, self.sig = None, None

    def _w(self, t):
        t = _dense(t)
        self.keep.append(t)
        return t.data_ptr()

    def _kp_weights(self, blk, kp):
        """kp_w: the weights as stored, [15, cin, cout].  kp_w_pad: input channels zero-padded to a multiple of 4
        when cin is not one (the 129-channel PCR-CG input), else NULL.  kp_wt: K-contiguous copy
        [cout, 15*cin_eff] of whichever of the two the gather kernel's output matches -- the contraction then is
        a C = A @ B^T product with both operands k-contiguous.  cin = 1: [cout, 16], the 16th column zero."""
        cin = kp.in_channels
        blk.kp_w = self._w(kp.weights.data)
        w = kp.weights.data
        if cin % 4 == 0 or cin < 4:
            blk.kp_w_pad, blk.cin_pad = None, 0
        else:
            cp = (cin + 3) // 4 * 4
            w = torch.zeros((kp.weights.shape[0], cp, kp.out_channels), dtype=torch.float32, device=kp.weights.device)
            w[:, :cin].copy_(kp.weights.data)
            blk.kp_w_pad, blk.cin_pad = self._w(w), cp
        k = w.shape[0] * w.shape[1]
        if cin == 1:
            # the first layer of the geometry-only configurations: rows of 16 floats (15 kernel points + a zero), so that its
            # contraction is the grouped A B^T product with the statistics in the epilogue like every other layer's
            w16 = torch.zeros((16, kp.out_channels), dtype=torch.float32, device=w.device)
            w16[:15].copy_(w.reshape(15, kp.out_channels))
            blk.kp_wt = self._w(w16.t())
        else:
            blk.kp_wt = self._w(w.reshape(k, kp.out_channels).t()) if k % 4 == 0 else None

    def _fill_block(self, blk, mod):
        if isinstance(mod, SimpleBlock):
            blk.type, blk.layer, blk.strided = BLK_SIMPLE, mod.layer_ind, int("strided" in mod.block_name)
            kp = mod.KPConv
            blk.in_dim, blk.out_dim, blk.mid_dim = kp.in_channels, kp.out_channels, kp.out_channels
            blk.extent = float(kp.KP_extent)
            blk.kp = self._w(kp.kernel_points.data)
            self._kp_weights(blk, kp)
        elif isinstance(mod, ResnetBottleneckBlock):
            blk.type, blk.layer, blk.strided = BLK_RESNETB, mod.layer_ind, int("strided" in mod.block_name)
            kp = mod.KPConv
            blk.in_dim, blk.out_dim, blk.mid_dim = mod.in_dim, mod.out_dim, kp.out_channels
            blk.extent = float(kp.KP_extent)
            blk.kp = self._w(kp.kernel_points.data)
            self._kp_weights(blk, kp)
            blk.unary1 = self._w(mod.unary1.mlp.weight.data) if isinstance(mod.unary1, UnaryBlock) else None
            blk.unary2 = self._w(mod.unary2.mlp.weight.data)
            blk.shortcut = (self._w(mod.unary_shortcut.mlp.weight.data)
                            if isinstance(mod.unary_shortcut, UnaryBlock) else None)
        elif isinstance(mod, (UnaryBlock, LastUnaryBlock)):
            blk.type = BLK_UNARY if isinstance(mod, UnaryBlock) else BLK_LAST_UNARY
            blk.in_dim, blk.out_dim = mod.in_dim, mod.out_dim
            w = aligned_weight(mod.mlp.weight)          # [out, in] view, rows 16-byte aligned
            self.keep.append(w)
            blk.mlp, blk.mlp_ld = w.data_ptr(), (w.stride(0) if w.shape[0] > 1 else w.shape[1])
        elif isinstance(mod, NearestUpsampleBlock):
            blk.type, blk.layer = BLK_UPSAMPLE, mod.layer_ind
        else:
            raise RuntimeError(f"pcrcg_amd.runner: unsupported block {type(mod).__name__}")
        if not getattr(mod, "use_bn", True):
            raise RuntimeError("pcrcg_amd.runner: use_batch_norm=False is handled by the op-by-op path only")

    def _fill_gnn(self, g, layer, heads):
        if isinstance(layer, SelfAttention):
            g.cross = 0

            def pack(conv):
                w = conv.weight.data.flatten(1)
                cin = w.shape[1] // 2
                wa, wb = w[:, :cin], w[:, cin:]
                return self._w(torch.cat([wa - wb, wb], 0))      # [2Cout, Cin]: rows = output channels (centre | neighbour)
            g.edge1, g.edge2 = pack(layer.conv1), pack(layer.conv2)
            g.conv3 = self._w(layer.conv3.weight.data.flatten(1))
        elif isinstance(layer, AttentionalPropagation):
            g.cross = 1
            att = layer.attn
            h, d = att.num_heads, att.dim
            dev = att.merge.weight.device
            perm = (torch.arange(h, device=dev)[:, None] + h * torch.arange(d, device=dev)[None, :]).reshape(-1)
            g.wq = self._w(att.proj[0].weight.data.squeeze(-1)[perm])
            g.bq = self._w(att.proj[0].bias.data[perm])
            # key and value projections behind each other ([2 ch, ch] and [2 ch]): csrc/runner.hip then runs them as ONE
            # product of width 2 ch over the rows they share (wv = wk + ch * ch, bv = bk + ch is what it looks for)
            wkv = torch.cat([att.proj[1].weight.data.squeeze(-1)[perm], att.proj[2].weight.data.squeeze(-1)[perm]], 0).contiguous()
            bkv = torch.cat([att.proj[1].bias.data[perm], att.proj[2].bias.data[perm]], 0).contiguous()
            ch = wkv.shape[1]
            g.wk, g.bk = self._w(wkv), self._w(bkv)
            g.wv, g.bv = g.wk + 4 * ch * ch, g.bk + 4 * ch
            g.wm = self._w(att.merge.weight.data.squeeze(-1)[:, perm])
            g.bm = self._w(att.merge.bias.data)
            g.w0, g.b0 = self._w(layer.mlp[0].weight.data.squeeze(-1)), self._w(layer.mlp[0].bias.data)
            g.w3, g.b3 = self._w(layer.mlp[3].weight.data.squeeze(-1)), self._w(layer.mlp[3].bias.data)
        else:
            raise RuntimeError(f"pcrcg_amd.runner: unsupported GNN layer {type(layer).__name__}")

    def descriptor(self):
        """The model descriptor for the current weight version.  Thread-safe: built under a lock into a fresh `keep`
        list and published together with it, so a concurrent caller never sees (or frees) a half-built one."""
        with self._lock:
            sig = self._signature()
            if self.desc is not None and sig == self.sig:
                return self.desc
            old_keep, self.keep = self.keep, []
            try:
                d = self._build()
            except BaseException:
                self.keep = old_keep
                raise
            # the re-packing copies / split kernels above ran on the CURRENT stream; forwards are enqueued on other
            # (non-blocking) streams: each of them waits for this event once (launch())
            self._built = torch.cuda.Event()
            self._built.record()
            self._seen = set()
            # the previous version's re-packed copies stay alive until every stream that launched forwards with them has
            # passed the point it is at now
            if old_keep:
                evs = [torch.cuda.ExternalStream(st, device=torch.device("cuda", di)).record_event() for (di, st) in self.ws]
                self._retired.append((old_keep, evs))
            self._retired = [(k, evs) for (k, evs) in self._retired if not all(e.query() for e in evs)]
            self.desc, self.sig = d, sig
            return d

    def _build(self):
        m = self.model
        d = Model()
        d.n_enc, d.n_dec, d.n_gnn = len(m.encoder_blocks), len(m.decoder_blocks), len(m.gnn.layers)
        if d.n_enc > MAX_BLOCKS or d.n_dec > MAX_BLOCKS or d.n_gnn > MAX_GNN:
            raise RuntimeError("pcrcg_amd.runner: architecture too deep for the descriptor")
        for i, mod in enumerate(m.encoder_blocks):
            self._fill_block(d.enc[i], mod)
            d.enc_skip[i] = int(i in m.encoder_skips)
        cur_dim = m.bottle.out_channels + 2                     # width entering the decoder (:538-565)
        for j, mod in enumerate(m.decoder_blocks):
            self._fill_block(d.dec[j], mod)
            d.dec_concat[j] = int(j in m.decoder_concats)
            if isinstance(mod, (UnaryBlock, LastUnaryBlock)):
                if j in m.decoder_concats and mod.in_dim > cur_dim:
                    # the skip columns of the weight as their own aligned matrix (fused upsample + concat in the runner)
                    cs = mod.in_dim - cur_dim
                    ws = mod.mlp.weight.data[:, cur_dim:]
                    pad = (-cs) % 4
                    ws = torch.nn.functional.pad(ws, (0, pad)).contiguous() if pad else ws.contiguous()
                    self.keep.append(ws)
                    d.dec[j].mlp_skip, d.dec[j].mlp_skip_ld, d.dec[j].skip_dim = ws.data_ptr(), cs + pad, cs
                cur_dim = mod.out_dim
        heads = None
        for i, layer in enumerate(m.gnn.layers):
            self._fill_gnn(d.gnn[i], layer, heads)
            if isinstance(layer, AttentionalPropagation):
                heads = layer.attn.num_heads
            else:
                d.knn_k = layer.k
        d.heads = heads or 1
        d.enc_out_dim, d.gnn_dim, d.final_dim = m.bottle.in_channels, m.bottle.out_channels, m.final_feats_dim
        d.bottle_w, d.bottle_b = self._w(m.bottle.weight.data.squeeze(-1)), self._w(m.bottle.bias.data)
        d.proj_gnn_w, d.proj_gnn_b = self._w(m.proj_gnn.weight.data.squeeze(-1)), self._w(m.proj_gnn.bias.data)
        d.proj_score_w, d.proj_score_b = (self._w(m.proj_score.weight.data.squeeze(-1)),
                                          self._w(m.proj_score.bias.data))
        d.temperature = float(torch.exp(m.epsilon.detach()).item()) + 0.03
        # the bf16 feature-storage VARIANT (include/pcrcg.h pcrcg_model.feature_bf16); off unless asked for
        d.feature_bf16 = int(bool(getattr(m, "feature_bf16", False)))
        return d

    @staticmethod
    def _table(dst, t, keep):
        if t.numel() == 0:
            dst.idx, dst.rows, dst.cols, dst.ld = None, int(t.shape[0]), 0, 1
            return
        if t.dtype != torch.int64 or not t.is_cuda:
            raise RuntimeError("pcrcg_amd.runner: index tables must be int64 tensors on the device")
        if t.shape[1] > 1 and t.stride(1) != 1:
            t = t.contiguous()
        keep.append(t)
        dst.idx, dst.rows, dst.cols = t.data_ptr(), int(t.shape[0]), int(t.shape[1])
        dst.ld = int(t.stride(0)) if t.shape[0] > 1 else int(t.shape[1])

    def batch_struct(self, batch):
        """The reference's batch dict -> (pcrcg_batch mirror, tensors it points into, device)."""
        keep = []
        b = Batch()
        pts = batch["points"]
        b.n_levels = len(pts)
        for l, p in enumerate(pts):
            p = p.contiguous()
            if p.dtype != torch.float32 or not p.is_cuda:
                raise RuntimeError("pcrcg_amd.runner: points must be float32 tensors on the device")
            keep.append(p)
            b.points[l], b.n_points[l] = p.data_ptr(), int(p.shape[0])
            self._table(b.neighbors[l], batch["neighbors"][l], keep)
            self._table(b.pools[l], batch["pools"][l], keep)
            self._table(b.upsamples[l], batch["upsamples"][l], keep)
        feats = batch["features"].to(torch.float32).contiguous()
        keep.append(feats)
        b.features, b.feat_dim = feats.data_ptr(), int(feats.shape[1])
        if "stack_lengths_host" in batch:
            b.len_src_c = int(batch["stack_lengths_host"][-1][0])
        else:
            b.len_src_c = int(batch["stack_lengths"][-1][0])
        return b, keep, feats.device

    def forward(self, batch):
        b, keep, dev = self.batch_struct(batch)
        return self.launch(b, dev)

    def launch_group(self, batches, n, dev, start=0):
        """Enqueue ONE forward call for n (1 to 4) consecutive pcrcg_batch structs of the ctypes array `batches`, from
        element `start` (as NativePyramid.build(group=2) returns them) on the current stream -> list of n output dicts.
        Weight products run once for all pairs (pcrcg_kpfcnn_forward_group)."""
        L = _bind()
        desc = self.descriptor()
        outs, o = [], (Outputs * n)()
        first = ctypes.byref(batches[start])
        for g in range(n):
            n0 = batches[start + g].n_points[0]
            out = {"feats_f": torch.empty((n0, desc.final_dim), dtype=torch.float32, device=dev),
                   "scores_overlap": torch.empty(n0, dtype=torch.float32, device=dev),
                   "scores_saliency": torch.empty(n0, dtype=torch.float32, device=dev)}
            o[g].feats_f, o[g].scores_overlap, o[g].scores_saliency = (out["feats_f"].data_ptr(), out["scores_overlap"].data_ptr(),
                                                                      out["scores_saliency"].data_ptr())
            outs.append(out)
        nbytes = L.pcrcg_kpfcnn_group_ws_bytes(ctypes.byref(desc), first, n)
        if nbytes == 0:
            raise RuntimeError("pcrcg_kpfcnn_group_ws_bytes rejected the descriptors: " + (L.pcrcg_last_error() or b"").decode())
        cur = torch.cuda.current_stream()
        stream = cur.cuda_stream
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), stream)
        with self._lock:
            ws = self.ws.get(key)
            if ws is None or ws.numel() < nbytes:
                ws = torch.empty(int(nbytes * 1.25), dtype=torch.uint8, device=dev)
                self.ws[key] = ws
            if key not in self._seen and self._built is not None:
                cur.wait_event(self._built)
                self._seen.add(key)
        _lib.check(L.pcrcg_kpfcnn_forward_group(ctypes.byref(desc), first, o, n, ws.data_ptr(), ws.numel(), stream),
                   "pcrcg_kpfcnn_forward_group")
        return outs

    def launch(self, b, dev):
        """Enqueue the forward for a pcrcg_batch (from batch_struct, or filled by pcrcg_pyramid_build) on the
        current stream; the caller keeps whatever `b` points into alive until the stream has passed."""
        L = _bind()
        desc = self.descriptor()
        n0 = b.n_points[0]
        out = {"feats_f": torch.empty((n0, desc.final_dim), dtype=torch.float32, device=dev),
               "scores_overlap": torch.empty(n0, dtype=torch.float32, device=dev),
               "scores_saliency": torch.empty(n0, dtype=torch.float32, device=dev)}
        o = Outputs(out["feats_f"].data_ptr(), out["scores_overlap"].data_ptr(), out["scores_saliency"].data_ptr())
        nbytes = L.pcrcg_kpfcnn_ws_bytes(ctypes.byref(desc), ctypes.byref(b))
        if nbytes == 0:
            raise RuntimeError("pcrcg_kpfcnn_ws_bytes rejected the descriptors: "
                               + (L.pcrcg_last_error() or b"").decode())
        cur = torch.cuda.current_stream()
        stream = cur.cuda_stream
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), stream)
        with self._lock:
            ws = self.ws.get(key)
            if ws is None or ws.numel() < nbytes:
                ws = torch.empty(int(nbytes * 1.25), dtype=torch.uint8, device=dev)
                self.ws[key] = ws
            if key not in self._seen and self._built is not None:
                cur.wait_event(self._built)        # the descriptor's re-packed weights are complete before the first read
                self._seen.add(key)
        _lib.check(L.pcrcg_kpfcnn_forward(ctypes.byref(desc), ctypes.byref(b), ctypes.byref(o), ws.data_ptr(),
                                          ws.numel(), stream), "pcrcg_kpfcnn_forward")
        return out
