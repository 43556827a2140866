"""Differentiable versions of the HIP ops: torch.autograd.Function wrappers whose
forward AND backward run in libpcrcg_hip.so (include/pcrcg.h, include/pcrcg_train.h).  torch.autograd only
records the graph; the gradient formulas are the ones of SURVEY.md appendix C:

  matmul        y = (a @ b) * row_scale + bias    da = (dy * rs) @ b^T,  db = a^T @ (dy * rs),  dbias = sum_m dy
  kpconv        y = (wf @ W) / n_q                dW = wf^T @ (dy/n),  d wf = (dy/n) @ W^T,  dx = scatter(w^T d wf)
  instnorm      y = lrelu((x - mean) * rstd)      dx = rstd * (g - mean(g) - xhat * mean(g * xhat))
  max_pool      y = max_h x[idx]                  dx[arg max] += dy
  closest_pool  y = x[idx[:, 0]]                  dx[idx[:, 0]] += dy
  softmax_rows  p = softmax(s * scale)            ds = scale * p * (dp - sum p * dp)
  edge_conv     y = lrelu(IN2d(max_j(ctr_i + nbr_idx[i,j])), 0.2)   (DGCNN edge conv of ref:models/gcn.py:37-64,121-129)

Geometry (points, neighbour tables, kernel points) never receives a gradient (rigid KPConv)."""
import torch

from . import _lib, ops

_F32 = torch.float32


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _Matmul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, row_scale, bias):
        ctx.save_for_backward(a, b, row_scale)
        ctx.has_bias = bias is not None
        return ops.gemm(a, b, row_scale=row_scale, bias=bias)

    @staticmethod
    def backward(ctx, dy):
        a, b, rs = ctx.saved_tensors
        dy = _c(dy)
        da = db = dbias = None
        if ctx.needs_input_grad[0]:
            da = ops.gemm(dy, b.t(), row_scale=rs, grad_operand=1)
        if ctx.needs_input_grad[1]:
            dys = dy if rs is None else dy * rs[:, None]
            db = ops.gemm(a.t(), dys, grad_operand=2)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            ones = torch.ones((1, dy.shape[0]), dtype=_F32, device=dy.device)
            dbias = ops.gemm(ones, dy, grad_operand=2).reshape(-1)
        return da, db, None, dbias


def matmul(a, b, row_scale=None, bias=None):
    """(a [m,k] @ b [k,n]) * row_scale[m] + bias[n]; a and b may be transposed views (consumed in place)."""
    return _Matmul.apply(a, b, row_scale, bias)


def linear(x, weight, bias=None):
    """nn.Linear / 1x1 convolution on row-major features: x [N, Cin] @ weight[Cout, Cin]^T (+ bias)."""
    return _Matmul.apply(x, weight.t(), None, bias)


class _KPConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weights, q_pts, s_pts, idx, kp, extent):
        L = _lib.lib()
        x = _c(x)
        q_pts, s_pts, kp = _c(q_pts), _c(s_pts), _c(kp)
        idx2, ld_idx = ops._rows(idx, torch.int64, "neighb_inds")
        nq, h = idx2.shape
        ns, cin = x.shape
        kdim = kp.shape[0]
        wf = torch.empty((nq, kdim * cin), dtype=_F32, device=x.device)
        inv_n = torch.empty(nq, dtype=_F32, device=x.device)
        nbytes = L.pcrcg_kpconv_ws_bytes(ns)
        ws = ops._ws.get("kpconv", nbytes, x.device)
        _lib.check(L.pcrcg_kpconv_aggregate(q_pts.data_ptr(), nq, s_pts.data_ptr(), ns, idx2.data_ptr(), h, ld_idx,
                                            x.data_ptr(), cin, kp.data_ptr(), float(extent), wf.data_ptr(),
                                            inv_n.data_ptr(), ws.data_ptr(), nbytes, ops._stream()),
                   "pcrcg_kpconv_aggregate")
        w2 = weights.reshape(kdim * cin, -1)
        ctx.save_for_backward(wf, inv_n, w2, q_pts, s_pts, idx2, kp)
        ctx.meta = (float(extent), ns, cin, ld_idx, tuple(weights.shape))
        return ops.gemm(wf, w2, row_scale=inv_n)

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        wf, inv_n, w2, q_pts, s_pts, idx2, kp = ctx.saved_tensors
        extent, ns, cin, ld_idx, wshape = ctx.meta
        dy = _c(dy)
        dx = dw = None
        if ctx.needs_input_grad[1]:
            dw = ops.gemm(wf.t(), dy * inv_n[:, None], grad_operand=2).reshape(wshape)
        if ctx.needs_input_grad[0]:
            d_wf = ops.gemm(dy, w2.t(), row_scale=inv_n, grad_operand=1)               # [nq, 15*cin]
            dx = torch.zeros((ns, cin), dtype=_F32, device=dy.device)
            nq, h = idx2.shape
            _lib.check(L.pcrcg_kpconv_backward_dx(q_pts.data_ptr(), nq, s_pts.data_ptr(), ns, idx2.data_ptr(), h, ld_idx,
                                                  d_wf.data_ptr(), cin, kp.data_ptr(), extent, dx.data_ptr(),
                                                  ops._stream()), "pcrcg_kpconv_backward_dx")
        return dx, dw, None, None, None, None, None


def kpconv(x, weights, q_pts, s_pts, idx, kernel_points, extent):
    """KPConv.forward (ref:models/blocks.py:229-374), differentiable in x and weights [15, cin, cout]."""
    return _KPConv.apply(x, weights, q_pts, s_pts, idx, kernel_points, extent)


class _InstNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope, eps):
        x = _c(x)
        stats = ops.instnorm_stats(x, eps)
        ctx.save_for_backward(x, stats)
        ctx.slope = float(slope)
        return ops.instnorm_apply(x, stats, slope)

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, stats = ctx.saved_tensors
        dy = _c(dy)
        n, c = x.shape
        dx = torch.empty_like(x)
        nbytes = L.pcrcg_instnorm_backward_ws_bytes(c)
        ws = ops._ws.get("instnorm_bwd", nbytes, x.device)
        _lib.check(L.pcrcg_instnorm_backward(x.data_ptr(), n, c, c, stats.data_ptr(), dy.data_ptr(), c, ctx.slope,
                                             dx.data_ptr(), c, ws.data_ptr(), nbytes, ops._stream()),
                   "pcrcg_instnorm_backward")
        return dx, None, None


def instnorm_lrelu(x, slope=1.0, eps=1e-5):
    """InstanceNorm over the rows (no affine, no running stats) + LeakyReLU(slope); slope 1.0 = identity."""
    return _InstNorm.apply(x, slope, eps)


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        x = _c(x)
        y = ops.gather_max(x, idx)
        ctx.save_for_backward(x, idx, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, idx, y = ctx.saved_tensors
        dy = _c(dy)
        idx2, ld_idx = ops._rows(idx, torch.int64, "inds")
        ns, c = x.shape
        nq, h = idx2.shape
        dx = torch.zeros_like(x)
        _lib.check(L.pcrcg_gather_max_backward(x.data_ptr(), ns, c, idx2.data_ptr(), nq, h, ld_idx, y.data_ptr(),
                                               dy.data_ptr(), dx.data_ptr(), ops._stream()), "pcrcg_gather_max_backward")
        return dx, None


def max_pool(x, inds):
    """ref:models/blocks.py:86-102."""
    return _MaxPool.apply(x, inds)


class _ClosestPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        x = _c(x)
        ctx.save_for_backward(idx)
        ctx.shape = tuple(x.shape)
        return ops.gather_first(x, idx)

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        (idx,) = ctx.saved_tensors
        dy = _c(dy)
        idx2, ld_idx = ops._rows(idx, torch.int64, "inds")
        ns, c = ctx.shape
        dx = torch.zeros(ctx.shape, dtype=_F32, device=dy.device)
        _lib.check(L.pcrcg_gather_first_backward(dy.data_ptr(), c, c, idx2.data_ptr(), idx2.shape[0], ld_idx, ns,
                                                 dx.data_ptr(), ops._stream()), "pcrcg_gather_first_backward")
        return dx, None


def closest_pool(x, inds):
    """ref:models/blocks.py:71-83."""
    return _ClosestPool.apply(x, inds)


class _SoftmaxRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, scale):
        p = ops.softmax_rows_(s.clone().contiguous(), scale)
        ctx.save_for_backward(p)
        ctx.scale = float(scale)
        return p

    @staticmethod
    def backward(ctx, dp):
        L = _lib.lib()
        (p,) = ctx.saved_tensors
        dp = _c(dp)
        rows, cols = p.shape
        ds = torch.empty_like(p)
        _lib.check(L.pcrcg_softmax_rows_backward(p.data_ptr(), cols, dp.data_ptr(), cols, rows, cols, ctx.scale,
                                                 ds.data_ptr(), cols, ops._stream()), "pcrcg_softmax_rows_backward")
        return ds, None


def softmax_rows(s, scale=1.0):
    """softmax(s * scale) over the last dimension of a 2-D tensor."""
    return _SoftmaxRows.apply(s, scale)


class _EdgeConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ctr, nbr, idx, slope, eps):
        ctr, nbr = _c(ctr), _c(nbr)
        emax, stats = ops.edgeconv_reduce(ctr, nbr, idx, eps)
        ctx.save_for_backward(ctr, nbr, idx, stats)
        ctx.slope = float(slope)
        return ops.instnorm_apply(emax, stats, slope)

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        ctr, nbr, idx, stats = ctx.saved_tensors
        dy = _c(dy)
        n, c = ctr.shape
        k = idx.shape[1]
        dctr = torch.empty_like(ctr)
        dnbr = torch.zeros_like(nbr)
        nbytes = L.pcrcg_edgeconv_backward_ws_bytes(c)
        ws = ops._ws.get("edgeconv_bwd", nbytes, ctr.device)
        _lib.check(L.pcrcg_edgeconv_backward(ctr.data_ptr(), nbr.data_ptr(), idx.data_ptr(), n, k, c, stats.data_ptr(),
                                             dy.data_ptr(), ctx.slope, dctr.data_ptr(), dnbr.data_ptr(), ws.data_ptr(),
                                             nbytes, ops._stream()), "pcrcg_edgeconv_backward")
        return dctr, dnbr, None, None, None


def edge_conv(ctr, nbr, idx, slope=0.2, eps=1e-5):
    """max_j LeakyReLU(InstanceNorm2d(ctr_i + nbr_idx[i,j])) with the statistics taken over all N*k edges."""
    return _EdgeConv.apply(ctr, nbr, idx, slope, eps)
