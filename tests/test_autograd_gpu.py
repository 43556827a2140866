"""GPU: gradients of the differentiable HIP ops (pcrcg_amd/autograd.py, kernels behind include/pcrcg_train.h)
against torch autograd of the CPU oracle (oracle/model_ref.py) in float64.  Bar: 1e-4 relative per tensor."""
import os

import pytest
import torch

from oracle import model_ref as MR
from pcrcg_amd import autograd as AG

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


@pytest.fixture(scope="module")
def mini(golden_dir):
    col = torch.load(os.path.join(golden_dir, "collate_mini.pt"))
    return col["batch"]


def _leaf(t, dev=None, dtype=None):
    t = t.detach().clone()
    if dtype is not None:
        t = t.to(dtype)
    if dev is not None:
        t = t.to(dev)
    return t.requires_grad_(True)


@pytest.mark.parametrize("m,k,n,tb", [(300, 64, 48, 0), (1000, 33, 17, 1), (64, 512, 380, 1)])
def test_matmul_grads(cuda, m, k, n, tb):
    g = torch.Generator().manual_seed(m)
    a, rs, bias = torch.randn(m, k, generator=g), torch.rand(m, generator=g) + 0.5, torch.randn(n, generator=g)
    b = torch.randn(n, k, generator=g) if tb else torch.randn(k, n, generator=g)
    dy = torch.randn(m, n, generator=g)
    a1, b1, bias1 = _leaf(a, cuda), _leaf(b, cuda), _leaf(bias, cuda)
    y = AG.matmul(a1, b1.t() if tb else b1, row_scale=rs.to(cuda), bias=bias1)
    y.backward(dy.to(cuda))
    a0, b0, bias0 = _leaf(a, dtype=torch.float64), _leaf(b, dtype=torch.float64), _leaf(bias, dtype=torch.float64)
    y0 = (a0 @ (b0.t() if tb else b0)) * rs.double()[:, None] + bias0
    y0.backward(dy.double())
    assert rel(y, y0) < TOL
    assert rel(a1.grad, a0.grad) < TOL and rel(b1.grad, b0.grad) < TOL and rel(bias1.grad, bias0.grad) < TOL


@pytest.mark.parametrize("level,strided,cin,cout", [(0, False, 8, 16), (1, False, 64, 64), (0, True, 128, 32),
                                                     (2, False, 1, 8), (1, True, 20, 12)])
def test_kpconv_grads(cuda, mini, level, strided, cin, cout):
    from pcrcg_amd.kernel_points import load_kernels
    g = torch.Generator().manual_seed(level * 10 + cin)
    s_pts = mini["points"][level]
    q_pts = mini["points"][level + 1] if strided else s_pts
    idx = mini["pools"][level] if strided else mini["neighbors"][level]
    radius, extent = 0.0625 * 2 ** level, 0.05 * 2 ** level
    kp = torch.tensor(load_kernels(radius, 15, dimension=3, fixed="center"), dtype=torch.float32)
    x = torch.randn(s_pts.shape[0], cin, generator=g)
    w = torch.randn(15, cin, cout, generator=g) * 0.2
    dy = torch.randn(q_pts.shape[0], cout, generator=g)
    x1, w1 = _leaf(x, cuda), _leaf(w, cuda)
    y = AG.kpconv(x1, w1, q_pts.to(cuda), s_pts.to(cuda), idx.to(cuda), kp.to(cuda), extent)
    y.backward(dy.to(cuda))
    x0, w0 = _leaf(x, dtype=torch.float64), _leaf(w, dtype=torch.float64)
    y0 = MR.kpconv(q_pts.double(), s_pts.double(), idx, x0, kp.double(), w0, extent)
    y0.backward(dy.double())
    assert rel(y, y0) < TOL
    assert rel(w1.grad, w0.grad) < TOL
    assert rel(x1.grad, x0.grad) < TOL


@pytest.mark.parametrize("level,strided,cin", [(0, False, 32), (0, True, 40), (1, False, 129), (2, False, 256), (0, False, 1)])
def test_kpconv_backward_dx_matrix_core_kernel_against_the_valu_kernel(cuda, mini, level, strided, cin):
    """pcrcg_kpconv_backward_dx (include/pcrcg_train.h): the 16-neighbour MFMA tiles (round 5, bwd_mfma=1) and the
    wavefront-per-neighbour VALU kernel they replace scatter the same sums up to order, and both match the float64 oracle's
    d x of sum(wf * d_wf)."""
    from pcrcg_amd import _lib, ops
    from pcrcg_amd.kernel_points import load_kernels
    g = torch.Generator().manual_seed(level * 7 + cin)
    s_pts = mini["points"][level]
    q_pts = mini["points"][level + 1] if strided else s_pts
    idx = (mini["pools"][level] if strided else mini["neighbors"][level]).to(cuda).contiguous()
    radius, extent = 0.0625 * 2 ** level, 0.05 * 2 ** level
    kp = torch.tensor(load_kernels(radius, 15, dimension=3, fixed="center"), dtype=torch.float32)
    nq, ns, h = q_pts.shape[0], s_pts.shape[0], idx.shape[1]
    d_wf = torch.randn(nq, 15 * cin, generator=g)
    L = _lib.lib()
    q_d, s_d, g_d, kp_d = q_pts.to(cuda).contiguous(), s_pts.to(cuda).contiguous(), d_wf.to(cuda), kp.to(cuda)

    def run(spec):
        dx = torch.zeros(ns, cin, device=cuda)
        try:
            _lib.check(L.pcrcg_debug_set(spec), "pcrcg_debug_set")
            _lib.check(L.pcrcg_kpconv_backward_dx(q_d.data_ptr(), nq, s_d.data_ptr(), ns, idx.data_ptr(), h, h, g_d.data_ptr(), cin,
                                                  kp_d.data_ptr(), extent, dx.data_ptr(), ops._stream()), "pcrcg_kpconv_backward_dx")
            torch.cuda.synchronize()
        finally:
            _lib.check(L.pcrcg_debug_set(None), "pcrcg_debug_set")
        return dx
    new, old = run(b"bwd_mfma=1"), run(b"bwd_mfma=0")
    # float64: dx[i] = sum over (q, h) with idx[q, h] == i of sum_k w[q, h, k] d_wf[q, k, :]
    idc = idx.cpu()
    valid = idc < ns
    nb = s_pts.double()[idc.clamp(max=ns - 1)] - q_pts.double()[:, None, :]
    w = (1.0 - (nb[:, :, None, :] - kp.double()[None, None]).norm(dim=-1) / extent).clamp(min=0.0) * valid[:, :, None]
    contrib = torch.einsum("qhk,qkc->qhc", w, d_wf.double().view(nq, 15, cin))
    want = torch.zeros(ns + 1, cin, dtype=torch.float64)
    want.index_add_(0, idc.clamp(max=ns).reshape(-1), contrib.reshape(-1, cin))
    assert rel(old, want[:ns]) < TOL, ("valu", rel(old, want[:ns]))
    assert rel(new, want[:ns]) < TOL, ("mfma", rel(new, want[:ns]))
    assert rel(new, old) < 1e-5


@pytest.mark.parametrize("n,ms,heads,d", [(381, 382, 4, 128), (382, 381, 4, 128), (33, 65, 2, 64), (100, 1216, 1, 32), (1, 3, 1, 32), (500, 700, 4, 128),
                                          (450, 31, 3, 64)])
def test_attention_backward_one_launch(cuda, n, ms, heads, d):
    """pcrcg_attention_backward (include/pcrcg_train.h): dq, dk, dv of out_h = softmax(q_h k_h^T / sqrt(d)) v_h
    (ref:models/gcn.py:151-155) against float64 autograd, ADDED to what the buffers hold; strided operands as the train
    tape has them (column slices of wider matrices); shapes beyond the kernel are refused."""
    from pcrcg_amd import _lib, ops
    if "deterministic=1" in os.environ.get("PCRCG_DEBUG", ""):
        pytest.skip("the one-launch backward adds by float atomics: the tape keeps the per-head path under deterministic=1")
    g = torch.Generator().manual_seed(n * 3 + ms + d)
    ch = heads * d
    wide_q = torch.randn(n, ch + 8, generator=g) * 1.3
    wide_kv = torch.randn(ms, 2 * ch + 4, generator=g) * 1.3
    d_out = torch.randn(n, ch, generator=g)
    q64 = wide_q[:, 4:4 + ch].double().clone().requires_grad_(True)
    k64 = wide_kv[:, :ch].double().clone().requires_grad_(True)
    v64 = wide_kv[:, ch + 4:].double().clone().requires_grad_(True)
    outs = []
    for h in range(heads):
        sl = slice(h * d, (h + 1) * d)
        outs.append(torch.softmax(q64[:, sl] @ k64[:, sl].t() / d ** 0.5, 1) @ v64[:, sl])
    out64 = torch.cat(outs, 1)
    out64.backward(d_out.double())
    L = _lib.lib()
    wq, wkv, do = wide_q.to(cuda), wide_kv.to(cuda), d_out.to(cuda)
    qd, kd, vd = wq[:, 4:4 + ch], wkv[:, :ch], wkv[:, ch + 4:]
    out = ops.attention(qd, kd, vd, heads)
    assert rel(out, out64.detach()) < 5e-6
    assert L.pcrcg_attention_backward_supported(n, ms, d, wq.stride(0), wkv.stride(0), wkv.stride(0), ch) == 1
    base = torch.randn(n, ch, generator=g).to(cuda)                     # dq already holds something: the kernel adds
    dq, dk, dv = base.clone(), torch.zeros(ms, ch, device=cuda), torch.zeros(ms, ch, device=cuda)
    _lib.check(L.pcrcg_attention_backward(qd.data_ptr(), wq.stride(0), kd.data_ptr(), wkv.stride(0), vd.data_ptr(), wkv.stride(0),
                                          out.data_ptr(), ch, do.data_ptr(), ch, dq.data_ptr(), ch, dk.data_ptr(), ch, dv.data_ptr(), ch,
                                          n, ms, heads, d, d ** -0.5, ops._stream()), "pcrcg_attention_backward")
    torch.cuda.synchronize()
    assert rel(dq - base, q64.grad) < 2e-5 and rel(dk, k64.grad) < 2e-5 and rel(dv, v64.grad) < 2e-5
    assert L.pcrcg_attention_backward_supported(n, 1217, d, ch, ch, ch, ch) == 0       # the score tile no longer fits LDS
    assert L.pcrcg_attention_backward_supported(n, ms, 48, ch, ch, ch, ch) == 0


@pytest.mark.parametrize("n,c,slope", [(500, 64, 0.1), (1000, 130, 1.0), (37, 8, 0.0), (4000, 32, 0.2)])
def test_instnorm_lrelu_grads(cuda, n, c, slope):
    g = torch.Generator().manual_seed(n + c)
    x, dy = torch.randn(n, c, generator=g) * 2 + 0.3, torch.randn(n, c, generator=g)
    x1 = _leaf(x, cuda)
    y = AG.instnorm_lrelu(x1, slope)
    y.backward(dy.to(cuda))
    x0 = _leaf(x, dtype=torch.float64)
    y0 = torch.nn.functional.leaky_relu(MR.instance_norm_rows(x0), slope)
    y0.backward(dy.double())
    assert rel(y, y0) < TOL and rel(x1.grad, x0.grad) < TOL


def test_pool_grads(cuda, mini):
    g = torch.Generator().manual_seed(5)
    for c in (16, 70):
        x = torch.randn(mini["points"][0].shape[0], c, generator=g)
        pools, ups = mini["pools"][0], mini["upsamples"][0]
        dy = torch.randn(pools.shape[0], c, generator=g)
        x1 = _leaf(x, cuda)
        y = AG.max_pool(x1, pools.to(cuda))
        y.backward(dy.to(cuda))
        x0 = _leaf(x, dtype=torch.float64)
        y0 = MR.max_pool(x0, pools)
        y0.backward(dy.double())
        assert rel(y, y0) < 1e-6 and rel(x1.grad, x0.grad) < TOL
        # nearest upsample: coarse features -> fine points
        xc = torch.randn(mini["points"][1].shape[0], c, generator=g)
        dyu = torch.randn(ups.shape[0], c, generator=g)
        xc1 = _leaf(xc, cuda)
        yu = AG.closest_pool(xc1, ups.to(cuda))
        yu.backward(dyu.to(cuda))
        xc0 = _leaf(xc, dtype=torch.float64)
        yu0 = MR.closest_pool(xc0, ups)
        yu0.backward(dyu.double())
        assert rel(yu, yu0) < 1e-6 and rel(xc1.grad, xc0.grad) < TOL


def test_softmax_rows_grads(cuda):
    g = torch.Generator().manual_seed(9)
    s, dp = torch.randn(190, 381, generator=g) * 3, torch.randn(190, 381, generator=g)
    s1 = _leaf(s, cuda)
    p = AG.softmax_rows(s1, 0.125)
    p.backward(dp.to(cuda))
    s0 = _leaf(s, dtype=torch.float64)
    p0 = torch.softmax(s0 * 0.125, dim=1)
    p0.backward(dp.double())
    assert rel(p, p0) < TOL and rel(s1.grad, s0.grad) < TOL


@pytest.mark.parametrize("n,c,k", [(96, 64, 10), (200, 130, 7)])
def test_edge_conv_grads(cuda, n, c, k):
    """DGCNN edge conv: InstanceNorm2d statistics over all n*k edges make every edge receive gradient."""
    from pcrcg_amd import ops
    g = torch.Generator().manual_seed(n)
    coords = torch.rand(n, 3, generator=g)
    ctr, nbr, dy = torch.randn(n, c, generator=g), torch.randn(n, c, generator=g), torch.randn(n, c, generator=g)
    idx = ops.knn(coords.to(cuda), k)
    c1, n1 = _leaf(ctr, cuda), _leaf(nbr, cuda)
    y = AG.edge_conv(c1, n1, idx, 0.2)
    y.backward(dy.to(cuda))
    c0, n0 = _leaf(ctr, dtype=torch.float64), _leaf(nbr, dtype=torch.float64)
    e = c0[:, None, :] + n0[idx.cpu().long()]                         # [n, k, c]
    mean = e.mean((0, 1), keepdim=True)
    var = e.var((0, 1), unbiased=False, keepdim=True)
    y0 = torch.nn.functional.leaky_relu((e - mean) / torch.sqrt(var + 1e-5), 0.2).max(1)[0]
    y0.backward(dy.double())
    assert rel(y, y0) < TOL
    assert rel(c1.grad, c0.grad) < TOL and rel(n1.grad, n0.grad) < TOL


def test_kpconv_whole_op_c_abi_matches_autograd_path(cuda, mini):
    """pcrcg_kpconv_forward / pcrcg_kpconv_backward called the way a C host would (raw pointers, caller
    workspaces) give the same output and gradients as the autograd wrappers."""
    import ctypes
    from pcrcg_amd import _lib
    from pcrcg_amd.kernel_points import load_kernels
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    cin, cout, level = 64, 48, 1
    s_pts = mini["points"][level].to(cuda)
    q_pts = mini["points"][level + 1].to(cuda)
    idx = mini["pools"][level].to(cuda).contiguous()
    kp = torch.tensor(load_kernels(0.125, 15, dimension=3, fixed="center"), dtype=torch.float32, device=cuda)
    x = torch.randn(s_pts.shape[0], cin, generator=g).to(cuda)
    w = (torch.randn(15, cin, cout, generator=g) * 0.2).to(cuda)
    dy = torch.randn(q_pts.shape[0], cout, generator=g).to(cuda)
    nq, h, ns = idx.shape[0], idx.shape[1], s_pts.shape[0]
    stream = torch.cuda.current_stream().cuda_stream
    fwd_bytes = L.pcrcg_kpconv_forward_ws_bytes(nq, ns, cin)
    fwd_ws = torch.empty(fwd_bytes, dtype=torch.uint8, device=cuda)
    out = torch.empty(nq, cout, device=cuda)
    rc = L.pcrcg_kpconv_forward(q_pts.data_ptr(), nq, s_pts.data_ptr(), ns, idx.data_ptr(), h, h, x.data_ptr(), cin,
                                kp.data_ptr(), 0.1, w.data_ptr(), cout, out.data_ptr(), cout, fwd_ws.data_ptr(), fwd_bytes,
                                stream)
    assert rc == 0, L.pcrcg_last_error()
    bwd_bytes = L.pcrcg_kpconv_backward_ws_bytes(nq, cin, cout)
    bwd_ws = torch.empty(bwd_bytes, dtype=torch.uint8, device=cuda)
    dx, dw = torch.zeros(ns, cin, device=cuda), torch.empty(15 * cin, cout, device=cuda)
    rc = L.pcrcg_kpconv_backward(q_pts.data_ptr(), nq, s_pts.data_ptr(), ns, idx.data_ptr(), h, h, cin, kp.data_ptr(), 0.1,
                                 w.data_ptr(), cout, dy.data_ptr(), cout, fwd_ws.data_ptr(), fwd_bytes, dx.data_ptr(),
                                 dw.data_ptr(), bwd_ws.data_ptr(), bwd_bytes, stream)
    assert rc == 0, L.pcrcg_last_error()
    x1, w1 = _leaf(x), _leaf(w)
    y = AG.kpconv(x1, w1, q_pts, s_pts, idx, kp, 0.1)
    y.backward(dy)
    assert rel(out, y) < 1e-6 and rel(dw.reshape(15, cin, cout), w1.grad) < 1e-5 and rel(dx, x1.grad) < 1e-5
    # too small a workspace is refused before any launch
    assert L.pcrcg_kpconv_forward(q_pts.data_ptr(), nq, s_pts.data_ptr(), ns, idx.data_ptr(), h, h, x.data_ptr(), cin,
                                  kp.data_ptr(), 0.1, w.data_ptr(), cout, out.data_ptr(), cout, fwd_ws.data_ptr(), 1024,
                                  stream) == -2


def test_kpconv_more_than_64_neighbours(cuda):
    """Tables wider than one wavefront (H > 64: the kernels loop over 64-neighbour chunks) -- forward and both
    gradients against the oracle, on a dense cloud where most rows really have > 64 neighbours."""
    from pcrcg_amd import ops
    from pcrcg_amd.kernel_points import load_kernels
    g = torch.Generator().manual_seed(4)
    pts = torch.rand(1500, 3, generator=g) * 0.2
    lens = torch.tensor([900, 600], dtype=torch.int32)
    grid = ops.CellGrid(pts.to(cuda), lens.to(cuda), 0.0625)
    idx, meta = grid.query(pts.to(cuda), lens.to(cuda), 90)
    assert int(meta[0]) > 64 and int((idx[:, 64] < 1500).sum()) > 200          # column 64 is really populated
    kp = torch.tensor(load_kernels(0.0625, 15, dimension=3, fixed="center"), dtype=torch.float32)
    for cin, cout in ((1, 8), (64, 16), (20, 12)):
        x, w = torch.randn(1500, cin, generator=g), torch.randn(15, cin, cout, generator=g) * 0.2
        dy = torch.randn(1500, cout, generator=g)
        x1, w1 = _leaf(x, cuda), _leaf(w, cuda)
        y = AG.kpconv(x1, w1, pts.to(cuda), pts.to(cuda), idx, kp.to(cuda), 0.05)
        y.backward(dy.to(cuda))
        x0, w0 = _leaf(x, dtype=torch.float64), _leaf(w, dtype=torch.float64)
        y0 = MR.kpconv(pts.double(), pts.double(), idx.cpu(), x0, kp.double(), w0, 0.05)
        y0.backward(dy.double())
        assert rel(y, y0) < TOL and rel(w1.grad, w0.grad) < TOL and rel(x1.grad, x0.grad) < TOL, (cin, cout)
