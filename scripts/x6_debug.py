import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcrcg_amd import _lib, ops
dev = torch.device("cuda:0")
L = _lib.lib()
L.pcrcg_gemm_set_mode(1)
for (m, n, k) in [(15456, 128, 512), (3934, 256, 1024), (763, 256, 3840), (763, 2048, 512), (381, 512, 2048), (256, 128, 256), (64, 64, 256), (64, 64, 128)]:
    torch.manual_seed(1)
    a = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    ref = a.double() @ w.double().t()
    for t in (3, 1):
        for sk in (1, 2, 4):
            os.environ["PCRCG_X6_TILE"], os.environ["PCRCG_X6_SPLITK"] = str(t), str(sk)
            errs = []
            for rep in range(4):
                out = ops.gemm(a, w.t())
                e = (out.double() - ref).abs()
                errs.append(float(e.max() / ref.abs().max()))
            bad = (e > 1e-3 * ref.abs().max())
            rows = bad.any(1).nonzero().flatten()
            cols = bad.any(0).nonzero().flatten()
            print(m, n, k, "tile", t, "splitk", sk, ["%.1e" % x for x in errs],
                  "bad rows", rows[:6].tolist(), len(rows), "bad cols", cols[:6].tolist(), len(cols), flush=True)
