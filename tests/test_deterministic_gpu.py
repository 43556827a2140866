"""GPU: PCRCG_DEBUG=deterministic=1 (include/pcrcg.h) -- results that are a function of the inputs alone.

By default the path adds with floating-point atomics in three places (split-K partial products, InstanceNorm column sums
from the GEMM epilogues, the backward's scatters), so two runs of one batch differ in the last bits (SURVEY.md section 5
"determinism test"; the reference's CPU path is deterministic).  Under the switch two forwards of the S30k pair at full
width -- lone and grouped -- must be bit-identical, and the result must still sit inside the parity bar."""
import numpy as np
import pytest
import torch

from pcrcg_amd import _lib, indoor_config, synthetic
from pcrcg_amd.architectures import KPFCNN
from pcrcg_amd.pyramid import build_pyramid

pytestmark = pytest.mark.gpu
KEYS = ("feats_f", "scores_overlap", "scores_saliency")


def _debug(spec):
    _lib.check(_lib.lib().pcrcg_debug_set(spec.encode() if spec is not None else None), "pcrcg_debug_set")


def _stack(src, tgt, dev):
    return (torch.from_numpy(np.concatenate([src, tgt])).to(dev),
            torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev))


def test_two_s30k_forwards_are_bit_identical(cuda):
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = indoor_config()
    net = KPFCNN(cfg).to(cuda).eval()
    limits = synthetic.LIMITS["S30k"]
    b0 = build_pyramid(*_stack(*synthetic.pair("S30k", 0), cuda), cfg, limits)
    b1 = build_pyramid(*_stack(*synthetic.pair("S30k", 1), cuda), cfg, limits)
    runner = net.runner()
    s0, keep0, dev = runner.batch_struct(b0)
    s1, keep1, _ = runner.batch_struct(b1)
    both = (type(s0) * 2)(s0, s1)                      # (keep0 / keep1 hold what the structs point into)
    try:
        with torch.no_grad():
            default = net(b0)
            _debug("deterministic=1")
            runs = [net(b0) for _ in range(3)]
            groups = [runner.launch_group(both, 2, dev) for _ in range(2)]
        torch.cuda.synchronize()
    finally:
        _debug(None)
    for k in KEYS:
        for r in runs[1:]:
            assert torch.equal(r[k], runs[0][k]), k
        for g in (0, 1):
            assert torch.equal(groups[0][g][k], groups[1][g][k]), (k, g)
        # the same arithmetic up to summation order: far inside the 1e-4 bar of the default path
        d = float((runs[0][k].double() - default[k].double()).abs().max() / default[k].double().abs().max())
        assert d < 1e-5, (k, d)
        d = float((groups[0][0][k].double() - runs[0][k].double()).abs().max() / runs[0][k].double().abs().max())
        assert d < 1e-5, (k, d)
