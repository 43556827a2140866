"""GPU: the C++ pyramid builder (pcrcg_pyramid_build, one C-ABI call per pair) against the op-by-op Python mirror
(pyramid_steps) -- which tests/test_frontend_gpu.py and tests/test_tieorder_gpu.py pin to the reference's raw table
digests -- and the multi-stream pair engine against sequential execution."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from pcrcg_amd import indoor_config, kitti_config, synthetic
from pcrcg_amd.architectures import KPFCNN
from pcrcg_amd.pyramid import NativePyramid, build_pyramid, build_pyramid_native, pyramid_steps

pytestmark = pytest.mark.gpu


def _python_pyramid(pts, lens, cfg, limits, tie_order):
    steps = pyramid_steps(pts, lens, cfg, limits, tie_order=tie_order)
    try:
        while True:
            next(steps).synchronize()
    except StopIteration as done:
        return done.value


def _pair(recipe, seed, dev):
    if recipe == "K120k":
        src, tgt = synthetic.slab_pair(120000, seed)
    elif recipe == "U30k":
        src, tgt = synthetic.uniform_pair(30000, 1.07, seed)
    else:
        src, tgt = synthetic.pair(recipe, seed)
    return (torch.from_numpy(np.concatenate([src, tgt])).to(dev),
            torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev))


@pytest.mark.parametrize("recipe,tie_order", [("mini", "auto"), ("C1", "auto"), ("C1", "index"), ("T8k", "auto"),
                                              ("S30k", "auto"), ("S30k", "index"), ("U30k", "auto"), ("K120k", "auto")])
def test_native_pyramid_equals_python_mirror(cuda, recipe, tie_order):
    cfg = kitti_config() if recipe == "K120k" else indoor_config()
    limits = synthetic.LIMITS.get(recipe, [20, 26, 30, 32] if recipe == "mini" else synthetic.LIMITS["C1"])
    pts, lens = _pair(recipe, 0, cuda)
    want = _python_pyramid(pts, lens, cfg, limits, tie_order)
    got = build_pyramid_native(pts, lens, cfg, limits, tie_order)
    torch.cuda.synchronize()
    assert got["stack_lengths_host"] == want["stack_lengths_host"]
    for l in range(cfg.num_layers):
        assert torch.equal(got["points"][l].view(torch.int32), want["points"][l].view(torch.int32)), l
        assert torch.equal(got["stack_lengths"][l], want["stack_lengths"][l].to(torch.int32)), l
        for key in ("neighbors", "pools", "upsamples"):
            assert got[key][l].shape == want[key][l].shape, (key, l, got[key][l].shape, want[key][l].shape)
            assert got[key][l].dtype == torch.int64
            assert torch.equal(got[key][l], want[key][l]), (key, l)
    assert torch.equal(got["features"], want["features"])


@pytest.mark.parametrize("tie_order", ["auto", "index"])
def test_native_pyramid_redo_pass_for_dense_rows(cuda, tie_order):
    """Rows with more than 256 hits: the builder launches the search's redo pass only after the metadata round trip has
    shown that a table needs it (normally none does).  Two dense clouds (about 400 hits per level-0 row, half of the
    points snapped to a lattice so that tie rows are among them) against the op-by-op mirror, whose searches always run
    both passes."""
    rng = np.random.RandomState(21)
    a = (rng.rand(3200, 3) * 0.16).astype(np.float32)
    b = (rng.rand(2800, 3) * 0.15).astype(np.float32)
    a[::2] = np.round(a[::2] * 64) / 64
    pts = torch.from_numpy(np.concatenate([a, b])).to(cuda)
    lens = torch.tensor([len(a), len(b)], dtype=torch.int32, device=cuda)
    cfg, limits = indoor_config(), [600, 300, 200, 100]
    want = _python_pyramid(pts, lens, cfg, limits, tie_order)
    assert want["neighbors"][0].shape[1] > 256                       # the case is what it claims to be
    got = build_pyramid_native(pts, lens, cfg, limits, tie_order)
    torch.cuda.synchronize()
    for l in range(cfg.num_layers):
        for key in ("neighbors", "pools", "upsamples"):
            assert torch.equal(got[key][l], want[key][l]), (key, l)


@pytest.mark.parametrize("recipe", ["C1", "S30k", "T8k"])
def test_native_pyramid_reference_digests(cuda, golden_dir, recipe):
    """Straight against the reference's own raw output: SHA-256 of every subsampled level and of every untruncated
    int32 table as the unmodified reference C++ produced them (tests/golden/frontend_digests.json).  With limits
    above the longest list the builder returns the reference's full-width tables."""
    dig = json.load(open(os.path.join(golden_dir, "frontend_digests.json")))[recipe]
    cfg = indoor_config()
    pts, lens = _pair(recipe, 0, cuda)
    b = build_pyramid(pts, lens, cfg, [200, 200, 200, 200])

    def sha(t, dtype):
        return hashlib.sha256(np.ascontiguousarray(t.cpu().numpy().astype(dtype)).tobytes()).hexdigest()

    for l in range(cfg.num_layers):
        assert sha(b["points"][l], np.float32) == dig[f"points{l}"]["sha256"], l
        assert sha(b["stack_lengths"][l], np.int32) == dig[f"lens{l}"]["sha256"], l
        for key, name in (("neighbors", "conv"), ("pools", "pool"), ("upsamples", "up")):
            if f"{name}{l}" in dig:
                t = b[key][l]
                assert list(t.shape) == dig[f"{name}{l}"]["shape"], (name, l, t.shape)
                assert sha(t, np.int32) == dig[f"{name}{l}"]["sha256"], (name, l)


def test_two_pairs_in_one_chain_equal_two_builds(cuda):
    """group = 2: two (different-sized) pairs stacked into one call come out as their own batches -- the tables each
    pair gets when built alone, entry for entry, incl. the reference's order inside tie groups (T8k)."""
    cfg = indoor_config()
    limits = synthetic.LIMITS["C1"]
    for ra, rb in (("C1", "T8k"), ("S30k", "C1"), ("mini", "mini")):
        (pa, la), (pb, lb) = _pair(ra, 0, cuda), _pair(rb, 1, cuda)
        nat = NativePyramid(cfg, limits, "auto")
        b, arena, lens_h, slot = nat.build(torch.cat([pa, pb]), torch.cat([la, lb]), group=2)
        torch.cuda.synchronize()
        assert int(nat.status[slot]) == 0 and len(b) == 2
        for i, (pts, lens) in enumerate(((pa, la), (pb, lb))):
            got = nat.as_dict(b[i], arena, lens_h, part=(2 * i, 2))
            want = build_pyramid_native(pts, lens, cfg, limits, "auto")
            assert got["stack_lengths_host"] == want["stack_lengths_host"], (ra, rb, i)
            assert b[i].len_src_c == want["stack_lengths_host"][-1][0]
            for l in range(cfg.num_layers):
                assert torch.equal(got["points"][l].view(torch.int32), want["points"][l].view(torch.int32)), (ra, rb, i, l)
                assert torch.equal(got["stack_lengths"][l], want["stack_lengths"][l]), (ra, rb, i, l)
                for key in ("neighbors", "pools", "upsamples"):
                    assert got[key][l].shape == want[key][l].shape, (ra, rb, i, key, l, got[key][l].shape, want[key][l].shape)
                    assert torch.equal(got[key][l], want[key][l]), (ra, rb, i, key, l)
            assert torch.equal(got["features"], want["features"])


def test_arena_grows_when_levels_shrink_less_than_assumed(cuda):
    """A cloud whose subsampled levels keep more than half of their rows (dl far below the point spacing)."""
    cfg = indoor_config(first_subsampling_dl=0.0005)
    rng = np.random.RandomState(0)
    pts = torch.from_numpy(rng.rand(4000, 3).astype(np.float32)).to(cuda)
    lens = torch.tensor([2000, 2000], dtype=torch.int32, device=cuda)
    limits = [8, 8, 8, 8]
    nat = NativePyramid(cfg, limits, "auto")
    b, arena, lens_h, slot = nat.build(pts, lens)
    torch.cuda.synchronize()
    assert nat.shrink == 1.0 and lens_h[1] == [2000, 2000]
    want = _python_pyramid(pts, lens, cfg, limits, "auto")
    got = nat.as_dict(b, arena, lens_h)
    for l in range(4):
        assert torch.equal(got["neighbors"][l], want["neighbors"][l])


def test_pair_engine_matches_sequential(cuda):
    """Front threads sharing the front-end stream, three model streams, arenas reused pair after pair: every output
    equals the sequential run."""
    from pcrcg_amd.pairstream import PairStreams
    cfg = indoor_config(first_feats_dim=64, gnn_feats_dim=128)
    torch.manual_seed(0)
    np.random.seed(0)
    net = KPFCNN(cfg).to(cuda).eval()
    limits = synthetic.LIMITS["C1"]
    pairs = [_pair("C1", seed, cuda) for seed in range(6)] + [_pair("mini", 0, cuda), _pair("T8k", 0, cuda)]
    ref = []
    with torch.no_grad():
        for pts, lens in pairs:
            ref.append(net(build_pyramid(pts, lens, cfg, limits)))
    torch.cuda.synchronize()
    # (model streams, front threads, pairs per build, pairs per forward call): one call per pair, the default two pairs
    # per call, and builds / calls of up to four pairs
    for workers, fronts, per_build, per_fwd in ((1, 1, 2, 1), (3, 2, 2, 2), (2, 1, 4, 4), (2, 1, 3, 2)):
        eng = PairStreams(net, cfg, limits, cuda, model_streams=workers, front_threads=fronts, up_nearest=(workers == 3),
                          pairs_per_build=per_build, pairs_per_forward=per_fwd,
                          adaptive_jobs=(workers == 3),      # the timing-dependent grouping in ONE configuration only
                          forest_stream=(2 if per_build == 4 else (1 if per_build == 3 else 0)))   # chains as a DAG over side streams, and in line
        outs, submitted, total = [], 0, 3 * len(pairs)
        for i in range(total):
            while submitted < min(total, i + 5):
                eng.submit(*pairs[submitted % len(pairs)])
                submitted += 1
            outs.append(eng.result())
            assert isinstance(outs[-1]["done_event"], torch.cuda.Event)
            if i % 3 == 0:
                assert PairStreams.check(outs[-1]) is outs[-1]        # this pair's own tie-restore status, after its kernels
        eng.drain()
        with pytest.raises(RuntimeError):
            eng.result()
        eng.close()
        for i, a in enumerate(outs):
            for k in ("feats_f", "scores_overlap", "scores_saliency"):
                # (split-K GEMMs accumulate with fp32 atomics: results are equal up to summation order)
                r = ref[i % len(pairs)][k]
                assert float((a[k] - r).abs().max()) <= 1e-5 * float(r.abs().max()), (workers, i, k)


def test_network_call_refuses_a_short_workspace(cuda):
    """pcrcg_kpfcnn_forward with less workspace than pcrcg_kpfcnn_ws_bytes asks for: status -2 and a message, nothing
    launched past the end of the buffer; bad descriptors are refused by both entry points."""
    import ctypes
    from pcrcg_amd import _lib
    from pcrcg_amd.runner import Outputs
    cfg = indoor_config(first_feats_dim=32, gnn_feats_dim=64)
    torch.manual_seed(0)
    net = KPFCNN(cfg).to(cuda).eval()
    pts, lens = _pair("mini", 0, cuda)
    batch = build_pyramid(pts, lens, cfg, [20, 26, 30, 32])
    runner = net.runner()
    b, keep, dev = runner.batch_struct(batch)
    desc = runner.descriptor()
    L = _lib.lib()
    need = L.pcrcg_kpfcnn_ws_bytes(ctypes.byref(desc), ctypes.byref(b))
    assert need > 4096
    n0 = b.n_points[0]
    outs = [torch.full((n0, desc.final_dim), 7.0, device=cuda), torch.full((n0,), 7.0, device=cuda), torch.full((n0,), 7.0, device=cuda)]
    o = Outputs(*(t.data_ptr() for t in outs))
    ws = torch.empty(need, dtype=torch.uint8, device=cuda)
    st = torch.cuda.current_stream().cuda_stream
    rc = L.pcrcg_kpfcnn_forward(ctypes.byref(desc), ctypes.byref(b), ctypes.byref(o), ws.data_ptr(), need // 2, st)
    assert rc == -2 and b"workspace too small" in L.pcrcg_last_error()
    torch.cuda.synchronize()
    assert all(float(t.min()) == 7.0 and float(t.max()) == 7.0 for t in outs)        # untouched
    assert L.pcrcg_kpfcnn_forward(ctypes.byref(desc), ctypes.byref(b), ctypes.byref(o), ws.data_ptr(), need, st) == 0
    torch.cuda.synchronize()
    with torch.no_grad():
        want = net(batch)
    assert torch.equal(outs[1], want["scores_overlap"]) or float((outs[1] - want["scores_overlap"]).abs().max()) < 1e-6
    b.len_src_c = 0                                                                     # a descriptor that cannot be right
    assert L.pcrcg_kpfcnn_ws_bytes(ctypes.byref(desc), ctypes.byref(b)) == 0
    assert L.pcrcg_kpfcnn_forward(ctypes.byref(desc), ctypes.byref(b), ctypes.byref(o), ws.data_ptr(), need, st) == -1


def test_engine_surfaces_bad_input_on_result(cuda):
    """A pair the front end cannot take (CPU tensor) does not hang the engine: its result() raises, later pairs work."""
    from pcrcg_amd.pairstream import PairStreams
    cfg = indoor_config(first_feats_dim=32, gnn_feats_dim=64)
    torch.manual_seed(0)
    net = KPFCNN(cfg).to(cuda).eval()
    good = _pair("mini", 0, cuda)
    eng = PairStreams(net, cfg, [20, 26, 30, 32], cuda, pairs_per_build=1)
    try:
        eng.submit(good[0].cpu(), good[1].cpu())
        eng.submit(*good)
        with pytest.raises(Exception):
            eng.result()
        out = PairStreams.check(eng.result())
        torch.cuda.synchronize()
        with torch.no_grad():
            ref = net(build_pyramid(*good, cfg, [20, 26, 30, 32]))
        assert float((out["scores_overlap"] - ref["scores_overlap"]).abs().max()) <= 1e-5
    finally:
        eng.close()


@pytest.mark.parametrize("recipe", ["T8k", "S30k"])
def test_one_column_upsample_tables_are_column_zero(cuda, recipe):
    """pcrcg_pyramid_cfg.up_nearest (what the pair engine builds): the upsample tables have ONE column, the nearest coarse
    point in the reference's order among equidistant ones -- column 0 of the full tables, entry for entry (T8k: thousands of
    rows where two coarse points are exactly equidistant); everything else is unchanged."""
    cfg, limits = indoor_config(), synthetic.LIMITS.get(recipe, synthetic.LIMITS["C1"])
    pts, lens = _pair(recipe, 0, cuda)
    full = build_pyramid_native(pts, lens, cfg, limits)
    nat = NativePyramid(cfg, limits, up_nearest=True)
    b, arena, lens_h, slot = nat.build(pts, lens, fresh_arena=True)
    torch.cuda.synchronize()
    assert int(nat.status[slot]) == 0
    got = nat.as_dict(b, arena, lens_h)
    for l in range(cfg.num_layers):
        assert torch.equal(got["neighbors"][l], full["neighbors"][l]) and torch.equal(got["pools"][l], full["pools"][l])
        if l + 1 < cfg.num_layers:
            assert got["upsamples"][l].shape == (full["upsamples"][l].shape[0], 1)
            assert torch.equal(got["upsamples"][l][:, 0], full["upsamples"][l][:, 0]), l


def test_side_streams_leave_the_tables_unchanged(cuda):
    """pcrcg_pyramid_cfg.side_stream / side_stream2 (round 6: the subsamplings and the KD-forests beside the searches): the
    same tables, entry for entry, as the one-stream build -- for a single pair and for a grouped build of three pairs."""
    cfg = indoor_config()
    limits = synthetic.LIMITS["C1"]
    sub, forest = torch.cuda.Stream(device=cuda), torch.cuda.Stream(device=cuda)
    for recipes in (("C1",), ("T8k", "C1", "mini")):
        parts = [_pair(r, i, cuda) for i, r in enumerate(recipes)]
        pts, lens = torch.cat([p for p, _ in parts]), torch.cat([l for _, l in parts])
        group = 2 if len(parts) > 1 else 0
        outs = []
        for side in (None, (sub, None), (sub, forest)):
            nat = NativePyramid(cfg, limits, "auto")
            if side is not None:
                nat.set_side_streams(*side)
            b, arena, lens_h, slot = nat.build(pts, lens, group=group)
            torch.cuda.synchronize()
            assert int(nat.status[slot]) == 0
            bs = list(b) if group else [b]
            outs.append([nat.as_dict(bb, arena, lens_h, part=(2 * i, 2) if group else None) for i, bb in enumerate(bs)])
        for other in outs[1:]:
            for got, want in zip(other, outs[0]):
                assert got["stack_lengths_host"] == want["stack_lengths_host"]
                for l in range(cfg.num_layers):
                    assert torch.equal(got["points"][l].view(torch.int32), want["points"][l].view(torch.int32))
                    for key in ("neighbors", "pools", "upsamples"):
                        assert torch.equal(got[key][l], want[key][l]), (recipes, key, l)


_ENGINE_PROBE = r"""
import json, sys, warnings
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from pcrcg_amd import indoor_config, synthetic
from pcrcg_amd.architectures import KPFCNN
from pcrcg_amd.pairstream import PairStreams
dev = torch.device("cuda:0")
cfg = indoor_config(first_feats_dim=32, gnn_feats_dim=64)
torch.manual_seed(0); np.random.seed(0)
net = KPFCNN(cfg).to(dev).eval()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    eng = PairStreams(net, cfg, [20, 26, 30, 32], dev)
src, tgt = synthetic.pair("mini", 0)
pts = torch.from_numpy(np.concatenate([src, tgt])).to(dev)
lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev)
for _ in range(6):
    eng.submit(pts, lens)
outs = [eng.result() for _ in range(6)]
eng.drain(); eng.close()
print(json.dumps({"classes": eng.pipe_classes, "warnings": [str(x.message)[:80] for x in w],
                  "finite": bool(all(torch.isfinite(o["feats_f"]).all() for o in outs))}))
"""


@pytest.mark.parametrize("queues", [None, "8"])
def test_engine_streams_sit_on_distinct_dispatchers(cuda, queues):
    """The engine picks its streams by hardware dispatcher class (pcrcg_stream_pipe_classes), so its premise -- the front end
    and each of the three model streams on a dispatcher of its own -- holds in a process with the runtime's default of four
    hardware queues and in one with GPU_MAX_HW_QUEUES=8 alike (rounds 1-5 needed the variable set before the first HIP
    call and lost 16 % silently without it).  Fresh processes: the variable is read once, when HIP starts."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    env["PCRCG_FOREST_STREAM"] = "2"            # with the engine's side streams (off by default), so that their class is checked too
    if queues:
        env["GPU_MAX_HW_QUEUES"] = queues
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _ENGINE_PROBE, repo], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    c = rec["classes"]
    assert rec["finite"] and c["distinct"] and not rec["warnings"], rec
    assert len(set([c["front"]] + c["model"])) == 4, c
    assert c["side_class"] == c["front"]


def test_stream_pipe_classes_finds_four_dispatchers(cuda):
    """pcrcg_stream_pipe_classes (the measurement the engine's stream choice rests on): among sixteen library-created
    streams it finds gfx950's four hardware dispatchers -- four classes, each with several members --, gives the same answer
    twice, puts a stream in its own class when it is listed twice, and rejects bad arguments before touching the GPU."""
    import ctypes
    from pcrcg_amd import _lib
    L = _lib.lib()
    streams = []
    for _ in range(16):
        h = ctypes.c_void_p()
        _lib.check(L.pcrcg_stream_create(ctypes.byref(h), 0), "pcrcg_stream_create")
        streams.append(h.value)
    scratch = torch.zeros(16, dtype=torch.int32, device=cuda)
    torch.cuda.synchronize()
    arr = (ctypes.c_void_p * 17)(*(streams + [streams[3]]))
    runs = []
    for _ in range(2):
        cls = (ctypes.c_int * 17)()
        _lib.check(L.pcrcg_stream_pipe_classes(arr, 17, cls, scratch.data_ptr()), "pcrcg_stream_pipe_classes")
        runs.append(list(cls))
    assert runs[0] == runs[1], runs
    cls = runs[0]
    assert cls[0] == 0 and cls[16] == cls[3]
    assert sorted(set(cls)) == [0, 1, 2, 3], cls                      # four dispatchers, no more, no fewer
    assert min(cls.count(c) for c in range(4)) >= 2, cls              # and every one serves several of sixteen streams
    out = (ctypes.c_int * 17)()
    assert L.pcrcg_stream_pipe_classes(arr, 0, out, scratch.data_ptr()) == -1
    assert L.pcrcg_stream_pipe_classes(None, 4, out, scratch.data_ptr()) == -1
    assert L.pcrcg_stream_pipe_classes(arr, 4, out, None) == -1
    for h in streams:
        _lib.check(L.pcrcg_stream_destroy(ctypes.c_void_p(h)), "pcrcg_stream_destroy")


def test_a_level_that_outgrows_its_bound_is_reported_and_rebuilt(cuda):
    """Round 6 sizes every level from the bound pcrcg_pyramid_cfg::shrink.  Clouds whose subsampled levels keep nearly all
    their rows (dl far below the point spacing) outgrow the default bound of one half: the library reports PCRCG_EWORKSPACE
    at the end of the chain -- with side streams and several pairs per chain as well --, nothing is corrupted, and the same
    builder, asked again with shrink = 1, returns the tables of a builder that started with shrink = 1."""
    import ctypes
    from pcrcg_amd import _lib
    cfg = indoor_config(first_subsampling_dl=0.0005)
    rng = np.random.RandomState(1)
    limits = [8, 8, 8, 8]
    pts = [torch.from_numpy(rng.rand(3000, 3).astype(np.float32)).to(cuda) for _ in range(3)]
    lens = [torch.tensor([1500, 1500], dtype=torch.int32, device=cuda) for _ in range(3)]
    sub, forest = torch.cuda.Stream(device=cuda), torch.cuda.Stream(device=cuda)
    want = NativePyramid(cfg, limits, "auto")
    want.shrink = 1.0
    wb, warena, wlens, wslot = want.build(pts, lens, group=2)
    torch.cuda.synchronize()
    nat = NativePyramid(cfg, limits, "auto")
    nat.set_side_streams(sub, forest)
    # the raw call with the default bound: the status code, not an exception, and a message that names the bound
    nat.cfg.shrink, nat.cfg.group = 0.5, 2
    L = _lib.lib()
    need = L.pcrcg_pyramid_ws_bytes(9000, 6, ctypes.byref(nat.cfg))
    arena = torch.empty(int(need), dtype=torch.uint8, device=cuda)
    from pcrcg_amd.runner import Batch
    b = (Batch * 3)()
    h_len = (ctypes.c_int * (4 * 6))()
    pp = (ctypes.c_void_p * 3)(*[p.data_ptr() for p in pts])
    lp = (ctypes.c_void_p * 3)(*[l.data_ptr() for l in lens])
    pn, ln = (ctypes.c_int * 3)(3000, 3000, 3000), (ctypes.c_int * 3)(2, 2, 2)
    rc = L.pcrcg_pyramid_build_parts(pp, pn, lp, ln, 3, ctypes.byref(nat.cfg), arena.data_ptr(), arena.numel(), nat.scratch.data_ptr(),
                                     ctypes.byref(b), h_len, nat.status.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    assert rc == -2 and b"shrink" in L.pcrcg_last_error()
    torch.cuda.synchronize()
    # the Python builder repeats the call with the bound that always fits
    gb, garena, glens, gslot = nat.build(pts, lens, group=2)
    torch.cuda.synchronize()
    assert nat.shrink == 1.0 and glens == wlens and glens[1] == [1500] * 6
    for i in range(3):
        got, ref = nat.as_dict(gb[i], garena, glens, part=(2 * i, 2)), want.as_dict(wb[i], warena, wlens, part=(2 * i, 2))
        for l in range(4):
            assert torch.equal(got["points"][l].view(torch.int32), ref["points"][l].view(torch.int32))
            for key in ("neighbors", "pools", "upsamples"):
                assert torch.equal(got[key][l], ref[key][l]), (i, key, l)


def test_build_pyramid_over_streams_of_other_dispatchers(cuda):
    """build_pyramid(side_streams=ops.streams_on_other_dispatchers()): the drop-in, one-pair-at-a-time path with its chain as a
    DAG over three dispatchers -- the same dict, entry for entry, as the chain in line."""
    from pcrcg_amd import ops
    cfg = indoor_config()
    limits = synthetic.LIMITS["C1"]
    sides = ops.streams_on_other_dispatchers(2)
    cls = ops.stream_pipe_classes([torch.cuda.current_stream()] + sides)
    assert len(set(cls)) == 3, cls
    for recipe in ("C1", "T8k"):
        pts, lens = _pair(recipe, 0, cuda)
        want = build_pyramid(pts, lens, cfg, limits)
        got = build_pyramid(pts, lens, cfg, limits, side_streams=tuple(sides))
        assert got["stack_lengths_host"] == want["stack_lengths_host"]
        for l in range(cfg.num_layers):
            assert torch.equal(got["points"][l].view(torch.int32), want["points"][l].view(torch.int32))
            for key in ("neighbors", "pools", "upsamples"):
                assert torch.equal(got[key][l], want[key][l]), (recipe, key, l)
