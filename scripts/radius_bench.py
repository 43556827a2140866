"""Radius search, table by table: the cell-cooperative LDS-staged kernel (k_radius_cells) against the per-query kernel
of rounds 1-3 (k_radius_query) -- results compared entry for entry, then timed (GPU box).

    python scripts/radius_bench.py [S30k] [K120k] [U30k] [T30k] [--reps 20]

Per table: rows x columns, microseconds old / new (HIP events around `reps` back-to-back calls, outputs preallocated by
the wrapper each call -- run it under `rocprofv3 --kernel-trace --stats` for the kernels' own durations), and the
SURVEY.md 8d algorithmic bytes 12 Nq + 12 Ns + 8 Nq cols over the new kernel's time."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pcrcg_amd import indoor_config, kitti_config, ops, synthetic  # noqa: E402
from pcrcg_amd.pyramid import build_pyramid  # noqa: E402

dev = torch.device("cuda:0")


def make(recipe):
    if recipe == "U30k":
        src, tgt = synthetic.uniform_pair(30000, 1.07, 0)
    elif recipe == "K120k":
        src, tgt = synthetic.slab_pair(120000, 0)
    else:
        src, tgt = synthetic.pair(recipe, 0)
    cfg = kitti_config() if recipe == "K120k" else indoor_config()
    pts = torch.from_numpy(np.concatenate([src, tgt])).to(dev)
    lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32, device=dev)
    limits = synthetic.LIMITS.get(recipe) or [40, 40, 40, 40]
    return cfg, pts, lens, limits


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def same(res_a, res_b, tag):
    ia, ma, ca, ta = res_a
    ib, mb, cb, tb = res_b
    ma, mb = ma.tolist(), mb.tolist()
    assert ma == mb, (tag, "meta", ma, mb)
    assert torch.equal(ia, ib), (tag, "tables differ in %d rows" % int((ia != ib).any(1).sum()))
    assert torch.equal(ca, cb), (tag, "counts")
    n = ma[2]
    assert torch.equal(ta[:n].sort().values, tb[:n].sort().values), (tag, "tie rows")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = 20
    if "--reps" in sys.argv:
        reps = int(sys.argv[sys.argv.index("--reps") + 1])
        args.remove(sys.argv[sys.argv.index("--reps") + 1])
    # --mode old|new: only that kernel runs, no comparison (the run a kernel trace is taken from: every table is then
    # exactly 3 + reps consecutive dispatches of one kernel, see scripts/radius_kernel_times.py)
    mode = None
    if "--mode" in sys.argv:
        mode = sys.argv[sys.argv.index("--mode") + 1]
        args.remove(mode)
    for recipe in args or ["S30k"]:
        cfg, pts, lens, limits = make(recipe)
        b = build_pyramid(pts, lens, cfg, limits, tie_order="index")
        L = len(b["points"])
        r0 = float(cfg.first_subsampling_dl) * float(cfg.conv_radius)
        tot_old = tot_new = 0.0
        print(f"== {recipe}: levels {[int(p.shape[0]) for p in b['points']]}, limits {limits}, r0 {r0}")
        grids = [ops.CellGrid(b["points"][l], b["stack_lengths"][l].to(torch.int32), r0 * 2 ** l) for l in range(L)]
        for l in range(L):
            p, ln = b["points"][l], b["stack_lengths"][l].to(torch.int32)
            jobs = [("conv", grids[l], p, ln, grids[l])]
            if l + 1 < L:
                p2, ln2 = b["points"][l + 1], b["stack_lengths"][l + 1].to(torch.int32)
                jobs.append(("pool", grids[l], p2, ln2, grids[l + 1]))
                jobs.append(("up", grids[l + 1], p, ln, grids[l]))
            only = os.environ.get("RADIUS_BENCH_ONLY")
            for kind, g, q, ql, qg in jobs:
                cols = limits[l]
                if only and only != f"{kind}{l}":
                    continue
                if mode:
                    qgm = qg if mode == "new" else None
                    t = timeit(lambda: g.query(q, ql, cols, want_ties=True, query_grid=qgm), reps)
                    print(f"  {kind}{l}: {q.shape[0]:6d} x {cols:2d} over {g.ns:6d}  {mode} {t:7.1f} us (wall)")
                    continue
                old = g.query(q, ql, cols, want_ties=True)
                new = g.query(q, ql, cols, want_ties=True, query_grid=qg)
                same(old, new, f"{recipe} {kind}{l}")
                t_old = timeit(lambda: g.query(q, ql, cols, want_ties=True), reps)
                t_new = timeit(lambda: g.query(q, ql, cols, want_ties=True, query_grid=qg), reps)
                tot_old += t_old
                tot_new += t_new
                nq, ns = q.shape[0], g.ns
                alg = 12 * nq + 12 * ns + 8 * nq * cols
                print(f"  {kind}{l}: {nq:6d} x {cols:2d} over {ns:6d}  old {t_old:7.1f} us  new {t_new:7.1f} us  "
                      f"max_count {old[1].tolist()[0]:3d} ties {old[1].tolist()[2]:5d}  "
                      f"{alg / 1e6:6.2f} MB -> {alg / t_new / 1e3:7.1f} GB/s")
        print(f"  sum of the 10 tables: old {tot_old:.0f} us, new {tot_new:.0f} us")


if __name__ == "__main__":
    main()
