"""bench.py -- fragment-pairs/s of the PCR-CG hot path on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one synthetic 3DMatch-shaped pair per GPU (configs[1]:
2 x 30 000 points, indoor hyper-parameters, full-width KPFCNN + GCN, fp32): raw stacked points in pinned host
memory (uploaded inside the timed region; --inputs hbm: already resident) -> point pyramid (3 grid
subsamplings, 10 radius searches) -> KPFCNN+GCN forward -> per-point descriptors / overlap / saliency.  Independent pairs shard across ranks with no data-path
collective: weak scaling, one pair per rank per step.

  python bench.py --gpus N --steps 20 --warmup 3 [--repeats 5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Without a rank environment `--gpus N` (N > 1) makes this process a parent that never touches the GPU: it starts N
fresh rank processes (pcrcg_amd/launcher.py: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, one CPU set per rank near its
GPU), and relays rank 0's line; N above the visible device count exits non-zero.  Under torch.distributed.run the
script is a rank as before.  `--launcher-dry-run` runs the same N-rank protocol over gloo with a stand-in workload
on the CPU (tests/test_launcher_cpu.py).

Rank 0 prints ONE JSON line.  The timed region -- exactly --steps steps between two barrier + synchronize fences,
starting and ending with an empty engine, MAX over ranks -- is run --repeats times back to back; `value` is the MEDIAN
region and every region's figure is listed (a 50-step region lasts ~0.1 s: one shot of it has a few percent of noise).
`roofline` is measured live with HIP start/stop events of every launch of the dominant kernel (the KPConv
neighbour-gather/aggregate kernel) inside those regions; `roofline.gemm` and `roofline.radius` do the same for the GEMM
family and the radius searches in one extra region of the same engine; `secondary` carries bounded runs of configs[4]
(K120k engine) and configs[2] (train step); `cpu_baseline` times the CPU oracle (oracle/: C front end + torch-CPU model, a restatement
pinned against the reference) on a bounded sample on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The pair engine
# uses a front-end stream + three model streams (+ the default stream): with 4 queues two of them share
# one and serialise (measured: 260 vs 338 pairs/s).  Must be set before the first HIP call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from pcrcg_amd import launcher  # noqa: E402  (host logic only: no torch, no HIP)


def _flag_value(argv, name, default):
    for i, a in enumerate(argv):
        if a == name and i + 1 < len(argv):
            return argv[i + 1]
        if a.startswith(name + "="):
            return a.split("=", 1)[1]
    return default


if __name__ == "__main__":
    # BEFORE anything that could initialise the GPU: the parent of an N-rank run only starts children and relays
    try:
        _n = int(_flag_value(sys.argv[1:], "--gpus", "1"))
    except ValueError:
        _n = 1
    if launcher.is_parent(_n):
        sys.exit(launcher.launch(os.path.abspath(__file__), sys.argv[1:], _n,
                                 dry_run="--launcher-dry-run" in sys.argv[1:]))
RANK_CPUS = launcher.apply_rank_affinity()      # a rank started by the launcher: the CPU set planned for its GPU

import numpy as np  # noqa: E402
import torch  # noqa: E402

if RANK_CPUS:                              # pinned to a CPU set: torch's host thread pool must not exceed it
    torch.set_num_threads(max(1, len(RANK_CPUS)))

from pcrcg_amd import indoor_config, kitti_config, ops, synthetic  # noqa: E402
from pcrcg_amd.architectures import KPFCNN  # noqa: E402
from pcrcg_amd.pairstream import PairStreams  # noqa: E402
from pcrcg_amd.sharding import pair_seeds_for_rank  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_TF = 157.3    # dense fp32 matrix peak (MI355X_MICROARCH.md)
MFMA_BF16_TF = 2500.0  # dense bf16 / fp16 matrix peak; the split forms spend 3 (fp16 two-term) or 6 (bf16 three-term) products per fp32 product
INPUT_ORDER = "generator"   # --input-order: "morton" sorts every synthetic cloud along a Z-order curve (never the headline)
RECIPE = "S30k"       # the workload BASELINE.json's metric is quoted on (configs[1]); --workload picks a secondary one
WORKLOADS = {
    "S30k": "S30k: 2x30000-pt shell pairs (3DMatch-shaped), indoor hyper-parameters",
    "C1": "C1 (secondary, configs[0]'s size): 2x5000-pt shell pairs, indoor hyper-parameters",
    "U30k": "U30k (secondary): 2x30000 uniform-random points in a 1.07 m cube, indoor hyper-parameters",
    "K120k": "K120k (secondary, configs[4]): 2x120000-pt KITTI-shaped slabs, KITTI hyper-parameters",
    "T30k": "T30k (secondary): the S30k pairs snapped to a 1/256 m lattice (voxelised-scan-like: most rows hold exactly "
            "equal distances), indoor hyper-parameters, limits calibrated on the first pair",
}


def _morton_order(p):
    q = ((p - p.min(0)) / max(float((p.max(0) - p.min(0)).max()), 1e-9) * 1023).astype(np.int64)
    code = np.zeros(len(p), np.int64)
    for b in range(10):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return np.argsort(code, kind="stable")


def make_pair(recipe, seed):
    if recipe == "U30k":
        pr = synthetic.uniform_pair(30000, 1.07, seed)
    elif recipe == "K120k":
        pr = synthetic.slab_pair(120000, seed)
    else:
        pr = synthetic.pair(recipe, seed)
    if INPUT_ORDER == "morton":      # experiment (--input-order morton): spatially coherent point order, DESIGN.md 5
        pr = tuple(np.ascontiguousarray(c[_morton_order(c)]) for c in pr)
    return pr


def _gemm_mode():
    from pcrcg_amd import _lib
    return int(_lib.lib().pcrcg_gemm_get_mode())


def kpconv_algorithmic_bytes(nq, h, cin, cout, e=4):
    """no-reuse gather model of one KPConv call."""
    return nq * h * (cin * e + 8 + 12) + nq * cout * 4        # e: bytes per stored feature element (2 in the bf16 variant)


def _cpu_front_end(args):
    """One pair through the CPU oracle's front end (worker of the pair-parallel leg)."""
    recipe, seed, cfg, limits = args
    from oracle import frontend as OF
    src, tgt = make_pair(recipe, seed)
    t0 = time.perf_counter()
    OF.oracle_pyramid(np.concatenate([src, tgt]), [len(src), len(tgt)], cfg, limits, tie_order="reference")
    return time.perf_counter() - t0


def _cpu_warm(_):
    """Worker start-up outside the clock: imports, the oracle's shared library."""
    from oracle import frontend as OF
    torch.set_num_threads(1)              # as torch's DataLoader does in its worker processes
    OF.oracle_pyramid(np.random.RandomState(0).rand(2000, 3).astype(np.float32), [1000, 1000],
                      dict(indoor_config()), [8, 8, 8, 8], tie_order="reference")
    return os.getpid()


def cpu_baseline(cfg, state_dict, limits):
    """CPU oracle on a bounded sample: the C front end -- the restatement of the reference's own algorithm (nanoflann
    KD-trees + std::sort: its tables entry for entry) -- on one pair single-threaded and on P pairs in P worker
    processes (the reference's parallelism model is DataLoader worker processes, ref:main.py:82-97), the torch-CPU
    model on all cores.  `value` = 1 / (front end of one pair on one core + model on all cores): one pair's latency
    the way the reference runs it with num_workers = 0; the pair-parallel front-end rate is reported beside it."""
    import multiprocessing as mp
    from oracle import frontend as OF
    from oracle import model_ref as MR
    src, tgt = make_pair(RECIPE, 12345)
    t0 = time.perf_counter()
    batch = OF.oracle_pyramid(np.concatenate([src, tgt]), [len(src), len(tgt)], dict(cfg), limits, tie_order="reference")
    t1 = time.perf_counter()
    MR.kpfcnn_forward(state_dict, dict(cfg), batch)
    t2 = time.perf_counter()
    workers = max(1, min(os.cpu_count() or 1, 16))
    par = None
    try:
        with mp.get_context("spawn").Pool(workers) as pool:
            pool.map(_cpu_warm, range(4 * workers), chunksize=1)          # every worker started and warm
            tp = time.perf_counter()
            pool.map(_cpu_front_end, [(RECIPE, 100 + i, dict(cfg), limits) for i in range(2 * workers)], chunksize=1)
            par = 2 * workers / (time.perf_counter() - tp)
    except Exception as e:       # the baseline is a reported number, never a reason to fail the bench
        par = None
        print("cpu_baseline: pair-parallel leg failed: %r" % (e,), file=sys.stderr)
    return {"value": round(1.0 / (t2 - t0), 4), "unit": "fragment-pairs/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "front_end_pairs_per_s_1_thread": round(1.0 / (t1 - t0), 3),
            "front_end_pairs_per_s_pair_parallel": None if par is None else round(par, 3),
            "front_end_worker_processes": workers,
            "model_pairs_per_s": round(1.0 / (t2 - t1), 3),
            "sample": f"1 {RECIPE} pair: oracle C front end (KD-trees as in the reference) {t1 - t0:.2f}s (1 thread) + torch-CPU model "
                      f"{t2 - t1:.2f}s ({torch.get_num_threads()} threads); front end alone on {2 * workers} pairs in {workers} "
                      f"warm worker processes"}


def kpconv_roofline(events, cout_of):
    """KPConv kernels bracketed by HIP events on their own stream (pcrcg_profile_kpconv): kind 0 = gather/aggregate
    kernel of the two-stage path, 1 = one-kernel KPConv (gather + aggregate + contraction), 2 = bf16-storage gather.
    Algorithmic bytes: SURVEY.md 8d no-reuse gather model of one KPConv call.  (Forwards are enqueued by several
    threads, so records of different pairs interleave: the output width of a gather launch is looked up by its input
    width, which is unique per KPConv in this architecture.)"""
    gather = {"ms": 0.0, "bytes": 0, "n": 0}
    fused = {"ms": 0.0, "bytes": 0, "flops": 0, "n": 0}
    for (ms, nq, h, cin, cout, kind) in events:
        if kind >= 3:
            continue
        co = cout if kind == 1 else cout_of[cin]
        d = fused if kind == 1 else gather
        d["ms"] += ms
        d["bytes"] += kpconv_algorithmic_bytes(nq, h, cin, co, 2 if kind == 2 else 4)
        d["n"] += 1
        if kind == 1:
            d["flops"] += 2 * nq * 15 * cin * (h + co)
    return gather, fused


def gemm_roofline(events, pairs):
    """Every GEMM of the split-bf16 family (kind 3 records: M, N, K, bf16 products per element), timed by its own
    start / stop events: fp32-equivalent TFLOP/s against the fp32 matrix peak and against the split's own ceiling."""
    ms = sum(e[0] for e in events if e[5] == 3)
    flops = sum(2.0 * e[1] * e[2] * e[3] for e in events if e[5] == 3)
    n = sum(1 for e in events if e[5] == 3)
    if ms <= 0 or n == 0:
        return None
    tf = flops / (ms * 1e-3) / 1e12
    # matrix-core work actually issued: e[4] = 16-bit products per fp32 product (3: fp16 two-term form, 6: bf16 three-term form)
    mfma_tf = sum(2.0 * e[1] * e[2] * e[3] * e[4] for e in events if e[5] == 3) / (ms * 1e-3) / 1e12
    return {"launches_per_pair": round(n / max(pairs, 1), 1), "GFLOP_per_pair": round(flops / max(pairs, 1) / 1e9, 1),
            "kernel_ms_per_pair": round(ms / max(pairs, 1), 3), "achieved_TFLOPs": round(tf, 1),
            "frac_of_fp32_mfma_peak_%.1fTF" % MFMA_F32_TF: round(tf / MFMA_F32_TF, 4),
            "products_per_fp32_product": round(mfma_tf / tf, 2),
            "frac_of_16bit_mfma_peak_%.0fTF" % MFMA_BF16_TF: round(mfma_tf / MFMA_BF16_TF, 4)}


def gemm_by_shape(events, forwards):
    """Per (M, N, K): launches per forward, average kernel microseconds, fp32-equivalent TFLOP/s."""
    acc = {}
    for e in events:
        if e[5] == 3:
            a = acc.setdefault((e[1], e[2], e[3], e[4]), [0, 0.0])
            a[0] += 1
            a[1] += e[0]
    rows = [{"m": m, "n": n, "k": k, "products": pr, "per_forward": round(c / max(forwards, 1), 2), "avg_us": round(1e3 * ms / c, 1),
             "us_per_forward": round(1e3 * ms / max(forwards, 1), 1), "TFLOPs": round(2.0 * m * n * k * c / (ms * 1e-3) / 1e12, 1)}
            for (m, n, k, pr), (c, ms) in acc.items()]
    return sorted(rows, key=lambda r: -r["us_per_forward"])


def radius_roofline(events, pairs):
    """Radius searches (kind 4 records: queries, columns, supports, kernel flavour) timed by their own start / stop events;
    algorithmic bytes per SURVEY.md 8d: 12 Nq + 12 Ns + 8 Nq cols per call (every point read once, the int64 table written)."""
    ev = [e for e in events if e[5] == 4]
    ms = sum(e[0] for e in ev)
    if not ev or ms <= 0:
        return None
    by = sum(12 * e[1] + 12 * e[3] + 8 * e[1] * e[2] for e in ev)
    gbs = by / (ms * 1e-3) / 1e9
    return {"launches_per_pair": round(len(ev) / max(pairs, 1), 1), "kernel_ms_per_pair": round(ms / max(pairs, 1), 3),
            "algorithmic_MB_per_pair": round(by / max(pairs, 1) / 1e6, 1), "achieved_GBs": round(gbs, 1),
            "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 5), "avg_launch_us": round(1e3 * ms / len(ev), 1),
            "cell_cooperative_launches": sum(1 for e in ev if e[4] == 1)}


def secondary_train_step(dev, steps, warmup=3):
    """configs[2]: train steps on the 3DLoMatch-shaped S30k pair (full-width model, fp32; pyramid + labels built once):
    forward with tape -> MetricLoss -> backward -> SGD, ms per step, and the same step phase by phase (each drained before
    the next is timed).  Mirrors scripts/bench_train.py / scripts/train_phases.py; ref:lib/trainer.py:216-265."""
    from pcrcg_amd.config import Config
    from pcrcg_amd.correspondences import get_correspondences
    from pcrcg_amd.loss import MetricLoss
    from pcrcg_amd.pyramid import collate_fn_descriptor
    from pcrcg_amd.trainer import LOSS_KEYS, Trainer
    cfg = indoor_config()
    torch.manual_seed(0)
    np.random.seed(0)
    net = KPFCNN(cfg).to(dev)
    loss = MetricLoss(Config(pos_margin=0.1, neg_margin=1.4, pos_radius=0.0375, safe_radius=0.1,
                             matchability_radius=0.05, max_points=256))
    tr = Trainer(net, loss)
    src, tgt, rot, trans = synthetic.lomatch_pair("S30k", 0, overlap=0.2)
    tsfm = np.eye(4)
    tsfm[:3, :3], tsfm[:3, 3] = rot, trans.flatten()
    corr = get_correspondences(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), tsfm, 0.0375)
    item = dict(src_pcd=src, tgt_pcd=tgt, src_feats=np.ones((len(src), 1), np.float32),
                tgt_feats=np.ones((len(tgt), 1), np.float32), rot=rot, trans=trans, correspondences=corr, sample=0)
    inputs = collate_fn_descriptor([item], cfg, synthetic.LIMITS["S30k"], device=dev)
    stats = None
    for _ in range(warmup):
        stats = tr.train_step(inputs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        stats = tr.train_step(inputs)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    ph = {"forward": 0.0, "loss": 0.0, "backward": 0.0, "sgd": 0.0}
    n_ph = 4
    for _ in range(n_ph):
        net.train(True)
        t0 = time.perf_counter()
        out = net.train_runner().forward(inputs)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        len_src = int(inputs["stack_lengths_host"][0][0])
        f = out["feats_f"]
        res = loss({"src_feats": f[:len_src], "tgt_feats": f[len_src:], "rot": inputs["rot"], "trans": inputs["trans"],
                    "scores_overlap": out["scores_overlap"], "scores_saliency": out["scores_saliency"],
                    "src_pcd_raw": inputs["src_pcd_raw"], "tgt_pcd_raw": inputs["tgt_pcd_raw"],
                    "correspondences": inputs["correspondences"]})
        total = sum(res[k] for k in res if k in LOSS_KEYS)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tr.bucket.arm(True)
        total.backward()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        tr.optimizer_step()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        for k, v in zip(ph, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            ph[k] += 1e3 * v / n_ph
    return {"workload": "configs[2]: S30k 3DLoMatch-shaped pair (overlap 0.2, %d correspondences), full-width KPFCNN+GCN, fp32; "
                        "forward with tape + MetricLoss + backward + SGD, pyramid and labels prebuilt" % int(corr.shape[0]),
            "steps": steps, "ms_per_step": round(ms, 2), "train_pairs_per_s": round(1e3 / ms, 2),
            "phases_ms_each_drained": {k: round(v, 2) for k, v in ph.items()},
            "phases_note": "the loss's geometry-only part overlaps the forward inside a real step, so the phases sum to more",
            "last_stats": {k: round(float(v), 4) for k, v in (stats or {}).items()},
            "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}


def secondary_k120k(dev, steps, warmup, workers, ppf, ppb):
    """configs[4] on one GPU: the same engine on 2 x 120 000-point KITTI-shaped slabs with the KITTI hyper-parameters
    (ref:configs/test/kitti.yaml:15-30): pairs/s of one region and the KPConv gather kernels' algorithmic GB/s in it."""
    cfg = kitti_config()
    limits = synthetic.LIMITS["K120k"]
    torch.manual_seed(0)
    np.random.seed(0)
    net = KPFCNN(cfg).eval().to(dev)
    cout_of = {blk.KPConv.in_channels: blk.KPConv.out_channels for blk in net.encoder_blocks}
    pool = []
    for s_ in range(4):
        a, b = synthetic.slab_pair(120000, s_)
        pool.append((torch.from_numpy(np.concatenate([a, b])).pin_memory(), torch.tensor([len(a), len(b)], dtype=torch.int32).pin_memory()))
    pipe = PairStreams(net, cfg, limits, dev, model_streams=workers, front_threads=1, up_nearest=False,
                       pairs_per_forward=ppf, pairs_per_build=ppb)

    def run(count):
        sub = 0
        for i in range(count):
            while sub < min(count, i + 4 * ppb):
                hp, hl = pool[sub % len(pool)]
                pipe.submit(hp.to(dev, non_blocking=True), hl.to(dev, non_blocking=True))
                sub += 1
            pipe.result(wait=False)

    run(2 * workers * ppf)
    pipe.drain()
    run(warmup)
    pipe.drain()
    torch.cuda.synchronize()
    ops.kpconv_profile_start(radius=True)
    t0 = time.perf_counter()
    run(steps)
    pipe.drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev = ops.kpconv_profile_stop()
    pipe.close()
    g, _ = kpconv_roofline([e for e in ev if e[5] != 4], cout_of)
    gbs = g["bytes"] / (g["ms"] * 1e-3) / 1e9 if g["ms"] > 0 else 0.0
    return {"workload": WORKLOADS["K120k"], "steps": steps, "value": round(steps / dt, 2), "unit": "fragment-pairs/s",
            "ms_per_step": round(1e3 * dt / steps, 2), "limits": limits,
            "kpconv_gather_in_engine": {"achieved_GBs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                                        "avg_launch_us": round(1e3 * g["ms"] / max(g["n"], 1), 1)},
            "radius_search_in_engine": radius_roofline(ev, steps)}


def secondary_image129(dev, steps, warmup, workers, ppf, ppb, isolated=6):
    """PCR-CG's SHIPPED configuration (ref:configs/test/indoor.yaml:21-34: image_feature True, img_num 2, in_feats_dim 129;
    injection ref:models/architectures.py:195-514) through the same engine: S30k pairs with synthetic 2-D inputs
    (pcrcg_amd.synthetic.image_inputs: two 128 x 120 x 160 feature maps per cloud, 45 % of the points projected per image,
    valid masks; the ResUNet that produces the maps on real data is the caller's and outside the clock -- the maps are
    resident in HBM like the ResUNet's outputs would be).  Inside the clock: points uploaded from pinned host memory,
    pyramid build, injection of the [N, 129] input, KPFCNN+GCN forward.  Reported with it: the FIRST KPConv's gather kernel
    (60 000 queries x 43 neighbours x 129 channels, rows of 132 floats) against the HBM roofline, alone and in the engine
    -- the dominant kernel of this configuration (1.38 GB by the SURVEY.md 8d formula against 2.37 GB for the whole
    geometry-only pair)."""
    cfg = indoor_config(image_feature=True, img_num=2, in_feats_dim=129)
    limits = synthetic.LIMITS["S30k"]
    torch.manual_seed(0)
    np.random.seed(0)
    net = KPFCNN(cfg).eval().to(dev)
    cout_of = {blk.KPConv.in_channels: blk.KPConv.out_channels for blk in net.encoder_blocks}

    def real_width(ev):      # the first layer's launches report the row width (132 floats); the algorithmic bytes count 129 channels
        return [(e[0], e[1], e[2], 129 if e[3] == 132 else e[3], e[4], e[5]) for e in ev]
    pool = []
    for s_ in range(4):
        a, b = synthetic.pair("S30k", s_)
        im = synthetic.image_inputs(len(a), len(b), s_, img_num=2)
        batch_like = {k: torch.from_numpy(v).to(dev) for k, v in im.items()}
        batch_like["points"] = [torch.empty(len(a) + len(b), 3, device=dev)]
        batch_like["src_pcd_raw"] = torch.empty(len(a), 3)
        _, _, images = net.image_list(batch_like)
        pool.append((torch.from_numpy(np.concatenate([a, b])).pin_memory(), torch.tensor([len(a), len(b)], dtype=torch.int32).pin_memory(),
                     images))

    def first_layer(ev):
        rows = [e for e in ev if e[5] == 0 and e[3] == 132]
        if not rows:
            return None
        ms = sum(e[0] for e in rows) / len(rows)
        by = kpconv_algorithmic_bytes(rows[0][1], rows[0][2], 129, cout_of[129], 4)
        return {"nq": rows[0][1], "h": rows[0][2], "cin": 129, "row_floats": 132, "cout": cout_of[129], "launches": len(rows),
                "avg_launch_us": round(1e3 * ms, 1), "algorithmic_MB": round(by / 1e6, 1), "achieved_GBs": round(by / (ms * 1e-3) / 1e9, 1),
                "frac_of_hbm_peak": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # the forward alone on one stream: what the first layer's gather does with the GPU to itself
    from pcrcg_amd.pyramid import build_pyramid
    hp, hl, images = pool[0]
    b0 = build_pyramid(hp.to(dev), hl.to(dev), cfg, limits)
    b0["features"] = ops.inject_image_features(int(hp.shape[0]), int(hl[0]), images, channels=128, width=net.IMAGE_WIDTH)
    runner = net.runner()
    with torch.no_grad():
        for _ in range(2):
            runner.forward(b0)
        torch.cuda.synchronize()
        ops.kpconv_profile_start()
        t0 = time.perf_counter()
        for _ in range(isolated):
            runner.forward(b0)
        torch.cuda.synchronize()
        iso_ms = 1e3 * (time.perf_counter() - t0) / isolated
    ev_iso = ops.kpconv_profile_stop()
    g_iso, _ = kpconv_roofline(real_width(ev_iso), cout_of)
    del b0
    pipe = PairStreams(net, cfg, limits, dev, model_streams=workers, front_threads=1, up_nearest=False,
                       pairs_per_forward=ppf, pairs_per_build=ppb)

    def run(count):
        sub = 0
        for i in range(count):
            while sub < min(count, i + 6 * ppb):
                hp, hl, images = pool[sub % len(pool)]
                pipe.submit(hp.to(dev, non_blocking=True), hl.to(dev, non_blocking=True), images=images)
                sub += 1
            pipe.result(wait=False)

    run(3 * workers * ppf)
    pipe.drain()
    run(warmup)
    pipe.drain()
    torch.cuda.synchronize()
    ops.kpconv_profile_start()
    t0 = time.perf_counter()
    run(steps)
    pipe.drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev = ops.kpconv_profile_stop()
    pipe.close()
    g, _ = kpconv_roofline(real_width(ev), cout_of)
    gbs = g["bytes"] / (g["ms"] * 1e-3) / 1e9 if g["ms"] > 0 else 0.0
    return {"workload": "S30k-img129 (secondary; PCR-CG's shipped configuration): S30k pairs + synthetic 2-D inputs (2 images per cloud, "
                        "128 x 120 x 160 maps resident in HBM, 45 % of the points projected per image), image_feature=True, img_num=2, "
                        "in_feats_dim=129; pyramid build + feature injection + KPFCNN+GCN forward",
            "steps": steps, "value": round(steps / dt, 2), "unit": "fragment-pairs/s", "ms_per_step": round(1e3 * dt / steps, 3),
            "forward_alone_ms": round(iso_ms, 3), "engine_streams_by_dispatcher": pipe.pipe_classes,
            "first_kpconv_gather": {"alone": first_layer(ev_iso), "in_engine": first_layer(ev)},
            "kpconv_gathers_all_layers": {"alone_GBs": round(g_iso["bytes"] / (g_iso["ms"] * 1e-3) / 1e9, 1) if g_iso["ms"] > 0 else None,
                                          "in_engine_GBs": round(gbs, 1), "in_engine_frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                                          "algorithmic_bytes_per_pair": int(g["bytes"] / max(steps, 1))}}


def dry_run_main(args, rank, world):
    """--launcher-dry-run: the N-rank protocol of this file -- process group from the launcher's environment, fences,
    EXACTLY --steps steps per region, MAX over ranks, rank 0's single line with `ranks_seen` and per-rank values --
    over gloo on the CPU.  The stand-in step is a digest of the rank's own synthetic pair (sharding as in the real
    run); the line says `dry_run: true` and is not a measurement."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29518")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    seeds = pair_seeds_for_rank(args.warmup + args.steps * max(1, args.repeats), rank, world)
    cursor, digest = [0], [0.0]

    def step():
        src, tgt = synthetic.pair("mini", seeds[cursor[0] % len(seeds)])
        cursor[0] += 1
        digest[0] += float(src.sum() + tgt.sum())

    for _ in range(args.warmup):
        step()
    regions = []
    for _ in range(max(1, args.repeats)):
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        dist.barrier()
        own = time.perf_counter() - t0
        t = torch.tensor([own], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        regions.append((float(t.item()), own))
    times = sorted(r[0] for r in regions)
    elapsed = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
    own = sorted(r[1] for r in regions)[len(regions) // 2]
    per_rank = launcher.rank_fields(dist, world, rank, round(args.steps / own, 3), RANK_CPUS)
    seen_seeds = [None] * world
    dist.all_gather_object(seen_seeds, seeds[:args.steps])
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({
            "metric": "fragment-pairs/s KPFCNN+GCN fwd, 30k-pt pairs", "value": round(args.steps * world / elapsed, 3),
            "unit": "fragment-pairs/s", "n_gpus": world, "ranks_seen": per_rank["ranks_seen"],
            "per_rank_pairs_per_s": per_rank["per_rank_value"], "per_rank_cpus": per_rank["per_rank_cpus"],
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dry_run": True,
            "config": {"workload": "launcher dry run: stand-in CPU step (digest of the rank's own `mini` pair) over gloo; "
                                   "NOT a measurement",
                       "parallelism": f"pairs sharded over {world} rank(s), no data-path collective",
                       "pair_seeds_first_region": seen_seeds}}), flush=True)


def live_pmc_traffic(timeout_s=90):
    """HBM bytes per KPConv gather launch from the PMC counters, collected NOW: two child runs of this script (`--isolated-only`,
    three forwards of one prepared pair, nothing else on the GPU) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and
    `... --pmc WRITE_SIZE` -- separate passes with the kernel trace only, as MI355X_MICROARCH.md prescribes (the two
    counters do not fit one pass), read with its gfx950 correction: fetch bytes = 2 x FETCH_SIZE[KiB] x 1024 (FETCH_SIZE
    tallies 128-byte requests at 64 bytes), write bytes = WRITE_SIZE[KiB] x 1024.  The children are started as ordinary
    child processes (this process keeps its GPU context and waits).  -> (bytes per launch or None, detail dict)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("PCRCG_BENCH_CHILD") or any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return None, {"skipped": "already running under a profiler"}
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, {"skipped": "rocprofv3 not found"}
    tot = {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]}
    t0 = time.perf_counter()
    for counter in tot:
        tmp = tempfile.mkdtemp(prefix="pcrcg_pmc_", dir="/tmp")
        env = dict(os.environ, PCRCG_BENCH_CHILD="1", TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", "p", "--",
               sys.executable, os.path.abspath(__file__), "--isolated-only", "--steps", "3", "--warmup", "1", "--workload", RECIPE]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
            files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, {"failed": f"{counter} pass: rc {r.returncode}, {len(files)} counter file(s)",
                              "stderr_tail": r.stderr.decode(errors="replace")[-300:]}
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] == counter and "k_kpconv" in row["Kernel_Name"]:
                    tot[counter][0] += 1
                    tot[counter][1] += float(row["Counter_Value"])
        except subprocess.TimeoutExpired:
            return None, {"failed": f"{counter} pass: no result within {timeout_s} s"}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    nf, nw = tot["FETCH_SIZE"][0], tot["WRITE_SIZE"][0]
    if not nf or not nw:
        return None, {"failed": "no KPConv gather launch in the counter files"}
    fetch = 2.0 * tot["FETCH_SIZE"][1] / nf * 1024.0
    write = tot["WRITE_SIZE"][1] / nw * 1024.0
    return int(fetch + write), {"fetch_bytes_per_launch": int(fetch), "write_bytes_per_launch": int(write), "launches_counted": nf,
                                "passes": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE, one child run each of "
                                          "`bench.py --isolated-only --steps 3 --warmup 1` (the launches alone on the GPU)",
                                "correction": "fetch = 2 x FETCH_SIZE[KiB] x 1024 (gfx950), write = WRITE_SIZE[KiB] x 1024",
                                "seconds": round(time.perf_counter() - t0, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=480)     # (whole groups of four pairs on three model streams; 0.8 s regions: fill and drain of the engine are ~1 % of them)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps steps each, run back to back; `value` is the median region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary regions (GEMM / radius events, HBM-resident inputs, one-column upsample tables, "
                         "the configs[2] train step, the configs[4] K120k engine)")
    ap.add_argument("--inputs", choices=["host", "hbm"], default="host",
                    help="host (default, SURVEY.md 8d's protocol): every pair is uploaded from pinned host memory inside the "
                         "timed region; hbm: inputs resident in HBM before the region starts")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="S30k")
    ap.add_argument("--input-order", choices=["generator", "morton"], default="generator",
                    help="experiment: Morton-sort each synthetic cloud (spatially coherent indices); recorded in config")
    ap.add_argument("--model-streams", type=int, default=3, help="engine: host threads / HIP streams enqueueing forwards")
    ap.add_argument("--front-threads", type=int, default=1, help="engine: host threads building pyramids")
    ap.add_argument("--depth", type=int, default=None,
                    help="pairs submitted ahead of the one being collected (default: six builds' worth -- 24, K120k 12)")
    ap.add_argument("--pairs-per-forward", type=int, default=None, choices=[1, 2, 3, 4],
                    help="pairs built together that share one pcrcg_kpfcnn_forward_group call (weight products once for all); "
                         "default 4 (K120k: 3), profiles/r05_ab_group_size.txt")
    ap.add_argument("--pairs-per-build", type=int, default=None, choices=[1, 2, 3, 4, 5, 6, 7, 8],
                    help="pairs one front-end kernel chain (pcrcg_pyramid_build call) carries; default 4 (K120k: 3)")
    ap.add_argument("--adaptive-jobs", action="store_true",
                    help="engine: decide from queue / stream state whether the pairs of a build share a forward call (round 4's "
                         "default; timing-dependent grouping, no throughput gain any more)")
    ap.add_argument("--front-priority", type=int, default=0, help="engine: HIP priority of the front-end stream (< 0: ahead of the forwards)")
    ap.add_argument("--front-streams", type=int, default=1, help="engine: front-end HIP streams (front threads take them in turn)")
    ap.add_argument("--fixed-jobs", action="store_true",
                    help="(the default since round 5, kept for old command lines) forward jobs always carry --pairs-per-forward pairs")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs behind roofline.traffic (traffic: null, ~30 s less)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="A/B aid: no start/stop events on the KPConv launches of the timed regions (roofline comes out empty)")
    ap.add_argument("--isolated-only", action="store_true",
                    help="no engine: --steps forwards of one prepared pair on one stream, nothing else running (the "
                         "run rocprofv3 is pointed at for the kernels' isolated durations and PMC traffic)")
    ap.add_argument("--variant", choices=["fp32", "bf16"], default="fp32",
                    help="bf16: the bf16 feature-storage VARIANT (pcrcg_model.feature_bf16) -- a separate line with its "
                         "measured error against the fp32 path; never the headline")
    ap.add_argument("--image129-only", action="store_true",
                    help="run only the S30k-img129 secondary leg (PCR-CG's shipped 129-channel configuration) and print its record")
    ap.add_argument("--launcher-dry-run", action="store_true",
                    help="the N-rank protocol (launcher, process group, fences, MAX over ranks, rank 0's line) over gloo "
                         "on the CPU with a stand-in workload: what tests/test_launcher_cpu.py runs; never a measurement")
    args = ap.parse_args()
    global RECIPE, INPUT_ORDER
    RECIPE = args.workload
    INPUT_ORDER = args.input_order
    # group sizes of the engine: four pairs per front-end chain and per forward call, three for the 2 x 120 000-point slabs
    group = 3 if RECIPE == "K120k" else 4
    if args.pairs_per_forward is None:
        args.pairs_per_forward = group
    if args.pairs_per_build is None:
        args.pairs_per_build = max(group, args.pairs_per_forward)
    if args.depth is None:
        args.depth = 6 * args.pairs_per_build

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; n_gpus reports the ranks that exist" % (args.gpus, world),
              file=sys.stderr)
    if args.launcher_dry_run:
        return dry_run_main(args, rank, world)
    if local >= torch.cuda.device_count():
        print("bench.py: rank %d wants device %d, %d visible" % (rank, local, torch.cuda.device_count()), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("PCRCG_FORCE_DIST") == "1":   # the latter: exercise the RCCL path on 1 GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)

    cfg = kitti_config() if RECIPE == "K120k" else indoor_config()
    limits = synthetic.LIMITS.get(RECIPE)
    if limits is None:      # the reference's calibration formula on the first pair (ref:datasets/dataloader.py:402-434)
        from pcrcg_amd.pyramid import calibrate_neighbors
        a, b = make_pair(RECIPE, 0)
        limits = [int(v) for v in calibrate_neighbors(
            [(torch.from_numpy(np.concatenate([a, b])).to(dev), torch.tensor([len(a), len(b)], dtype=torch.int32, device=dev))],
            cfg, samples_threshold=0)]
    torch.manual_seed(0)
    np.random.seed(0)
    net = KPFCNN(cfg).eval()
    state_dict = {k: v.clone() for k, v in net.state_dict().items()} if rank == 0 else None
    net = net.to(dev)
    BF16 = args.variant == "bf16"
    FE = 2 if BF16 else 4                 # bytes per stored feature element in the KPConv gathers
    R = max(1, args.repeats)

    # synthetic inputs; every step sees a different pair (16 distinct pairs per rank, cycled).  `pool`: resident in HBM
    # before the timed regions (the headline).  `host_pool`: the same pairs in pinned host memory, for the one extra
    # region that uploads every pair inside the clock (PCIe-inclusive figure, never `value`).
    total = 2 * args.warmup + args.steps * (R + 6) + 64
    seeds = pair_seeds_for_rank(total, rank, world)
    pool, host_pool = {}, {}
    for s in sorted(set(x % 16 for x in seeds)):
        src, tgt = make_pair(RECIPE, s)
        pts = torch.from_numpy(np.concatenate([src, tgt]))
        lens = torch.tensor([len(src), len(tgt)], dtype=torch.int32)
        host_pool[s] = (pts.pin_memory(), lens.pin_memory())
        pool[s] = (pts.to(dev), lens.to(dev))

    variant_error = None
    if BF16:
        # the variant's error against the fp32 path on the first pair, measured before anything is timed
        from pcrcg_amd.pyramid import build_pyramid
        b0 = build_pyramid(*pool[seeds[0] % 16], cfg, limits)
        with torch.no_grad():
            o32 = {k: v.double() for k, v in net(b0).items()}
            net.feature_bf16 = True
            o16 = {k: v.double() for k, v in net(b0).items()}
        variant_error = {k: float((o16[k] - o32[k]).abs().max() / o32[k].abs().max()) for k in o32}
        del b0, o32, o16

    cout_of = {blk.KPConv.in_channels: blk.KPConv.out_channels for blk in net.encoder_blocks}
    assert all(cout_of[blk.KPConv.in_channels] == blk.KPConv.out_channels for blk in net.encoder_blocks)
    per_pair = len(net.encoder_blocks)

    if args.image129_only:
        print(json.dumps(secondary_image129(dev, args.steps, args.warmup, args.model_streams, args.pairs_per_forward,
                                            args.pairs_per_build)), flush=True)
        return
    if args.isolated_only:
        from pcrcg_amd.pyramid import build_pyramid
        batch_iso = build_pyramid(*pool[seeds[0] % 16], cfg, limits)
        with torch.no_grad():
            for _ in range(args.warmup):
                net(batch_iso)
            torch.cuda.synchronize()
            ops.kpconv_profile_start(gemm=True)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                net(batch_iso)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        ev = ops.kpconv_profile_stop()
        g, f = kpconv_roofline(ev, cout_of)
        ms, by = g["ms"] + f["ms"], g["bytes"] + f["bytes"]
        print(json.dumps({"mode": "isolated-only", "workload": RECIPE, "variant": args.variant, "forwards": args.steps,
                          "forward_ms": round(1e3 * dt / args.steps, 3),
                          "kpconv_launches": g["n"] + f["n"], "kpconv_avg_launch_us": round(1e3 * ms / max(g["n"] + f["n"], 1), 2),
                          "kpconv_algorithmic_GBs": round(by / (ms * 1e-3) / 1e9, 1),
                          "kpconv_frac_of_hbm_peak": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "gemm": gemm_roofline(ev, args.steps), "gemm_by_shape": gemm_by_shape(ev, args.steps)}), flush=True)
        return

    # Pair engine (pcrcg_amd/pairstream.py): a front thread builds pyramids on the front-end stream, WORKERS threads with
    # one HIP stream each enqueue the forwards; pairs are submitted up to DEPTH ahead.
    WORKERS, FRONTS, DEPTH = args.model_streams, args.front_threads, args.depth
    cursor = [0]

    def run_pairs(pipe, count, from_host=False):
        """Push `count` pairs through the engine, at most DEPTH in flight.  from_host: every pair is uploaded from pinned
        host memory (two async copies on the caller's stream) inside this call."""
        out, submitted, first = None, 0, cursor[0]
        cursor[0] += count
        for i in range(count):
            while submitted < min(count, i + DEPTH):
                key = seeds[(first + submitted) % len(seeds)] % 16
                if from_host:
                    hp, hl = host_pool[key]
                    pipe.submit(hp.to(dev, non_blocking=True), hl.to(dev, non_blocking=True))
                else:
                    pipe.submit(*pool[key])
                submitted += 1
            out = pipe.result(wait=False)         # (throughput loop: see PairStreams.result)
        return out

    def fence(pipe):
        pipe.drain()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    HEAD_FROM_HOST = args.inputs == "host"

    def region(pipe, from_host=HEAD_FROM_HOST):
        """EXACTLY --steps steps from an empty engine to an empty engine; -> seconds (MAX over ranks), submit seconds."""
        fence(pipe)
        t0 = time.perf_counter()
        out = run_pairs(pipe, args.steps, from_host)
        submit = time.perf_counter() - t0      # host time until the last forward was enqueued (GPU may still be busy)
        fence(pipe)
        elapsed = time.perf_counter() - t0
        assert out["feats_f"].shape[1] == cfg.final_feats_dim
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), submit, elapsed

    # ---- headline: the reference's batch contract inside the engine too (full [N, limit] upsample tables) ----------
    pipe = PairStreams(net, cfg, limits, dev, model_streams=WORKERS, front_threads=FRONTS, up_nearest=False,
                       pairs_per_forward=args.pairs_per_forward, pairs_per_build=args.pairs_per_build,
                       front_priority=args.front_priority, front_streams=args.front_streams,
                       adaptive_jobs=args.adaptive_jobs and not args.fixed_jobs)
    ENGINE_PIPES = pipe.pipe_classes
    # engine priming (untimed, before the W warm-up steps): every model stream's first call allocates its workspace
    # and every front-end arena its tables; a handful of pairs per stream gets that out of the way
    run_pairs(pipe, 4 * WORKERS * args.pairs_per_forward)
    fence(pipe)
    run_pairs(pipe, args.warmup)
    fence(pipe)
    pipe.reset_stats()
    if not args.no_kernel_events:
        ops.kpconv_profile_start()        # HIP events around every KPConv gather/aggregate launch from here on
    cpu0 = os.times()
    regions = [region(pipe) for _ in range(R)]
    cpu1 = os.times()
    # user + system CPU seconds of THIS process (the submitting thread, the front thread, the model threads, the runtime's
    # own threads) per pair over the timed regions: what a rank costs its host -- eight ranks need eight times that
    cpu_s_per_pair = ((cpu1.user - cpu0.user) + (cpu1.system - cpu0.system)) / max(R * args.steps, 1)
    events = ops.kpconv_profile_stop()
    stats = pipe.stats_snapshot()
    own = sorted(r[2] for r in regions)[len(regions) // 2]      # this rank's own clock, median region
    per_rank = launcher.rank_fields(dist, world, rank, round(args.steps / own, 3), RANK_CPUS, device=dev)
    times = sorted(r[0] for r in regions)
    elapsed = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
    submit = sorted(r[1] for r in regions)[len(regions) // 2]

    extras = {}
    gemm_engine = gemm_iso = iso = radius_engine = radius_iso = None
    if not args.no_extras:
        # (a) the GEMM family and the radius searches inside the engine: one more region with start / stop events on
        # those launches as well
        ops.kpconv_profile_start(gemm=True, radius=True)
        t_g = region(pipe)[0]
        ev_g = ops.kpconv_profile_stop()
        gemm_engine = gemm_roofline(ev_g, args.steps)
        radius_engine = radius_roofline(ev_g, args.steps)
        if gemm_engine:
            gemm_engine["pairs_per_s_of_this_region"] = round(args.steps * world / t_g, 1)
        # (b) the other input hand-over: resident in HBM before the region starts (or, with --inputs hbm, pinned host)
        t_h = region(pipe, from_host=not HEAD_FROM_HOST)[0]
        extras["hbm_resident_inputs" if HEAD_FROM_HOST else "pinned_host_inputs"] = {
            "value": round(args.steps * world / t_h, 3), "unit": "fragment-pairs/s",
            "note": "one region with the OTHER hand-over: " + ("points + lengths already in HBM when the region starts" if HEAD_FROM_HOST
                    else "every pair's points + lengths (720 KB) copied from pinned host memory inside the region")}
    # the same kernels once more WITHOUT any concurrent stream: the kernels in isolation
    if rank == 0:
        from pcrcg_amd.pyramid import build_pyramid
        batch_iso = build_pyramid(*pool[seeds[0] % 16], cfg, limits)
        pipe.synchronize()
        ops.kpconv_profile_start(gemm=True)
        with torch.no_grad():
            for _ in range(3):
                net(batch_iso)
        torch.cuda.synchronize()
        iso = ops.kpconv_profile_stop()
        gemm_iso = gemm_roofline(iso, 3)
        iso = [e for e in iso if e[5] < 3]
        # the front end alone: three pyramid builds on an idle GPU, the radius kernels' own events
        ops.kpconv_profile_start(radius=True, kpconv=False)
        for _ in range(3):
            build_pyramid(*pool[seeds[0] % 16], cfg, limits)
        torch.cuda.synchronize()
        radius_iso = radius_roofline(ops.kpconv_profile_stop(), 3)
    if os.environ.get("PCRCG_PIPE_STATS") and rank == 0:       # (verbose host-side engine statistics on stderr)
        n = max(stats["pairs"], 1)
        print("pair engine, host ms per pair: " + ", ".join("%s %.3f" % (k[:-2], 1e3 * v / n) for k, v in stats.items()
                                                            if k not in ("pairs", "builds"))
              + "; %.2f pairs per build" % (n / max(stats.get("builds", n), 1)), file=sys.stderr, flush=True)
    if not args.no_extras:
        # (c) the same engine with ONE-column upsample tables (the nearest coarse point is all KPFCNN.forward reads of
        # them, ref:models/blocks.py:77-87): not the batch contract, so not the headline
        fence(pipe)
        pipe.set_up_nearest(True)
        run_pairs(pipe, args.warmup)
        t_u = sorted(region(pipe)[0] for _ in range(min(R, 3)))
        extras["one_column_upsample_tables"] = {"value": round(args.steps * world / t_u[len(t_u) // 2], 3),
                                                "unit": "fragment-pairs/s", "regions": len(t_u)}
    pipe.close()
    if not args.no_extras and rank == 0 and world == 1 and RECIPE == "S30k" and not BF16:
        # (d) configs[4] and configs[2] on this GPU, bounded (a dozen steps each): driver-visible numbers for both
        del pipe
        legs = os.environ.get("PCRCG_BENCH_LEGS", "K120k,train_step,S30k_img129").split(",")     # (a measurement aid: which legs, in which order)
        for leg in legs:
            if os.environ.get("PCRCG_BENCH_KEEP_CACHE") != "1":
                torch.cuda.empty_cache()
            try:       # secondary figures never fail the headline
                if leg == "K120k":
                    extras[leg] = secondary_k120k(dev, 96, 6, WORKERS, 3, 3)
                elif leg == "train_step":
                    extras[leg] = secondary_train_step(dev, 12)
                elif leg == "S30k_img129":
                    extras[leg] = secondary_image129(dev, 96, 8, WORKERS, 4, 4)
            except Exception as e:
                import traceback
                extras[leg] = {"error": repr(e), "traceback": traceback.format_exc()[-1500:]}
                print("bench.py: secondary leg %s failed: %r" % (leg, e), file=sys.stderr)

    if rank == 0:
        gather, fused = kpconv_roofline(events, cout_of)
        k_bytes = gather["bytes"] + fused["bytes"]
        k_ms = gather["ms"] + fused["ms"]
        n_launch = gather["n"] + fused["n"]
        achieved = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        g_gbs = gather["bytes"] / (gather["ms"] * 1e-3) / 1e9 if gather["ms"] > 0 else 0.0
        ig, if_ = kpconv_roofline(iso, cout_of)
        iso_ms, iso_bytes = ig["ms"] + if_["ms"], ig["bytes"] + if_["bytes"]
        iso_gbs = iso_bytes / (iso_ms * 1e-3) / 1e9 if iso_ms > 0 else 0.0
        iso_rows = []                        # the last isolated forward, launch by launch
        for (ms, nq, h, cin, cout, kind) in iso[-per_pair:]:
            co = cout if kind == 1 else cout_of[cin]
            b = kpconv_algorithmic_bytes(nq, h, cin, co, 2 if kind == 2 else 4)
            iso_rows.append({"nq": nq, "h": h, "cin": cin, "cout": co, "kind": kind, "us": round(ms * 1e3, 1),
                             "GBs": round(b / (ms * 1e-3) / 1e9, 0) if ms > 0 else None})
        # `traffic` (HBM bytes per launch from PMC counters): rocprofv3 --pmc needs its own passes, so this process runs them
        # as two child runs once its own measurements are over (live_pmc_traffic).  The committed separate passes of the
        # round are quoted next to it with their source.
        traffic, traffic_detail, traffic_offline = None, {"skipped": "--no-pmc / --no-extras / N > 1 / not the S30k fp32 line"}, None
        if not args.no_pmc and not args.no_extras and world == 1 and RECIPE == "S30k" and not BF16:
            traffic, traffic_detail = live_pmc_traffic()
        for name in ("r05_pmc_kpconv.json", "r04_pmc_kpconv.json", "r03_pmc_kpconv.json"):
            pmc_path = os.path.join(REPO, "profiles", name)
            if os.path.exists(pmc_path):
                traffic_offline = {"hbm_bytes_per_launch": json.load(open(pmc_path)).get("hbm_bytes_per_launch"),
                                   "source": f"profiles/{name}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                             "`bench.py --isolated-only`, corrected as MI355X_MICROARCH.md prescribes; NOT measured by this run"}
                break
        from pcrcg_amd import _lib
        lib_path, lib_sha = _lib.lib_identity()
        ppb = stats["pairs"] / max(stats.get("builds", 1), 1)
        tie = os.environ.get("PCRCG_TIE_ORDER", "auto")
        line = {
            "metric": ("fragment-pairs/s KPFCNN+GCN fwd, 30k-pt pairs" if RECIPE == "S30k"
                       else f"fragment-pairs/s KPFCNN+GCN fwd, {RECIPE} pairs (secondary workload)")
                      + (" [bf16 feature-storage VARIANT, not the fp32 headline]" if BF16 else ""),
            "value": round(args.steps * world / elapsed, 3),
            "unit": "fragment-pairs/s",
            "n_gpus": world,
            "ranks_seen": per_rank["ranks_seen"],
            "per_rank_pairs_per_s": per_rank["per_rank_value"],
            "per_rank_cpus": per_rank["per_rank_cpus"],
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "host_submit_ms_per_step": round(submit / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16 feature storage in the KPConv gathers, f32 weights / accumulation / outputs" if BF16 else "f32",
            "data": "synthetic",
            "repeats": {"regions": R, "steps_per_region": args.steps, "statistic": "median",
                        "pairs_per_s": [round(args.steps * world / r[0], 1) for r in regions],
                        "min": round(args.steps * world / max(times), 1), "max": round(args.steps * world / min(times), 1)},
            "config": {"workload": WORKLOADS[RECIPE] + ("" if INPUT_ORDER == "generator" else " [clouds Morton-sorted: an "
                                                        "experiment, not the headline workload]") + ", pyramid build + KPFCNN+GCN forward, random-init full-width weights, "
                                   "1 pair/GPU/step, " + ("inputs in pinned host memory, uploaded (two async copies, 720 KB) inside the "
                                   "timed region (SURVEY.md 8d)" if HEAD_FROM_HOST else "inputs resident in HBM") + "; pair engine: %d front thread(s) build pyramids "
                                   "(pcrcg_pyramid_build, %.2f pairs per call on average: the pairs that are waiting, up to four, share one kernel "
                                   "chain) on one front-end HIP stream, %d host threads enqueue the forwards "
                                   "(pcrcg_kpfcnn_forward_group: %s) on one model stream each; every table as the batch contract "
                                   "defines it ([N, limit] upsample tables included); every timed region starts and ends "
                                   "with an empty engine; neighbour tables in the reference's own order inside groups of "
                                   "exactly equal distance (tie_order=%s); GEMM arithmetic mode %d (1 = fp32 operands split "
                                   "into 16-bit terms on the matrix cores, fp32-class accuracy: the two-term fp16 form, three products, with "
                                   "the exact three-term bf16 form, six products, for any tile that leaves fp16's range; 0 = fp32 MFMA)"
                                   % (FRONTS, ppb, WORKERS,
                                      "up to %d pairs of a build in ONE call, every weight product once for all of them"
                                      % args.pairs_per_forward if args.pairs_per_forward >= 2 else "one call per pair",
                                      tie, _gemm_mode()),
                       "inputs": "pinned host memory, uploaded inside the timed region" if HEAD_FROM_HOST else "resident in HBM",
                       "tie_order": tie, "up_nearest": 0, "pairs_per_pyramid_build": round(ppb, 2),
                       "pairs_per_forward_call": args.pairs_per_forward,
                       "engine_streams_by_dispatcher": ENGINE_PIPES,
                       "limits": limits, "parallelism": f"pairs sharded over {world} GPU(s), no data-path collective",
                       "lib_path": os.path.relpath(lib_path, REPO), "lib_sha16": lib_sha},
            "secondary": extras,
            "roofline": {"bound": "hbm", "kernel": "KPConv neighbour-gather / aggregate kernels (k_kpconv_mfma, k_kpconv_c1), "
                                                     "%d launches/pair" % per_pair,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_live": traffic_detail, "traffic_offline": traffic_offline,
                         "avg_launch_us": round(k_ms * 1e3 / max(n_launch, 1), 2),
                         "algorithmic_bytes_per_pair": int(k_bytes / max(args.steps * R, 1)),
                         "note": "durations are the kernels' own start/stop events (hipExtLaunchKernel); achieved/frac "
                                 "are measured inside the timed regions, where several HIP streams share the GPU; `isolated` "
                                 "is the same launches run alone right after them.  Under rocprofv3 the engine runs ~7 % slower "
                                 "and its kernels overlap less, so the profiler's table (profiles/*_kernel_stats.csv, which "
                                 "states the in-engine and the isolated launches separately) shows SHORTER in-engine launches "
                                 "(63 us; the event pairs of that same profiled run: 73 us) than an unprofiled run's event pairs "
                                 "(82 us): frac here is the unprofiled, conservative figure; alone the two clocks agree",
                         "isolated": {"achieved": round(iso_gbs, 1), "frac": round(iso_gbs / HBM_PEAK_GBS, 4),
                                      "avg_launch_us": round(iso_ms * 1e3 / max(len(iso), 1), 2),
                                      "per_launch": iso_rows},
                         "gather_only_kernels": {"launches_per_pair": gather["n"] // max(args.steps * R, 1),
                                                 "achieved_GBs": round(g_gbs, 1),
                                                 "frac": round(g_gbs / HBM_PEAK_GBS, 4)},
                         "radius": {"bound": "hbm", "kernel": "k_radius_cells (cell-cooperative, LDS-staged; the pyramid's ten "
                                              "searches per pair), algorithmic bytes 12 Nq + 12 Ns + 8 Nq cols per call",
                                    "in_engine": radius_engine, "isolated": radius_iso,
                                    "note": "integer / latency work: the algorithmic bytes are tiny against what the kernel "
                                            "does per query (27-64 hash probes per CELL, ~130 candidates swept and ~36 hits "
                                            "rank-sorted per query), so the fraction is small by construction; durations are "
                                            "the kernels' own events (in_engine: beside three model streams)"},
                         "gemm": {"bound": "mfma", "kernel": "k_gemm_x6 family (every C = A * B^T product of the forward)",
                                  "unit": "TFLOP/s (fp32-equivalent: 2*M*N*K per launch)",
                                  "in_engine": gemm_engine, "isolated": gemm_iso,
                                  "note": "in_engine: one extra region of the same engine with start/stop events on every "
                                          "GEMM launch as well -- GEMMs of three model streams overlap there (kernel_ms_per_pair "
                                          "can exceed ms_per_step), so in-engine durations are stretched by the concurrent "
                                          "streams and are NOT roofline evidence; isolated: three forwards alone on one stream"}},
        }
        n = max(stats["pairs"], 1)
        line["host"] = {"note": "host_submit_ms_per_step is the main thread's time until the last pair is accepted; it blocks on "
                                "the engine's depth limit, so it tracks GPU throughput.  The engine threads' own times, ms "
                                "per pair over the timed regions: inside pcrcg_pyramid_build (enqueue + its host round "
                                "trips) / enqueueing the restore step + pcrcg_kpfcnn_forward",
                        "front_thread_build_ms_per_pair": round(1e3 * stats["build_s"] / n, 3),
                        "model_threads_enqueue_ms_per_pair": round(1e3 * stats["launch_s"] / n, 3),
                        "pairs_per_pyramid_build": round(ppb, 2),
                        "cpu_s_per_pair": round(cpu_s_per_pair, 6),
                        "cpu_cores_busy": round(cpu_s_per_pair * args.steps * R / max(sum(r[0] for r in regions), 1e-9), 2),
                        "cpu_note": "user + system seconds of this whole process per pair over the timed regions (submitting thread, "
                                    "front thread, three model threads, HIP runtime threads); cpu_cores_busy = the same as cores kept "
                                    "busy while a region runs -- the per-rank host budget of an N-GPU run"}
        if BF16:
            line["variant"] = {"name": "bf16 feature storage", "max_abs_error_over_max_abs_vs_fp32_path": variant_error,
                               "note": "outside the 1e-4 parity bound of the fp32 path by construction; "
                                       "tests/test_bf16_gpu.py states and checks the bound"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, state_dict, limits)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio; flush it so that the JSON line is the LAST line
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
