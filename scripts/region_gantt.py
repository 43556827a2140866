"""Per-stream busy share of bench.py's LAST timed regions from a rocprofv3 rocpd kernel trace: regions = clusters of kernels
separated by >= GAP ms without any kernel; per region and stream the busy share per BIN ms.
python scripts/region_gantt.py results.db [gap_ms=0.3] [bin_ms=1.0] [min_kernels=1500] [regions=2]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
ktab = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel_dispatch" in t][0]
cols = [r[1] for r in c.execute(f"pragma table_info({ktab})")]
sid = "stream_id" if "stream_id" in cols else "queue_id"
rows = c.execute(f"select name, {sid}, start, end from {ktab} order by start").fetchall()
gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.3e6
binw = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 1.0e6
minK = int(sys.argv[4]) if len(sys.argv) > 4 else 1500
nreg = int(sys.argv[5]) if len(sys.argv) > 5 else 2
regions, cur, last_end = [], [], None
for r in rows:
    if last_end is not None and r[2] - last_end > gap:
        regions.append(cur)
        cur = []
    cur.append(r)
    last_end = r[3] if last_end is None else max(last_end, r[3])
regions.append(cur)
big = [g for g in regions if len(g) >= minK]
print(f"{len(regions)} clusters, {len(big)} with >= {minK} kernels; showing the last {nreg}")
for g in big[-nreg:]:
    t0, t1 = g[0][2], max(r[3] for r in g)
    nb = int((t1 - t0) / binw) + 1
    print(f"\nregion: {len(g)} kernels, {(t1 - t0) / 1e6:.2f} ms; busy share per {binw / 1e6:g} ms bin (0-9, '.' = idle)")
    streams = {}
    for n, s, a, b in g:
        streams.setdefault(s, []).append((n, a, b))
    for s, ks in sorted(streams.items(), key=lambda kv: kv[1][0][1]):
        bins = [0.0] * nb
        for n, a, b in ks:
            i0, i1 = int((a - t0) / binw), int((b - t0) / binw)
            for i in range(i0, i1 + 1):
                lo, hi = t0 + i * binw, t0 + (i + 1) * binw
                bins[i] += max(0.0, min(b, hi) - max(a, lo)) / binw
        kind = "front" if any("k_radius_cells" in k[0] or "k_radius_query" in k[0] for k in ks) else (
            "forest" if any("k_kd_forest" in k[0] for k in ks) else ("model" if any("gemm" in k[0] for k in ks) else "other"))
        line = "".join("." if x < 0.05 else str(min(9, int(x * 10))) for x in bins)
        busy = sum(b - a for _, a, b in ks) / 1e6
        print(f"  stream {s:3d} {kind:6s} {len(ks):5d} k, busy {busy:6.2f} ms, {(ks[0][1] - t0) / 1e6:6.2f} .. {(ks[-1][2] - t0) / 1e6:6.2f} ms  {line}")
    # forward starts on model streams: the first kernel after an idle gap > 30 us whose name is a fill (zero arena)
    for s, ks in sorted(streams.items(), key=lambda kv: kv[1][0][1]):
        if not any("gemm" in k[0] for k in ks):
            continue
        marks = [f"{(a - t0) / 1e6:.1f}" for n, a, b in ks if "fillBuffer" in n]
        print(f"  stream {s:3d} zero-arena fills (one per forward call) at ms: " + " ".join(marks))
