"""Fill / drain view of bench.py's timed regions from a rocprofv3 rocpd kernel trace: regions = clusters of kernels
separated by >= GAP ms without any kernel; per region the kernels in flight per 0.5 ms bin and each stream's first /
last kernel.  python scripts/region_timeline.py results.db [gap_ms]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
ktab = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel_dispatch" in t][0]
cols = [r[1] for r in c.execute(f"pragma table_info({ktab})")]
sid = "stream_id" if "stream_id" in cols else "queue_id"
rows = c.execute(f"select name, {sid}, start, end from {ktab} order by start").fetchall()
gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.4e6
regions, cur, last_end = [], [], None
for r in rows:
    if last_end is not None and r[2] - last_end > gap:
        regions.append(cur)
        cur = []
    cur.append(r)
    last_end = r[3] if last_end is None else max(last_end, r[3])
regions.append(cur)
big = [g for g in regions if len(g) > 1000]
print(f"{len(regions)} clusters, {len(big)} with > 1000 kernels")
for g in big[-3:]:
    t0, t1 = g[0][2], max(r[3] for r in g)
    print(f"\nregion: {len(g)} kernels, {(t1 - t0) / 1e6:.2f} ms")
    streams = {}
    for n, s, a, b in g:
        streams.setdefault(s, []).append((n, a, b))
    for s, ks in sorted(streams.items(), key=lambda kv: kv[1][0][1]):
        busy = sum(b - a for _, a, b in ks) / 1e6
        rq = sum(1 for k in ks if "k_radius_query" in k[0] or "k_radius_cells" in k[0])
        print(f"  stream {s}: {len(ks):5d} kernels, first at {(ks[0][1] - t0) / 1e6:6.2f} ms, last ends {(ks[-1][2] - t0) / 1e6:6.2f} ms, "
              f"busy {busy:6.2f} ms{'  (front end)' if rq else ''}")
    nb = int((t1 - t0) / 0.5e6) + 1
    bins = [0.0] * nb
    for n, s, a, b in g:
        i0, i1 = int((a - t0) / 0.5e6), int((b - t0) / 0.5e6)
        for i in range(i0, i1 + 1):
            lo, hi = t0 + i * 0.5e6, t0 + (i + 1) * 0.5e6
            bins[i] += max(0.0, min(b, hi) - max(a, lo)) / 0.5e6
    print("  kernels in flight per 0.5 ms: " + " ".join(f"{x:.1f}" for x in bins))
