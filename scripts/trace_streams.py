"""Per-stream view of a rocprofv3 rocpd kernel trace of bench.py: busy time, kernel count, and for the front-end
stream the span of each pair's chain.  python scripts/trace_streams.py db [t0_frac t1_frac]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, stream_id, queue_id, start, end from kernels order by start").fetchall() \
    if False else None
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
sel = "stream_id" if "stream_id" in cols else "queue_id"
rows = c.execute(f"select name, {sel}, start, end from kernels order by start").fetchall()
t_lo, t_hi = rows[0][2], max(r[3] for r in rows)
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.9
a, b = t_lo + f0 * (t_hi - t_lo), t_lo + f1 * (t_hi - t_lo)
rows = [r for r in rows if r[2] >= a and r[3] <= b]
wall = (b - a) / 1e6
print(f"window {wall:.1f} ms, {len(rows)} kernels")
streams = {}
for name, sid, s, e in rows:
    streams.setdefault(sid, []).append((name, s, e))
for sid, ks in sorted(streams.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for _, s, e in ks) / 1e6
    # union of intervals (kernels of one stream do not overlap, but be safe)
    names = {}
    for n, s, e in ks:
        short = n.split("(")[0].replace("void ", "")[-40:]
        names[short] = names.get(short, 0) + (e - s)
    top = sorted(names.items(), key=lambda kv: -kv[1])[:3]
    print(f"stream {sid}: {len(ks):6d} kernels, busy {busy:8.2f} ms ({100 * busy / wall:5.1f}% of window)  top: " +
          ", ".join(f"{n} {t / 1e6:.1f}ms" for n, t in top))
# front-end chains: from k_init/k_minmax of level 0 ... use k_grid_init occurrences with the largest grid as pair starts
front = max(streams.items(), key=lambda kv: sum(1 for k in kv[1] if ("k_radius_query" in k[0] or "k_radius_cells" in k[0])))[1]
starts = [s for n, s, e in front if "k_kd_init" in n]
if starts:
    print(f"tie phases in window: {len(starts)}; mean period {(starts[-1] - starts[0]) / max(len(starts) - 1, 1) / 1e6:.3f} ms")
    # duration of each tie phase: k_kd_init start -> k_reorder end
    ends = [e for n, s, e in front if "k_reorder" in n]
    d = [(e - s) / 1e6 for s, e in zip(starts, ends) if e > s]
    print(f"tie phase span: mean {sum(d) / max(len(d), 1):.3f} ms, max {max(d):.3f} ms")
    for tag in ("k_kd_big", "k_kd_sub", "k_reorder", "k_radius_query", "k_order_emit"):
        v = [(e - s) / 1e3 for n, s, e in front if tag in n]
        if v:
            print(f"  {tag:16s} n={len(v):5d} mean {sum(v) / len(v):8.1f} us  total/pair {sum(v) / len(starts) / 1e3:.3f} ms")
