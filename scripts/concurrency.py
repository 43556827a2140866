"""How many kernels run at once in the engine: from a rocprofv3 rocpd kernel trace of bench.py, the share of the
steady-state window during which 0, 1, 2, ... kernels are executing (over all streams), and the same weighted by the
workgroups each running kernel has (a rough occupancy proxy: grid size / 1024, capped at 1).
python scripts/concurrency.py db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
gcols = [k for k in ("grid_size_x", "grid_x", "grid_size") if k in cols]
wcols = [k for k in ("workgroup_size_x", "workgroup_x", "workgroup_size") if k in cols]
sel = "name, start, end" + (f", {gcols[0]}" if gcols else ", 0") + (f", {wcols[0]}" if wcols else ", 1")
rows = c.execute(f"select {sel} from kernels order by start").fetchall()
oe = [r for r in rows if "k_order_emit" in r[0]]
t_lo, t_hi = oe[0][1], oe[-1][2]
a, b = t_lo + 0.3 * (t_hi - t_lo), t_lo + 0.9 * (t_hi - t_lo)
ev = []
for name, s, e, g, w in rows:
    if e <= a or s >= b:
        continue
    s, e = max(s, a), min(e, b)
    wgs = (g / max(w, 1)) if g and w else 1.0
    fill = min(wgs / 1024.0, 1.0)
    ev.append((s, 1, fill))
    ev.append((e, -1, -fill))
ev.sort()
hist, filled = {}, 0.0
n, f, prev = 0, 0.0, a
for t, dn, df in ev:
    hist[n] = hist.get(n, 0) + (t - prev)
    filled += min(f, 1.0) * (t - prev)
    prev = t
    n += dn
    f += df
hist[n] = hist.get(n, 0) + (b - prev)
tot = b - a
print("kernels running at once -> share of the window")
for k in sorted(hist):
    print(f"  {k}: {100.0 * hist[k] / tot:5.1f} %")
print(f"mean concurrency {sum(k * v for k, v in hist.items()) / tot:.2f}; "
      f"time-average of min(1, sum of grid/1024 over running kernels) = {filled / tot:.2f}")
