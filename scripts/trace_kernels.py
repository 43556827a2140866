"""Print per-kernel mean durations from a rocprofv3 rocpd database: python trace_kernels.py db [filter]"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = c.execute("select name, grid_x, grid_y, grid_z, workgroup_x, start, end from kernels order by start").fetchall()
agg = collections.OrderedDict()
for name, x, y, z, w, s, e in rows:
    if flt and flt not in name:
        continue
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    k = (short, x // max(w, 1), y, z)
    agg.setdefault(k, []).append((e - s) / 1e3)
for k, v in agg.items():
    v = sorted(v)
    print(f"{v[len(v) // 2]:9.1f} us (n={len(v):3d})  {k[1]}x{k[2]}x{k[3]}  {k[0]}")
