"""The kernel sequence of ONE isolated forward from a rocprofv3 rocpd kernel trace of `bench.py --isolated-only`: every
launch in stream order with its duration and the gap before it, then totals by section (encoder / coarse + GNN / decoder).
python scripts/forward_sequence.py db [which]      (which: index of the forward from the end, default 1 = the last)"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
sel = "stream_id" if "stream_id" in cols else "queue_id"
rows = c.execute(f"select name, {sel}, start, end from kernels order by start").fetchall()


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").replace("pcrcg::", "").split("(")[0][:60]


# a forward starts with the first layer's KPConv (k_kpconv_c1)
starts = [i for i, r in enumerate(rows) if "k_kpconv_c1" in r[0]]
i0 = starts[-which]
i1 = starts[-which + 1] if which > 1 else len(rows)
sid = rows[i0][1]
seq = [r for r in rows[i0:i1] if r[1] == sid]
# cut at the end of this forward: the last k_sigmoid_scores / k_l2norm_rows block
last = max(i for i, r in enumerate(seq) if "k_sigmoid_scores" in r[0] or "k_l2norm" in r[0])
seq = seq[:last + 1]
prev = None
tot_d = tot_g = 0.0
small = 0
for n, _, s, e in seq:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    d = (e - s) / 1e3
    tot_d += d
    tot_g += max(gap, 0.0)
    small += d < 12.0
    print(f"{d:8.1f} us  gap {gap:7.1f}  {short(n)}")
    prev = e
print(f"# {len(seq)} launches, kernels {tot_d:.0f} us, gaps {tot_g:.0f} us, wall {(seq[-1][3] - seq[0][2]) / 1e3:.0f} us; {small} launches under 12 us")
