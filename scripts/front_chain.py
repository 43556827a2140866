"""Front-end chain anatomy from a rocprofv3 rocpd kernel trace of bench.py: for the stream that runs the radius
searches, per kernel name: mean duration, mean gap to the previous kernel's end, per-pair totals.
python scripts/front_chain.py db [t0_frac t1_frac]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
sel = "stream_id" if "stream_id" in cols else "queue_id"
rows = c.execute(f"select name, {sel}, start, end from kernels order by start").fetchall()
oe = [r for r in rows if "k_order_emit" in r[0]]
t_lo, t_hi = oe[0][2], oe[-1][3]            # the span in which pyramids are built
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.9
a, b = t_lo + f0 * (t_hi - t_lo), t_lo + f1 * (t_hi - t_lo)
rows = [r for r in rows if r[2] >= a and r[3] <= b]
streams = {}
for name, sid, s, e in rows:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("pcrcg::", "").split("(")[0][:44]
    streams.setdefault(sid, []).append((short, s, e))
for sid, ks in streams.items():
    busy = sum(e - s for _, s, e in ks)
    print(f"stream {sid}: {len(ks)} kernels, busy {busy / 1e6:.2f} ms = {100.0 * busy / (b - a):.1f}% of the window")
sid, front = max(streams.items(), key=lambda kv: sum(1 for k in kv[1] if ("k_radius_query" in k[0] or "k_radius_cells" in k[0])))
# kernel CHAINS (three subsampled levels each; round 6: the subsamplings may run on a side stream); a chain carries up to four pairs
pairs = max(sum(1 for ks in streams.values() for k in ks if "k_order_emit" in k[0]) / 3.0, 1)
# chain latency on the front-end stream: first kernel after the previous chain's k_reorder .. this chain's k_reorder end
spans, first = [], None
for n, s, e in front:
    if first is None:
        first = s
    if "k_reorder" in n:
        spans.append((e - first) / 1e6)
        first = None
if spans:
    sp = sorted(spans)
    print(f"chain latency (first kernel .. k_reorder end) over {len(sp)} chains: median {sp[len(sp) // 2]:.2f} ms, min {sp[0]:.2f}, max {sp[-1]:.2f}")
for osid, ks in streams.items():      # side streams of the front end
    if osid != sid and any(("k_order_emit" in k[0] or "k_kd_forest" in k[0]) for k in ks):
        agg = {}
        for n, s, e in ks:
            d = agg.setdefault(n, [0, 0.0])
            d[0] += 1
            d[1] += e - s
        print(f"side stream {osid}: per chain " + ", ".join(f"{n} {v[0] / pairs:.1f}x{v[1] / v[0] / 1e3:.0f}us" for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]))
print(f"front-end stream {sid}: window {(b - a) / 1e6:.1f} ms, {pairs:.1f} front-end chains (up to four pairs each) -> {(b - a) / 1e6 / pairs:.3f} ms per chain; the columns below are per CHAIN")
stat = {}
prev_end = None
for n, s, e in front:
    d = stat.setdefault(n, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += e - s
    if prev_end is not None:
        d[2] += max(s - prev_end, 0)
    prev_end = max(e, prev_end or e)
tot_d = sum(v[1] for v in stat.values()) / pairs / 1e3
tot_g = sum(v[2] for v in stat.values()) / pairs / 1e3
print(f"per pair: kernels {tot_d:.0f} us + gaps before kernels {tot_g:.0f} us")
print(f"{'kernel':46s} {'n/pair':>6s} {'avg us':>8s} {'gap us':>8s} {'us/pair':>8s} {'gap/pair':>8s}")
for n, (cnt, dur, gap) in sorted(stat.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{n:46s} {cnt / pairs:6.1f} {dur / cnt / 1e3:8.1f} {gap / cnt / 1e3:8.1f} {dur / pairs / 1e3:8.0f} {gap / pairs / 1e3:8.0f}")
