"""Per-kernel-family counters of the pair engine under `rocprofv3 --pmc` (scripts/r06_counters.sh): L2 hit rate, waves, busy
cycles, for the run alone (--isolated-only), with three and with four model streams; and whether kernels still ran beside
each other under counter collection (mean number of kernels in flight from the kernel trace of the same run).
python scripts/pmc_engine_summary.py label=dir ..."""
import collections
import csv
import glob
import sys


def family(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("pcrcg::", "").split("(")[0].split("<")[0]
    return n


def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = family(r["Kernel_Name"])
            out[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
    return out, {k: len(v) for k, v in calls.items()}


def concurrency(d):
    ev = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "probe" in r["Kernel_Name"]:
                continue
            ev.append((int(r["Start_Timestamp"]), 1))
            ev.append((int(r["End_Timestamp"]), -1))
    ev.sort()
    if not ev:
        return None
    # the middle half of the trace
    t0, t1 = ev[0][0], ev[-1][0]
    a, b = t0 + (t1 - t0) // 4, t1 - (t1 - t0) // 4
    run, last, area = 0, ev[0][0], 0.0
    for t, s in ev:
        lo, hi = max(last, a), min(t, b)
        if hi > lo:
            area += run * (hi - lo)
        run += s
        last = t
    return area / (b - a)


def main(args):
    runs = [a.split("=", 1) for a in args]
    data = {}
    for label, d in runs:
        c, n = counters(d)
        data[label] = (c, n, concurrency(d))
    for label, _ in runs:
        cc = data[label][2]
        print(f"{label}: kernels in flight on average (middle half of the trace, under counter collection): {cc:.2f}" if cc else f"{label}: no kernel trace")
    names = sorted({k for c, _, _ in data.values() for k in c}, key=lambda k: -max(data[l][0].get(k, {}).get("SQ_BUSY_CYCLES", data[l][0].get(k, {}).get("TCC_HIT_sum", 0)) for l, _ in runs))
    cols = sorted({cn for c, _, _ in data.values() for k in c for cn in c[k]})
    print("counters collected:", " ".join(cols))
    hdr = "%-28s" % "kernel family"
    for label, _ in runs:
        hdr += " | %-34s" % (label + ": calls L2hit% waves/launch busyMcyc")
    print(hdr)
    for k in names[:22]:
        line = "%-28s" % k[:28]
        for label, _ in runs:
            c, n, _cc = data[label]
            v = c.get(k)
            if not v:
                line += " | %-34s" % "-"
                continue
            hit, miss = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
            calls = max(n.get(k, 1), 1)
            hr = "%5.1f" % (100 * hit / (hit + miss)) if hit + miss > 0 else "  n/a"
            line += " | %5d %s %9.0f %8.3f      " % (calls, hr, v.get("SQ_WAVES", 0) / calls, v.get("SQ_BUSY_CYCLES", 0) / calls / 1e6)
        print(line)


if __name__ == "__main__":
    main(sys.argv[1:])
