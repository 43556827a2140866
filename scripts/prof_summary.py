"""Turn a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats ... -> *_results.db) into the
per-kernel summary committed under profiles/ (name, calls, total/avg duration, share)."""
import sqlite3
import sys


def gather_split(c):
    """The KPConv gather launches of a bench.py run come in two kinds -- inside the timed regions, beside the other streams'
    kernels, and bench.py's isolated leg, alone on the GPU -- and the table's average mixes them: say both."""
    try:
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        sel = "stream_id" if "stream_id" in cols else "queue_id"
        rows = c.execute(f"select name, {sel}, start, end from kernels order by start").fetchall()
    except sqlite3.Error:
        return ""
    rows = [r for r in rows if "k_probe" not in r[0]]
    gathers = [r for r in rows if "k_kpconv_mfma" in r[0] or "k_kpconv_c1" in r[0]]
    if not gathers:
        return ""
    import bisect
    starts = [r[2] for r in rows]
    longest = max(r[3] - r[2] for r in rows)
    beside, alone = [], []
    for name, sid, s, e in gathers:
        lo = bisect.bisect_left(starts, s - longest)
        hi = bisect.bisect_right(starts, e)
        shared = any(r[1] != sid and r[2] < e and r[3] > s for r in rows[lo:hi])
        (beside if shared else alone).append((e - s) / 1e3)
    out = "# KPConv gather launches (k_kpconv_mfma, k_kpconv_c1): "
    if beside:
        out += f"{len(beside)} beside other streams' kernels, average {sum(beside) / len(beside):.2f} us"
    if alone:
        out += f"{'; ' if beside else ''}{len(alone)} with the GPU to themselves (bench.py's isolated leg), average {sum(alone) / len(alone):.2f} us"
    return out + " -- the table's average mixes the two\n"


def main(db, out, steps=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    total = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary ({db.split('/')[-1]})\n")
        f.write(f"# total kernel time {total / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches")
        if steps:
            f.write(f"; {steps} bench steps -> {total / 1e3 / steps:.3f} ms of kernels per step")
        f.write("\n# durations in microseconds\n")
        f.write(gather_split(c))
        f.write("calls,total_us,avg_us,percent,name\n")
        for name, calls, tot, avg, pct in rows:
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if len(short) > 110:
                short = short[:107] + "..."
            f.write(f"{calls},{tot:.1f},{avg:.2f},{pct:.2f},{short}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else None)
