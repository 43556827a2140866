"""Turn a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats ... -> *_results.db) into the
per-kernel summary committed under profiles/ (name, calls, total/avg duration, share)."""
import sqlite3
import sys


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
        f.write("calls,total_us,avg_us,percent,name\n")
        for name, calls, tot, avg, pct in rows:
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if len(short) > 110:
                short = short[:107] + "..."
            f.write(f"{calls},{tot:.1f},{avg:.2f},{pct:.2f},{short}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else None)
