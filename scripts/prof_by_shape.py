"""Group a rocprofv3 kernel trace (rocpd *_results.db) by (kernel, grid size): calls, total and mean
duration -- shows which problem shapes a kernel family spends its time on."""
import collections
import sqlite3
import sys


def main(db, out, steps):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    gx = "grid_x" if "grid_x" in cols else "grid_size_x"
    wx = "workgroup_x" if "workgroup_x" in cols else "workgroup_size_x"
    gy = gx.replace("_x", "_y")
    gz = gx.replace("_x", "_z")
    rows = c.execute(f"select name, {gx}, {gy}, {gz}, {wx}, start, end from kernels").fetchall()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, x, y, z, w, s, e in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        k = (short, x // max(w, 1), y, z)
        agg[k][0] += 1
        agg[k][1] += (e - s) / 1e3
    tot = sum(v[1] for v in agg.values())
    with open(out, "w") as f:
        f.write(f"# total {tot / 1e3:.3f} ms over {steps} steps; per-step us by (kernel, blocks x,y,z)\n")
        f.write("us_per_step,calls_per_step,avg_us,blocks,kernel\n")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:120]:
            f.write(f"{v[1] / steps:.1f},{v[0] / steps:.2f},{v[1] / v[0]:.2f},{k[1]}x{k[2]}x{k[3]},{k[0]}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
