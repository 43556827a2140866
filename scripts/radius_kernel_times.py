"""Per-table kernel durations of `scripts/radius_bench.py RECIPE --mode old|new --reps R` from the rocprofv3 rocpd
database of that run: the main search kernel's dispatches come in groups of 3 + R (warm-up + timed) per table, in
table order.   python scripts/radius_kernel_times.py db R [kernel-substring]"""
import sqlite3
import sys

import numpy as np

db, reps = sys.argv[1], int(sys.argv[2])
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
names = ["conv0", "pool0", "up0", "conv1", "pool1", "up1", "conv2", "pool2", "up2", "conv3"]
want = sys.argv[3] if len(sys.argv) > 3 else None
for key in ([want] if want else ["k_radius_cells", "k_radius_query<256", "k_radius_query<1024"]):
    d = [(e - s) / 1e3 for n, s, e in rows if key in n.replace("(anonymous namespace)::", "")]
    if not d:
        continue
    g = 3 + reps
    # the pyramid build at the start of the script also launches these kernels: keep the LAST len(names) groups
    d = d[len(d) - g * (len(d) // g if len(d) // g < len(names) else len(names)):]
    print(f"{key}: {len(d)} dispatches")
    tot = 0.0
    for i in range(len(d) // g):
        grp = d[i * g + 3:(i + 1) * g]
        tot += float(np.median(grp))
        print(f"  {names[i % len(names)]:6s} median {np.median(grp):8.1f} us   min {min(grp):8.1f}  max {max(grp):8.1f}")
    print(f"  sum of medians {tot:.1f} us")
