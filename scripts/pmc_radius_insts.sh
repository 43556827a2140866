#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/sq_counter_names.txt
P="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SENDMSG"
rm -rf /tmp/q9
RADIUS_BENCH_ONLY=conv0 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/q9 -o p -- python3 $R/scripts/radius_bench.py S30k --mode new --reps 5 > /dev/null 2>&1
python3 - <<PY
import csv, collections, glob
f = glob.glob("/tmp/q9/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_radius_cells" in r["Kernel_Name"]:
        d[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
# the last 8 dispatches are conv0
ids = sorted(d, key=int)[-8:]
acc = collections.defaultdict(float)
for i in ids:
    for k, v in d[i].items(): acc[k] += v / len(ids)
for k, v in sorted(acc.items()): print(f"{k:24s} {v/1e6:8.2f} M per conv0 launch = {v/60000:7.1f} per query")
PY
