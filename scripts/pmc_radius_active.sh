#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
P="SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_IFETCH"
rm -rf /tmp/q9
RADIUS_BENCH_ONLY=conv0 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/q9 -o p -- python3 $R/scripts/radius_bench.py S30k --mode new --reps 5 > /dev/null 2>&1
python3 - <<PY
import csv, collections, glob
f = glob.glob("/tmp/q9/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if "k_radius_cells" in r["Kernel_Name"]:
        d[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
ids = sorted(d, key=int)[-8:]
acc = collections.defaultdict(float)
for i in ids:
    for k, v in d[i].items(): acc[k] += v / len(ids)
for k, v in sorted(acc.items()): print(f"{k:24s} {v/1e6:10.3f} M per conv0 launch")
PY
