#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "(TCP|TCC|TA|TD|GRBM|SQ)_[A-Z0-9_]*(LATENCY|UTCL|STALL|MISS|BUSY)[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $O/mem_counter_names.txt
echo >> $O/mem_counter_names.txt
run() {
  rm -rf /tmp/q9
  RADIUS_BENCH_ONLY=conv0 rocprofv3 --kernel-trace --pmc $1 --output-format csv -d /tmp/q9 -o p -- python3 $R/scripts/radius_bench.py S30k --mode $2 --reps 5 > /dev/null 2>&1
  python3 - "$2" <<PY
import csv, collections, glob, sys
f = glob.glob("/tmp/q9/**/*counter_collection.csv", recursive=True)
if not f: print("no csv"); sys.exit()
d = collections.defaultdict(lambda: collections.defaultdict(float))
key = "k_radius_cells" if sys.argv[1] == "new" else "k_radius_query<256"
for r in csv.DictReader(open(f[0])):
    if key in r["Kernel_Name"].replace("(anonymous namespace)::", ""):
        d[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
ids = sorted(d, key=int)[-8:]
acc = collections.defaultdict(float)
for i in ids:
    for k, v in d[i].items(): acc[k] += v / len(ids)
for k, v in sorted(acc.items()): print(f"{sys.argv[1]:4s} {k:40s} {v:14.0f} per conv0 launch")
PY
}
for M in new old; do
run "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" $M
run "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" $M
run "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" $M
done
