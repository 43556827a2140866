#!/bin/bash
# GPU box: same-day baseline of a tree: gpu tests, driver-setting bench, front chain under the tracer.  usage: r06_baseline.sh TAG
TAG=${1:-r06_base}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R && timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/${TAG}_pytest.txt
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
python3 $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench20_$i.json
done
python3 $R/bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench480.json
db() { find "$1" -name "*results.db" | head -1; }
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p -- python3 $R/bench.py --steps 48 --warmup 5 --repeats 1 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_under_rocprof.json
python3 $R/scripts/prof_summary.py $(db /tmp/p1) $O/${TAG}_kernel_stats.csv 82
python3 $R/scripts/front_chain.py $(db /tmp/p1) > $O/${TAG}_front_chain.txt 2>&1
python3 - <<'P' $O $TAG
import json,sys,glob
O,T=sys.argv[1:3]
for f in sorted(glob.glob(f"{O}/{T}_bench*.json")):
    try:
        d=json.loads(open(f).read()); print(f.split('/')[-1], d["value"], d.get("ms_per_step"))
    except Exception as e: print(f, "ERR", e)
P
