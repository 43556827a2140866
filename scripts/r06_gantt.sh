#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*results.db" | head -1; }
for V in "1 auto" "0 off"; do
  set -- $V
  rm -rf /tmp/p1; PCRCG_FOREST_STREAM=$1 PCRCG_ENGINE_PIPES=$2 rocprofv3 --kernel-trace -d /tmp/p1 -o p -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_g20_$2.json
  python3 $R/scripts/region_gantt.py $(db /tmp/p1) 0.3 1.0 1500 2 > $O/r06_gantt20_forest$1_$2.txt 2>&1
  cat $O/r06_gantt20_forest$1_$2.txt
  python3 -c "import json; d=json.load(open('$O/r06_g20_$2.json')); print(d['value'], d['repeats']['pairs_per_s'])"
done
