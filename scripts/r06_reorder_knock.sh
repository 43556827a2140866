#!/bin/bash
# GPU box: where k_reorder (the tie-order restore step) spends its time on the tie-rich T30k pairs: phases knocked out (the
# order is then wrong; tie status checks are off in bench.py's engine path only as far as the status word stays clean).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
{
for k in 0 1 2 3 4 7; do
  rm -rf /tmp/rk$k
  v=$(PCRCG_DEBUG=reorder_knock=$k rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rk$k -o p -- python3 $R/bench.py --workload T30k --steps 48 --warmup 4 --repeats 1 --no-extras --no-cpu-baseline --no-pmc --no-kernel-events 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
  t=$(grep k_reorder $(find /tmp/rk$k -name "*kernel_stats.csv" | head -1) | head -1 | awk -F, '{print "calls", $2, "avg_us", $4/1000}')
  echo "reorder_knock=$k: T30k under the profiler $v pairs/s; k_reorder $t"
done
for k in 0 7; do
  v=$(PCRCG_DEBUG=reorder_knock=$k python3 $R/bench.py --workload T30k --repeats 3 --no-extras --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'])")
  echo "reorder_knock=$k: T30k $v"
done
} > $O/r06_knock_reorder_phases.txt 2>&1
cat $O/r06_knock_reorder_phases.txt
