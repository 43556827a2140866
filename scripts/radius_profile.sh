#!/bin/bash
# GPU box: kernel durations of the radius search, table by table, old (per-query) vs new (cell-cooperative) kernel.
# usage: scripts/radius_profile.sh TAG [RECIPE ...]   -> gpurun_out/TAG_radius_<recipe>.txt
TAG=${1:-r04}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*results.db" | head -1; }
for W in ${@:-S30k}; do
  : > $O/${TAG}_radius_$W.txt
  for M in old new; do
    rm -rf /tmp/pr; rocprofv3 --kernel-trace --stats -d /tmp/pr -o p -- python3 $R/scripts/radius_bench.py $W --mode $M --reps 20 > /tmp/pr.log 2>&1
    echo "## $W, --mode $M (rocprofv3 --kernel-trace: the kernels' own durations)" >> $O/${TAG}_radius_$W.txt
    python3 $R/scripts/radius_kernel_times.py $(db /tmp/pr) 20 >> $O/${TAG}_radius_$W.txt 2>&1
  done
  cat $O/${TAG}_radius_$W.txt
done
