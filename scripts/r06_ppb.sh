#!/bin/bash
TAG=${1:-r06_ppb}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
{
for cfg in "4 24" "8 32" "8 48" "6 36"; do
  set -- $cfg
  for ST in 20 480; do
    RP=5; [ $ST = 480 ] && RP=3
    v=$(python3 $R/bench.py --steps $ST --warmup 5 --repeats $RP --pairs-per-build $1 --depth $2 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'], d['config']['pairs_per_pyramid_build'])")
    echo "pairs_per_build=$1 depth=$2 steps=$ST : $v"
  done
done
} > $O/${TAG}.txt 2>&1
cat $O/${TAG}.txt
