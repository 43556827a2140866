#!/bin/bash
# GPU box: the engine's group sizes and depth once more, after the GEMM changes of the round.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp
{
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'])"; }
for round in 1 2; do
for cfg in "" "--pairs-per-forward 3 --pairs-per-build 3" "--pairs-per-forward 2 --pairs-per-build 4" "--depth 36" "--model-streams 4" "--pairs-per-build 8 --depth 40"; do
  v=$(python3 $R/bench.py $cfg --repeats 3 --no-extras --no-cpu-baseline --no-pmc 2>&1 | tail -1 | line)
  echo "[$cfg] steps=480: $v"
done
done
} > $O/r06_ab_engine_sweep_after_gemm.txt 2>&1
cat $O/r06_ab_engine_sweep_after_gemm.txt
