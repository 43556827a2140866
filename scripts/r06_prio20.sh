#!/bin/bash
# GPU box: the front-end stream at high HIP priority, 20-step regions (the driver's setting), interleaved.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp
{
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'])"; }
for round in 1 2 3; do
for cfg in "" "--front-priority -1"; do
  v=$(python3 $R/bench.py $cfg --steps 20 --warmup 5 --repeats 7 --no-extras --no-cpu-baseline --no-pmc 2>&1 | tail -1 | line)
  echo "[$cfg] steps=20: $v"
done
done
} > $O/r06_ab_front_priority_20_steps.txt 2>&1
cat $O/r06_ab_front_priority_20_steps.txt
