#!/bin/bash
# GPU box: which side of the engine bounds the streaming rate -- the front-end stream given priority, fewer / more model streams,
# two front-end streams.  480-step regions, interleaved.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp
{
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'])"; }
for round in 1 2; do
for cfg in "" "--front-priority -1" "--model-streams 2" "--model-streams 2 --front-priority -1" "--front-threads 2 --front-streams 2" "--front-threads 2 --front-streams 2 --model-streams 2"; do
  v=$(python3 $R/bench.py $cfg --repeats 3 --no-extras --no-cpu-baseline 2>&1 | tail -1 | line)
  echo "[$cfg] steps=480: $v"
done
done
} > $O/r06_ab_engine_balance.txt 2>&1
cat $O/r06_ab_engine_balance.txt
