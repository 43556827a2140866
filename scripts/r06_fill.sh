#!/bin/bash
# GPU box: the engine's fill mode (chains of an EMPTY engine spread over the idle model dispatchers) and the default side-stream setting.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp
{
for round in 1 2 3; do
for cfg in "1 0" "0 0" "1 2" "0 2"; do
  set -- $cfg
  v=$(PCRCG_FILL_STREAMS=$1 PCRCG_FOREST_STREAM=$2 python3 $R/bench.py --steps 20 --warmup 5 --repeats 5 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'])")
  echo "fill_streams=$1 side_streams=$2 steps=20: $v"
done
done
for cfg in "1 0" "0 0" "1 2"; do
  set -- $cfg
  v=$(PCRCG_FILL_STREAMS=$1 PCRCG_FOREST_STREAM=$2 python3 $R/bench.py --repeats 3 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'])")
  echo "fill_streams=$1 side_streams=$2 steps=480: $v"
done
} > $O/r06_ab_fill_streams.txt 2>&1
cat $O/r06_ab_fill_streams.txt
