#!/bin/bash
# GPU box: the engine's stream placement by dispatcher class -- A/B at long and short regions.
TAG=${1:-r06_pipes}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-extras --no-cpu-baseline --repeats 3"
run() { # name, env..., -- args
  name=$1; shift
  env "$@" > /dev/null 2>&1
}
{
echo "# 480-step regions"
for cfg in "auto 1 3" "auto 0 3" "off 0 3" "off 1 3" "auto 1 6" "auto 1 4" "front4 1 4" "auto 1 9"; do
  set -- $cfg
  v=$(PCRCG_ENGINE_PIPES=$1 PCRCG_FOREST_STREAM=$2 $B --model-streams $3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'], d['config'].get('engine_streams_by_dispatcher'))")
  echo "pipes=$1 forest_stream=$2 model_streams=$3 : $v"
done
echo "# 20-step regions (--steps 20 --warmup 5)"
for cfg in "auto 1 3" "auto 0 3" "off 0 3" "auto 1 6" "auto 1 4" "front4 1 4"; do
  set -- $cfg
  v=$(PCRCG_ENGINE_PIPES=$1 PCRCG_FOREST_STREAM=$2 $B --steps 20 --warmup 5 --repeats 5 --model-streams $3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'])")
  echo "pipes=$1 forest_stream=$2 model_streams=$3 : $v"
done
} > $O/${TAG}.txt 2>&1
cat $O/${TAG}.txt
