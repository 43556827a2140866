#!/bin/bash
# GPU box: in-engine A/B of library switches (PCRCG_DEBUG="name=value,..", include/pcrcg.h) or bench flags: one line per
# setting with the median and the regions.
# usage: scripts/ab_env.sh [-s STEPS] "" "PCRCG_DEBUG=x6_tile=1" "ARGS=--model-streams=4" ...   (an empty string = defaults)
STEPS=100
if [ "$1" = "-s" ]; then STEPS=$2; shift 2; fi
for cfg in "$@"; do
  extra=""; envs=""
  for tok in $cfg; do case $tok in ARGS=*) extra="$extra ${tok#ARGS=}";; *) envs="$envs $tok";; esac; done
  v=$(env $envs python bench.py $extra --no-cpu-baseline --no-extras --steps $STEPS --repeats 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'])")
  echo "[$cfg] $v"
done
