#!/bin/bash
# GPU box: overlapped builds -- two front threads on ONE front-end stream (the library keeps the chains whole; one thread's round
# trip and host work overlap the other's chain) against the one-thread engine, 480-step and 20-step regions, interleaved.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp
{
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'])"; }
for round in 1 2; do
for ft in 1 2 3; do
  v=$(python3 $R/bench.py --front-threads $ft --repeats 3 --no-extras --no-cpu-baseline 2>&1 | tail -1 | line)
  echo "front_threads=$ft steps=480: $v"
done
done
for round in 1 2 3; do
for ft in 1 2; do
  v=$(python3 $R/bench.py --front-threads $ft --steps 20 --warmup 5 --repeats 5 --no-extras --no-cpu-baseline 2>&1 | tail -1 | line)
  echo "front_threads=$ft steps=20: $v"
done
done
python3 $R/scripts/engine_stats.py 1 2>&1 | tail -3; python3 $R/scripts/engine_stats.py 2 2>&1 | tail -3
} > $O/r06_ab_overlapped_builds.txt 2>&1
cat $O/r06_ab_overlapped_builds.txt
