#!/bin/bash
# GPU box: the split-K targets of products planned beside other streams (x6_t1 / x6_t2), after the k-loop fix: 480-step and
# 20-step regions, K120k and T30k.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp
{
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'])"; }
for round in 1 2; do
for cfg in "x6_t1=32,x6_t2=128" "x6_t1=16,x6_t2=64" "x6_t1=64,x6_t2=256" "x6_t1=1,x6_t2=1" "x6_t1=32,x6_t2=512"; do
  v=$(PCRCG_DEBUG=$cfg python3 $R/bench.py --repeats 3 --no-extras --no-cpu-baseline --no-pmc 2>&1 | tail -1 | line)
  echo "$cfg steps=480: $v"
done
done
for round in 1 2 3; do
for cfg in "x6_t1=32,x6_t2=128" "x6_t1=16,x6_t2=64" "x6_t1=1,x6_t2=1"; do
  v=$(PCRCG_DEBUG=$cfg python3 $R/bench.py --steps 20 --warmup 5 --repeats 7 --no-extras --no-cpu-baseline --no-pmc 2>&1 | tail -1 | line)
  echo "$cfg steps=20: $v"
done
done
for cfg in "x6_t1=32,x6_t2=128" "x6_t1=1,x6_t2=1"; do
  v=$(PCRCG_DEBUG=$cfg python3 $R/bench.py --workload K120k --steps 120 --repeats 3 --no-extras --no-cpu-baseline --no-pmc 2>&1 | tail -1 | line)
  echo "$cfg K120k: $v"
  v=$(PCRCG_DEBUG=$cfg python3 $R/bench.py --workload T30k --repeats 3 --no-extras --no-cpu-baseline --no-pmc 2>&1 | tail -1 | line)
  echo "$cfg T30k: $v"
done
} > $O/r06_ab_splitk_targets.txt 2>&1
cat $O/r06_ab_splitk_targets.txt
