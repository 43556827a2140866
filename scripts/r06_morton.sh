#!/bin/bash
# GPU box: what an internal spatial point order would be worth (knock-out: PCRCG_DEBUG=pyr_morton=1, csrc/morton_knock.hip).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
{
for round in 1 2; do
for cfg in "0 generator" "1 generator" "0 morton" "1 morton"; do
  set -- $cfg
  v=$(PCRCG_DEBUG=pyr_morton=$1 python3 $R/bench.py --input-order $2 --repeats 3 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'])")
  echo "levels>=1 sorted=$1 level-0 input order=$2 (480-step regions): $v"
done
done
# the gathers alone: isolated forward, KPConv gather GB/s
for cfg in "0 generator" "1 generator" "1 morton"; do
  set -- $cfg
  v=$(PCRCG_DEBUG=pyr_morton=$1 python3 $R/bench.py --input-order $2 --isolated-only --steps 10 --warmup 3 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('forward_ms', d['forward_ms'], 'kpconv avg us', d['kpconv_avg_launch_us'], 'GB/s', d['kpconv_algorithmic_GBs'], 'gemm ms/pair', d['gemm']['kernel_ms_per_pair'])")
  echo "isolated forward, levels>=1 sorted=$1 level-0 order=$2: $v"
done
} > $O/r06_knock_internal_morton_order.txt 2>&1
cat $O/r06_knock_internal_morton_order.txt
