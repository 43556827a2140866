#!/bin/bash
# GPU box: upper bound of a fused resnet-block tail (statistics from Gram matrices, unary2 and shortcut as ONE product that
# writes the block's output): the shortcut product and its half of the closing pass knocked out at the first 1 / 2 / 4 layers.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp
{
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'])"; }
for round in 1 2; do
for k in 0 1 2 4; do
  v=$(PCRCG_DEBUG=knock_tail=$k python3 $R/bench.py --repeats 3 --no-extras --no-cpu-baseline --no-pmc 2>&1 | tail -1 | line)
  echo "knock_tail=$k steps=480: $v"
done
done
} > $O/r06_knock_block_tail.txt 2>&1
cat $O/r06_knock_block_tail.txt
