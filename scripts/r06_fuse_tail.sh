#!/bin/bash
# GPU box: the fused resnet-block tail (needs profiles/r06_fused_block_tail.patch.txt applied: PCRCG_DEBUG=fuse_tail=L,
# fuse_tail_kind=K), interleaved, 480-step and 20-step regions; forward alone; by layer and kind of block.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp
{
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'])"; }
for round in 1 2 3; do
for k in 0 1; do
  v=$(PCRCG_DEBUG=fuse_tail=$k python3 $R/bench.py --repeats 3 --no-extras --no-cpu-baseline --no-pmc 2>&1 | tail -1 | line)
  echo "fuse_tail=$k steps=480: $v"
done
done
for round in 1 2; do
for k in 0 1; do
  v=$(PCRCG_DEBUG=fuse_tail=$k python3 $R/bench.py --steps 20 --warmup 5 --repeats 5 --no-extras --no-cpu-baseline --no-pmc 2>&1 | tail -1 | line)
  echo "fuse_tail=$k steps=20: $v"
done
done
for k in 0 1; do
  PCRCG_DEBUG=fuse_tail=$k python3 $R/bench.py --isolated-only --steps 20 --warmup 3 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fuse_tail=$k forward alone', d['forward_ms'], 'ms; GEMM kernels per pair', d['gemm']['kernel_ms_per_pair'], 'ms over', d['gemm']['launches_per_pair'], 'launches')"
done
for cfg in "fuse_tail=0" "fuse_tail=1" "fuse_tail=1,fuse_tail_kind=1" "fuse_tail=1,fuse_tail_kind=2" "fuse_tail=2,fuse_tail_kind=2" "fuse_tail=4,fuse_tail_kind=2" "fuse_tail=4,fuse_tail_kind=1" "fuse_tail=0"; do
  v=$(PCRCG_DEBUG=$cfg python3 $R/bench.py --repeats 3 --no-extras --no-cpu-baseline --no-pmc 2>&1 | tail -1 | line)
  echo "$cfg steps=480: $v"
done
} > $O/r06_ab_fused_block_tail.txt 2>&1
cat $O/r06_ab_fused_block_tail.txt
