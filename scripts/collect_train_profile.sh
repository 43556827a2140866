#!/bin/bash
# GPU box: train-step line + rocprofv3 kernel summary (writes under gpurun_out/$1_train_*).
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/scripts/bench_train.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/${TAG}_train_bench.json
rm -rf /tmp/pt; rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 $R/scripts/bench_train.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/${TAG}_train_bench_under_rocprof.json
python3 $R/scripts/prof_summary.py $(find /tmp/pt -name "*results.db" | head -1) $O/${TAG}_train_kernel_stats.csv 13
cat $O/${TAG}_train_bench.json
head -40 $O/${TAG}_train_kernel_stats.csv
