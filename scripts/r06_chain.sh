#!/bin/bash
# GPU box: front chain anatomy under the tracer (48-step region) for the side-stream settings given.  usage: r06_chain.sh TAG "2 0"
TAG=${1:-r06_chain}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*results.db" | head -1; }
for FS in ${2:-2 0}; do
rm -rf /tmp/p1; PCRCG_FOREST_STREAM=$FS rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p -- python3 $R/bench.py --steps 48 --warmup 5 --repeats 1 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_under_rocprof_fs$FS.json
python3 $R/scripts/front_chain.py $(db /tmp/p1) > $O/${TAG}_front_chain_fs$FS.txt 2>&1
head -40 $O/${TAG}_front_chain_fs$FS.txt
done
