#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
{
for Q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$Q timeout 120 scripts/micro/queue_pipes 8
done
GPU_MAX_HW_QUEUES=16 timeout 200 scripts/micro/queue_pipes 12
} > $O/r06_queue_pipes.txt 2>&1
cat $O/r06_queue_pipes.txt
# the 20-step region's timeline with the new front end
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*results.db" | head -1; }
rm -rf /tmp/p1; PCRCG_FOREST_STREAM=0 rocprofv3 --kernel-trace -d /tmp/p1 -o p -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 2 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_t20_bench.json
python3 $R/scripts/region_timeline.py $(db /tmp/p1) 2.0 > $O/r06_t20_timeline.txt 2>&1
cat $O/r06_t20_timeline.txt | tail -30
