#!/bin/bash
# GPU box: the pair engine under rocprofv3 --pmc (round 5's review, item 4): three and four model streams (streams in creation
# order, as rounds 3-5 took them: the fourth shares the front-end stream's dispatcher) and the forward alone; one pass for the
# SQ / GRBM counters, one for the L2's.  The program directly after `--`; --pmc with --kernel-trace only.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
P1="SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="TCC_HIT_sum TCC_MISS_sum"
{
for ms in 3 4; do
  rm -rf /tmp/c$ms; mkdir -p /tmp/c$ms
  i=0
  for P in "$P1" "$P2"; do
    i=$((i+1))
    PCRCG_ENGINE_PIPES=off timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/c$ms/p$i -o p -- python3 $R/bench.py --model-streams $ms --steps 48 --warmup 4 --repeats 1 --no-extras --no-cpu-baseline --no-pmc --no-kernel-events 2> /tmp/c$ms/err$i | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('model streams $ms, counters [$P]: bench under the profiler', d['value'], 'pairs/s')"
  done
done
rm -rf /tmp/c0; mkdir -p /tmp/c0
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/c0/p$i -o p -- python3 $R/bench.py --isolated-only --steps 3 --warmup 1 > /dev/null 2> /tmp/c0/err$i
done
python3 $R/scripts/pmc_engine_summary.py alone=/tmp/c0 three_model_streams=/tmp/c3 four_model_streams=/tmp/c4
} > $O/r06_concurrency_counters.txt 2>&1
tail -5 /tmp/c3/err1 >> $O/r06_concurrency_counters.txt
cat $O/r06_concurrency_counters.txt
