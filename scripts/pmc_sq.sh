#!/bin/bash
# GPU box: SQ-counter passes over the isolated forward (bench.py --isolated-only), per-kernel summary.
# usage: scripts/pmc_sq.sh <tag>
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/${TAG}_sq_counter_names.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rm -rf /tmp/q$i
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/q$i -o p -- python3 $R/bench.py --isolated-only --steps 2 --warmup 1 > /dev/null 2> $O/${TAG}_sq_pass$i.err
  cp $(find /tmp/q$i -name "*counter_collection.csv" | head -1) $O/${TAG}_sq_pass$i.csv
done
python3 $R/scripts/pmc_sq_summary.py $O/${TAG}_sq_pass1.csv $O/${TAG}_sq_pass2.csv > $O/${TAG}_sq_summary.txt
rm -f $O/${TAG}_sq_pass1.csv $O/${TAG}_sq_pass2.csv
