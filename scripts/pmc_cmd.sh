#!/bin/bash
# GPU box: two SQ-counter passes (rocprofv3 --pmc, counters only + kernel trace) over an arbitrary python command,
# per-kernel summary.   usage: scripts/pmc_cmd.sh TAG script.py [args...]   -> gpurun_out/TAG_sq_summary.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rm -rf /tmp/q$i
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/q$i -o p -- python3 "$@" > /dev/null 2> $O/${TAG}_sq_pass$i.err
  cp $(find /tmp/q$i -name "*counter_collection.csv" | head -1) $O/${TAG}_sq_pass$i.csv
done
python3 $R/scripts/pmc_sq_summary.py $O/${TAG}_sq_pass1.csv $O/${TAG}_sq_pass2.csv > $O/${TAG}_sq_summary.txt
rm -f $O/${TAG}_sq_pass1.csv $O/${TAG}_sq_pass2.csv
cat $O/${TAG}_sq_summary.txt
