#!/bin/bash
# GPU box, tuning aid: kernel durations of k_radius_cells on the S30k tables + instruction counts of the conv0 table
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*results.db" | head -1; }
python3 $R/scripts/radius_bench.py S30k T30k --reps 3 2>&1 | grep -E "sum|rror|assert"
rm -rf /tmp/pr; rocprofv3 --kernel-trace --stats -d /tmp/pr -o p -- python3 $R/scripts/radius_bench.py S30k --mode new --reps 10 > /tmp/pr.log 2>&1
python3 $R/scripts/radius_kernel_times.py $(db /tmp/pr) 10 k_radius_cells | grep -E "conv0|pool0|up0|conv1 |sum"
$R/scripts/pmc_radius_insts.sh 2>&1 | grep -E "VALU|SALU|BRANCH|LDS"
