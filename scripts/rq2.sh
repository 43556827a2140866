#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*results.db" | head -1; }
for V in "$@"; do
export PCRCG_DEBUG=$V
rm -rf /tmp/pr; rocprofv3 --kernel-trace --stats -d /tmp/pr -o p -- python3 $R/scripts/radius_bench.py S30k --mode new --reps 10 > /tmp/pr.log 2>&1
echo "## $V"; python3 $R/scripts/radius_kernel_times.py $(db /tmp/pr) 10 k_radius_cells | grep -E "conv0|pool0|up0|conv1 |conv3|sum"
done
