cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/p2; rocprofv3 --kernel-trace --stats -d /tmp/p2 -o p -- python3 $R/bench.py --isolated-only --steps 6 --warmup 2 > /dev/null 2>&1
python3 $R/scripts/forward_sequence.py $(find /tmp/p2 -name "*results.db" | head -1) > $R/gpurun_out/r05_forward_sequence.txt 2>&1
