R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_train_step_gpu.py tests/test_autograd_gpu.py -q -x 2>&1 | grep -E "^E  |passed|failed|FAILED" | tail -8
for i in 1 2; do timeout 300 python scripts/bench_train.py --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train ms/step', d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 $R/scripts/bench_train.py --steps 10 --warmup 3 > /dev/null 2>&1
python3 $R/scripts/prof_summary.py $(find /tmp/pt -name "*results.db" | head -1) $R/gpurun_out/r05b_train_kernel_stats.csv 13
head -16 $R/gpurun_out/r05b_train_kernel_stats.csv | cut -c1-120
