# GPU box: the round's final evidence in one call (every leg bounded)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TAG=${1:-r06}
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "^E  |passed|failed|FAILED|rror" | tail -8 > gpurun_out/${TAG}_gpu_tests.log
timeout 1500 bash scripts/collect_profiles.sh $TAG all > gpurun_out/${TAG}_collect.log 2>&1
# the price of deterministic=1: the engine with a fixed pairing, and the train step
timeout 200 python bench.py --fixed-jobs --no-cpu-baseline --no-extras --repeats 3 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_fixed_jobs.json
PCRCG_DEBUG=deterministic=1 timeout 200 python bench.py --fixed-jobs --no-cpu-baseline --no-extras --repeats 3 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_deterministic.json
PCRCG_DEBUG=deterministic=1 timeout 300 python scripts/bench_train.py --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/${TAG}_train_bench_deterministic.json
for w in S30k K120k; do timeout 120 python scripts/redo_probe.py $w 2>&1 | tail -1; done > gpurun_out/${TAG}_gemm_redo_counts.txt
