R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_deterministic_gpu.py tests/test_gemm_range_gpu.py -q -x 2>&1 | grep -E "^E  |passed|failed|FAILED" | tail -6
PCRCG_DEBUG=deterministic=1 timeout 400 python -m pytest tests/test_model_gpu.py tests/test_train_step_gpu.py tests/test_autograd_gpu.py -q 2>&1 | grep -E "passed|failed|FAILED" | tail -8
timeout 200 python bench.py --fixed-jobs --no-cpu-baseline --no-extras --steps 50 --repeats 3 2>/dev/null | tail -1 > gpurun_out/r05_bench_fixed_jobs.json
PCRCG_DEBUG=deterministic=1 timeout 200 python bench.py --fixed-jobs --no-cpu-baseline --no-extras --steps 50 --repeats 3 2>/dev/null | tail -1 > gpurun_out/r05_bench_deterministic.json
timeout 300 python scripts/bench_train.py --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r05_train_bench_b.json
PCRCG_DEBUG=deterministic=1 timeout 300 python scripts/bench_train.py --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r05_train_bench_deterministic.json
python - <<'PY'
import json
for f in ("r05_bench_fixed_jobs","r05_bench_deterministic"):
    d=json.load(open(f"gpurun_out/{f}.json")); print(f, d["value"], d["repeats"]["pairs_per_s"])
for f in ("r05_train_bench_b","r05_train_bench_deterministic"):
    d=json.load(open(f"gpurun_out/{f}.json")); print(f, d["ms_per_step"])
PY
