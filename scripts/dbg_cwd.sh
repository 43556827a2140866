R=${GRAFT_REPO_ROOT:-/root/repo}
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'K120k', d['secondary']['K120k']['value'], 'train', d['secondary']['train_step']['ms_per_step'])"; }
cd $R; timeout 300 python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | show "[repo cwd]"
cd /tmp; timeout 300 python $R/bench.py --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | show "[/tmp cwd]"
cd /tmp; export TMPDIR=/tmp; timeout 300 python3 $R/bench.py --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | show "[/tmp cwd, TMPDIR, python3]"
cd $R; unset TMPDIR; timeout 400 python bench.py 2> gpurun_out/r05_bench_direct.err | tail -1 > gpurun_out/r05_bench_direct.json; cat gpurun_out/r05_bench_direct.json | show "[repo cwd, full default]"
