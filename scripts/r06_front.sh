#!/bin/bash
# GPU box: front-end work of round 6 -- the front-end test files, then driver-setting / long-region bench lines per side-stream
# setting and the front chain under the tracer.  usage: r06_front.sh TAG [quick]
TAG=${1:-r06_front}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "${2:-}" != "quick" ]; then
timeout 900 python3 -m pytest tests/test_frontend_gpu.py tests/test_pairstream_gpu.py tests/test_tieorder_gpu.py tests/test_radius_cells_gpu.py tests/test_c_host_gpu.py -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15 > $O/${TAG}_pytest.txt
cat $O/${TAG}_pytest.txt
fi
cd /tmp && export TMPDIR=/tmp
{
for FS in 2 1 0; do
  for ST in 20 480; do
    W=5; RP=5; [ $ST = 480 ] && RP=3
    v=$(PCRCG_FOREST_STREAM=$FS python3 $R/bench.py --steps $ST --warmup 5 --repeats $RP --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['pairs_per_s'])")
    echo "side_streams=$FS steps=$ST : $v"
  done
done
} > $O/${TAG}_ab.txt 2>&1
cat $O/${TAG}_ab.txt
db() { find "$1" -name "*results.db" | head -1; }
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p -- python3 $R/bench.py --steps 48 --warmup 5 --repeats 1 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_under_rocprof.json
python3 $R/scripts/prof_summary.py $(db /tmp/p1) $O/${TAG}_kernel_stats.csv 82
python3 $R/scripts/front_chain.py $(db /tmp/p1) > $O/${TAG}_front_chain.txt 2>&1
head -32 $O/${TAG}_front_chain.txt
rm -rf /tmp/p1; rocprofv3 --kernel-trace -d /tmp/p1 -o p -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_g20.json
python3 $R/scripts/region_gantt.py $(db /tmp/p1) 0.3 1.0 1500 2 > $O/${TAG}_gantt20.txt 2>&1
cat $O/${TAG}_gantt20.txt
