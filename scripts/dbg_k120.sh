R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_deterministic_gpu.py -x -q 2>&1 | grep -E "^E  |passed|failed|FAILED" | tail -5 > gpurun_out/r05_t7.log
for spec in "" "gnn_merge=0,edge_rows=0,att_mfma=0" ""; do
  PCRCG_DEBUG=$spec python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$spec]', d['value'], 'K120k', d['secondary']['K120k']['value'], 'train', d['secondary']['train_step']['ms_per_step'])"
done
cp pcrcg_amd/libpcrcg_hip.so /tmp/cur.so; cp ab/c1.so pcrcg_amd/libpcrcg_hip.so
python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[c1 lib]', d['value'], 'K120k', d['secondary']['K120k']['value'], 'train', d['secondary']['train_step']['ms_per_step'])"
cp /tmp/cur.so pcrcg_amd/libpcrcg_hip.so
