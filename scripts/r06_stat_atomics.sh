cd /tmp
for cfg in "stat_sums=1" "stat_sums=0"; do
  PCRCG_DEBUG=$cfg python3 $GRAFT_REPO_ROOT/bench.py --isolated-only --steps 20 --warmup 3 2>/dev/null | tail -1 > /tmp/iso_$cfg.json
done
python3 - <<'PY'
import json
a=json.load(open('/tmp/iso_stat_sums=1.json')); b=json.load(open('/tmp/iso_stat_sums=0.json'))
print('forward ms', a['forward_ms'], b['forward_ms'], 'gemm ms/pair', a['gemm']['kernel_ms_per_pair'], b['gemm']['kernel_ms_per_pair'])
bs={(s['m'],s['n'],s['k']):s for s in b['gemm_by_shape']}
for s in a['gemm_by_shape']:
    t=bs.get((s['m'],s['n'],s['k']))
    if s['m']>=15000: print(s['m'],s['n'],s['k'],'sums',s['avg_us'],'partials',t['avg_us'] if t else None, 'per_forward', s['per_forward'])
PY
