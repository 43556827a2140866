R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for spec in "" "stat_sums_rows=20000" "x6_tile=1" "x6_tile=3" "x6_big=1"; do
  echo "== [$spec]"
  PCRCG_DEBUG=$spec python bench.py --isolated-only --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('forward_ms', d['forward_ms'], 'gemm ms', d['gemm']['kernel_ms_per_pair'])
for g in d['gemm_by_shape']:
    if g['m']>=15000 and g['k']<=256: print('   %6d x %4d x %4d n=%g avg %6.1f us'%(g['m'],g['n'],g['k'],g['per_forward'],g['avg_us']))
"
done
