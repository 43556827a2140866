#!/bin/bash
# GPU box: per-dispatch FETCH_SIZE / WRITE_SIZE of the GEMM launches of isolated forwards (separate counter passes).
# usage: scripts/pmc_gemm_dispatches.sh TAG   -> gpurun_out/TAG_pmc_gemm_dispatches.txt
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pg; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pg -o p -- python3 $R/bench.py --isolated-only --steps 2 --warmup 1 > /dev/null 2>&1
  cp $(find /tmp/pg -name "*counter_collection.csv" | head -1) /tmp/pg_$C.csv
done
python3 - <<PY > $O/${TAG}_pmc_gemm_dispatches.txt
import csv
def load(p):
    rows=[r for r in csv.DictReader(open(p))]
    return rows
f=load('/tmp/pg_FETCH_SIZE.csv'); w=load('/tmp/pg_WRITE_SIZE.csv')
print(list(f[0].keys()))
fg=[r for r in f if 'k_gemm_x6' in r['Kernel_Name']]
wg=[r for r in w if 'k_gemm_x6' in r['Kernel_Name']]
n=min(len(fg),len(wg))
per=n//3 if n%3==0 else n
for a,b in list(zip(fg,wg))[-per:]:
    print(a['Kernel_Name'][:48], a['Grid_Size'], a.get('Workgroup_Size'), "fetch_MB %.1f"%(2*float(a['Counter_Value'])*1024/1e6), "write_MB %.1f"%(float(b['Counter_Value'])*1024/1e6))
PY
tail -80 $O/${TAG}_pmc_gemm_dispatches.txt
