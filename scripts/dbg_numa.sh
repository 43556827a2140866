R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python - <<'PY'
import glob, os
print("affinity", len(os.sched_getaffinity(0)), "cpus")
for d in sorted(glob.glob("/sys/devices/system/node/node*")):
    print(os.path.basename(d), open(d + "/cpulist").read().strip())
for p in sorted(glob.glob("/dev/dri/renderD*")):
    ok = os.access(p, os.R_OK | os.W_OK)
    n = os.path.basename(p)
    try: numa = open(f"/sys/class/drm/{n}/device/numa_node").read().strip()
    except Exception as e: numa = repr(e)
    print(p, "accessible" if ok else "-", "numa", numa)
PY
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'K120k', d['secondary']['K120k']['value'], 'train', d['secondary']['train_step']['ms_per_step'])"; }
N0=$(cat /sys/devices/system/node/node0/cpulist); N1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
PCRCG_RANK_CPUS=$N0 timeout 300 python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | show "[node0 cpus]"
[ -n "$N1" ] && PCRCG_RANK_CPUS=$N1 timeout 300 python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | show "[node1 cpus]"
timeout 300 python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | show "[unpinned]"
