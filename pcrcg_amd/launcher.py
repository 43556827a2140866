"""One-node launcher: `bench.py --gpus N` (and scripts/bench_train.py) start their own N ranks.

The reference has no counterpart -- ref:main.py:15 hard-codes one device and ref:datasets/dataloader.py:207 one pair
per batch -- BASELINE.json's north_star adds "throughput at 1/2/4/8 GPUs" (pair i -> GPU i mod G, no
data-path collective).  The user-facing process becomes a PARENT that never touches the GPU: it starts one fresh child
process per GPU (fork + exec of an interpreter that has made no HIP call -- a process that has initialised the GPU is
never replaced or forked), hands each child RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT and a CPU set
near its GPU, waits, and relays rank 0's single JSON line.  Under `torch.distributed.run` (WORLD_SIZE already set) the
scripts are ranks themselves and this module only applies the CPU set when one was handed over.

Pure host logic, no torch import at module level: tests/test_launcher_cpu.py drives it on CPU over gloo.
"""
import glob
import json
import os
import socket
import subprocess
import sys
import threading

CPUS_ENV = "PCRCG_RANK_CPUS"          # "0-15,64-79": the CPU set the parent planned for this rank
# PCRCG_SYSFS_ROOT (tests only): a stand-in /sys tree.  The topology is then read from it, the "available" CPUs are the
# union of its NUMA nodes' lists, and ranks REPORT the set planned for them without applying it (the stand-in machine's
# CPUs do not exist here) -- how tests/test_launcher_cpu.py plans an 8-GPU two-socket node inside an 8-CPU container.
SYSFS_ENV = "PCRCG_SYSFS_ROOT"


def _sysfs():
    return os.environ.get(SYSFS_ENV) or "/sys"


def _available_cpus():
    if os.environ.get(SYSFS_ENV):
        return sorted({c for cpus in numa_cpus().values() for c in cpus})
    try:
        return sorted(os.sched_getaffinity(0))
    except AttributeError:
        return list(range(os.cpu_count() or 1))


def is_parent(gpus):
    """True in the process the user (or the driver) started with --gpus N > 1 and no rank environment."""
    return gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def format_cpulist(cpus):
    cpus = sorted(set(cpus))
    parts, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        parts.append(str(cpus[i]) if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(parts)


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def gpu_numa_nodes(sysfs=None, visible=None):
    """NUMA node of every GPU in HIP device order, from sysfs only (no HIP call in the parent): KFD topology nodes
    with simd_count > 0 are the GPUs, in the order the runtime enumerates them; `domain` + `location_id`
    (bus << 8 | devfn) name the PCI function whose `numa_node` is read.  -> list (None where unknown).  `visible`:
    the index list of HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES, applied afterwards."""
    sysfs = sysfs or _sysfs()
    nodes = []
    for d in sorted(glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/*")),
                    key=lambda p: int(os.path.basename(p)) if os.path.basename(p).isdigit() else 1 << 30):
        txt = _read(os.path.join(d, "properties"))
        if txt is None:
            continue
        props = {}
        for line in txt.splitlines():
            k, _, v = line.partition(" ")
            if v.strip().lstrip("-").isdigit():
                props[k] = int(v)
        if props.get("simd_count", 0) <= 0:
            continue
        loc, dom = props.get("location_id", 0), props.get("domain", 0)
        bdf = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7)
        numa = _read(os.path.join(sysfs, "bus/pci/devices", bdf, "numa_node"))
        numa = int(numa) if numa is not None and numa.strip().lstrip("-").isdigit() else None
        nodes.append(numa if numa is not None and numa >= 0 else None)
    if visible:
        nodes = [nodes[i] if 0 <= i < len(nodes) else None for i in visible]
    return nodes


def numa_cpus(sysfs=None):
    sysfs = sysfs or _sysfs()
    out = {}
    for d in glob.glob(os.path.join(sysfs, "devices/system/node/node*")):
        name = os.path.basename(d)[4:]
        txt = _read(os.path.join(d, "cpulist"))
        if name.isdigit() and txt is not None:
            out[int(name)] = parse_cpulist(txt)
    return out


def _visible_indices():
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            try:
                return [int(x) for x in v.split(",") if x.strip() != ""]
            except ValueError:
                return None
    return None


def plan_affinity(n, available, gpu_numa=None, node_cpus=None):
    """CPU set per rank: the cores of the NUMA node its GPU hangs off, divided evenly among the ranks that share the
    node; an even contiguous split of `available` where the topology is unknown.  Every rank gets at least one CPU and
    no CPU is given to two ranks unless there are fewer CPUs than ranks.  (A rank runs 1 front-end + 3 model host
    threads that sit in library calls; SURVEY.md 8e names host staging as the expected scaling limiter.)"""
    available = sorted(set(available))
    if n <= 0 or not available:
        return [list(available) for _ in range(max(n, 0))]
    plan = [None] * n
    if gpu_numa and node_cpus and len(gpu_numa) >= n and all(g is not None and g in node_cpus for g in gpu_numa[:n]):
        by_node = {}
        for r in range(n):
            by_node.setdefault(gpu_numa[r], []).append(r)
        ok = True
        for node, ranks in by_node.items():
            cpus = [c for c in node_cpus[node] if c in set(available)]
            if len(cpus) < len(ranks):
                ok = False
                break
            per = len(cpus) // len(ranks)
            for i, r in enumerate(ranks):
                plan[r] = cpus[i * per:(i + 1) * per]
        if ok:
            return plan
    if len(available) < n:
        return [[available[r % len(available)]] for r in range(n)]
    per = len(available) // n
    return [available[r * per:(r + 1) * per] for r in range(n)]


def apply_rank_affinity():
    """In a rank: pin the process (and the threads it will start) to a CPU set near its GPU.  Started by launch(): the
    set the parent planned (PCRCG_RANK_CPUS).  Started by torch.distributed.run (LOCAL_RANK / LOCAL_WORLD_SIZE set, more
    than one rank on the node): every rank computes the same plan from sysfs and takes its own entry -- the driver's
    launch gets the NUMA-near placement too.  PCRCG_NO_AFFINITY=1 switches it off.  -> the set, or None."""
    if os.environ.get("PCRCG_NO_AFFINITY") == "1":
        return None
    txt = os.environ.get(CPUS_ENV)
    try:
        if txt:
            cpus = parse_cpulist(txt)
        else:
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
            local_rank = int(os.environ.get("LOCAL_RANK", "0"))
            if local_world <= 1 or not (0 <= local_rank < local_world):
                return None
            cpus = plan_affinity(local_world, _available_cpus(), gpu_numa_nodes(visible=_visible_indices()), numa_cpus())[local_rank]
        if not cpus:
            return None
        if not os.environ.get(SYSFS_ENV):
            os.sched_setaffinity(0, cpus)
    except (OSError, ValueError, AttributeError):
        return None
    return cpus


def visible_gpu_count(sysfs=None):
    """Devices the children will see, WITHOUT loading the HIP runtime in the parent: the KFD topology's GPU nodes (the
    enumeration gpu_numa_nodes() walks), cut down by HIP_/ROCR_/CUDA_VISIBLE_DEVICES.  Only where sysfs shows no KFD
    topology at all (a container without it) does the parent ask torch.cuda.device_count(), which on this image counts
    through amdsmi/sysfs too but may initialise HIP elsewhere."""
    sysfs = sysfs or _sysfs()
    n = len(gpu_numa_nodes(sysfs, visible=_visible_indices()))
    if n > 0 or os.path.isdir(os.path.join(sysfs, "class/kfd/kfd/topology/nodes")):
        if sysfs == "/sys" and _visible_indices() is None:
            # a container may be handed fewer GPUs than the host's sysfs lists: it then holds only their render nodes
            render = [p for p in glob.glob("/dev/dri/renderD*") if os.access(p, os.R_OK | os.W_OK)]
            if render:
                n = min(n, len(render))
        return n
    import torch
    return int(torch.cuda.device_count())


DEFAULT_TIMEOUT_S = 3600.0      # a rank wedged in a collective must not block the parent forever


def launch(script, argv, gpus, dry_run=False, env_extra=None, timeout=DEFAULT_TIMEOUT_S):
    """Start `gpus` ranks of `script argv...`, relay rank 0's LAST JSON line on stdout (everything else any rank
    prints goes to stderr, prefixed with its rank) and return the exit code: 0 only if every rank exited 0 and rank 0
    printed a line.  dry_run: no device check (the ranks run their CPU / gloo stand-in).  timeout (seconds, None: wait
    forever; PCRCG_LAUNCH_TIMEOUT overrides): when it expires the remaining ranks are killed and 124 is returned."""
    if os.environ.get("PCRCG_LAUNCH_TIMEOUT"):
        try:
            timeout = float(os.environ["PCRCG_LAUNCH_TIMEOUT"]) or None
        except ValueError:
            pass
    if not dry_run:
        have = visible_gpu_count()
        if gpus > have:
            print("launcher: --gpus %d but only %d device(s) are visible" % (gpus, have), file=sys.stderr)
            return 2
    plan = plan_affinity(gpus, _available_cpus(), gpu_numa_nodes(visible=_visible_indices()), numa_cpus())
    port = free_port()
    procs, pumps, last_json = [], [], [None]

    def pump(rank, stream):
        for raw in stream:
            line = raw.rstrip("\n")
            if rank == 0 and line.startswith("{") and line.endswith("}"):
                try:
                    json.loads(line)
                    if last_json[0] is not None:
                        print("[rank 0] " + last_json[0], file=sys.stderr, flush=True)
                    last_json[0] = line
                    continue
                except ValueError:
                    pass
            print("[rank %d] %s" % (rank, line), file=sys.stderr, flush=True)

    for r in range(gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(gpus), "LOCAL_WORLD_SIZE": str(gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), CPUS_ENV: format_cpulist(plan[r]),
                    "PCRCG_LAUNCHED_BY": "pcrcg_amd.launcher"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, len(plan[r]))))
        if env_extra:
            env.update(env_extra)
        p = subprocess.Popen([sys.executable, "-u", script] + list(argv), env=env, stdout=subprocess.PIPE,
                             stderr=None, text=True, bufsize=1)
        procs.append(p)
        t = threading.Thread(target=pump, args=(r, p.stdout), daemon=True)
        t.start()
        pumps.append(t)

    rc = 0
    try:
        pending = set(range(gpus))
        import time
        t_end = None if timeout is None else time.monotonic() + timeout
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print("launcher: rank %d exited with %d; stopping the other ranks" % (r, code), file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()          # our own children, by PID
            if pending:
                if t_end is not None and time.monotonic() > t_end:
                    print("launcher: timeout after %.0f s; killing rank(s) %s" % (timeout, sorted(pending)), file=sys.stderr)
                    rc = rc or 124
                    for q in pending:
                        procs[q].kill()
                    break
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
        for t in pumps:
            t.join(timeout=5)
    if rc == 0 and last_json[0] is None:
        print("launcher: rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    if last_json[0] is not None and rc == 0:
        sys.stdout.flush()
        print(last_json[0], flush=True)
    return rc


def rank_fields(dist, world, rank, local_value, cpus, device=None):
    """What every rank contributes to rank 0's line: -> dict with ranks_seen (from the process group), per-rank values
    and CPU sets.  `dist` is torch.distributed (initialised) or None for a single process.  Plain tensor collectives
    only (one all_gather of a fixed-size record on `device`, the process group's device): no pickled objects through
    RCCL."""
    text = format_cpulist(cpus) if cpus else ""
    if dist is None:
        return {"ranks_seen": 1, "per_rank_value": [local_value], "per_rank_cpus": [text or None]}
    import torch
    rec = torch.zeros(2 + 240, dtype=torch.float64)
    rec[0], rec[1] = float(rank), float(local_value)
    raw = text.encode()[:240]
    rec[2:2 + len(raw)] = torch.tensor(list(raw), dtype=torch.float64)
    rec = rec.to(device) if device is not None else rec
    got = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(got, rec)
    rows = sorted((t.cpu() for t in got), key=lambda t: int(t[0]))
    cpus_out = []
    for t in rows:
        b = bytes(int(v) for v in t[2:].tolist() if int(v) != 0)
        cpus_out.append(b.decode() or None)
    return {"ranks_seen": len({int(t[0]) for t in rows}),
            "per_rank_value": [round(float(t[1]), 3) for t in rows],
            "per_rank_cpus": cpus_out}
