"""CPU: the N-rank launch path of bench.py / scripts/bench_train.py (pcrcg_amd/launcher.py).

`python bench.py --gpus N` with no rank environment must start N fresh rank processes itself, hand them
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and one CPU set each, and relay rank 0's single JSON line; N above the
visible device count must exit non-zero.  The dry run exercises exactly that over gloo (no GPU in this container).
The reference has no counterpart (ref:main.py:15 one device); SURVEY.md 8e is the specification."""
import json
import os
import subprocess
import sys

import pytest

from pcrcg_amd import launcher

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PYTHONPATH"] = REPO + os.pathsep + env.get("PYTHONPATH", "")
    return env


def test_cpulist_round_trip():
    assert launcher.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert launcher.format_cpulist([11, 0, 1, 2, 3, 8, 10]) == "0-3,8,10-11"
    assert launcher.parse_cpulist("") == []


def test_even_split_without_topology():
    plan = launcher.plan_affinity(4, range(16))
    assert plan == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]
    plan = launcher.plan_affinity(8, range(6))            # fewer CPUs than ranks: everyone still gets one
    assert all(len(p) == 1 for p in plan) and len(plan) == 8


def _fake_sysfs(root, gpus, nodes):
    """gpus: [(kfd node id, domain, bus, numa)], nodes: {numa: cpulist}; KFD node 0/1 are CPU nodes."""
    for i in (0, 1):
        d = root / "class/kfd/kfd/topology/nodes" / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for nid, dom, bus, numa in gpus:
        d = root / "class/kfd/kfd/topology/nodes" / str(nid)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain %d\n" % (bus << 8, dom))
        p = root / "bus/pci/devices" / ("%04x:%02x:00.0" % (dom, bus))
        p.mkdir(parents=True)
        (p / "numa_node").write_text("%d\n" % numa)
    for numa, cpus in nodes.items():
        d = root / "devices/system/node" / ("node%d" % numa)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")


def test_numa_near_plan_from_sysfs(tmp_path):
    # 8 GPUs, KFD nodes 2..9 (10 sorts after 9 numerically, not lexically), four per socket; SMT siblings in the lists
    gpus = [(2 + i, 0, 0x05 + 0x10 * i, 0 if i < 4 else 1) for i in range(8)]
    _fake_sysfs(tmp_path, gpus, {0: "0-63,128-191", 1: "64-127,192-255"})
    numa = launcher.gpu_numa_nodes(str(tmp_path))
    assert numa == [0, 0, 0, 0, 1, 1, 1, 1]
    cpus = launcher.numa_cpus(str(tmp_path))
    plan = launcher.plan_affinity(8, range(256), numa, cpus)
    assert all(len(p) == 32 for p in plan)
    flat = [c for p in plan for c in p]
    assert len(set(flat)) == 256                                       # nobody shares a CPU
    for r in range(8):
        assert set(plan[r]) <= set(cpus[0 if r < 4 else 1])           # every rank on its GPU's socket
    # two ranks on one socket take halves of it
    plan2 = launcher.plan_affinity(2, range(256), numa, cpus)
    assert set(plan2[0]) | set(plan2[1]) == set(cpus[0])
    # visible-device remapping
    assert launcher.gpu_numa_nodes(str(tmp_path), visible=[7, 0]) == [1, 0]
    # container restricted to a few CPUs of one socket: falls back to an even split of what is allowed
    plan3 = launcher.plan_affinity(8, range(8), numa, cpus)
    assert sorted(c for p in plan3 for c in p) == list(range(8))


def test_ranks_under_torchrun_plan_their_own_cpu_set(monkeypatch):
    """torch.distributed.run sets LOCAL_RANK / LOCAL_WORLD_SIZE and no PCRCG_RANK_CPUS: every rank derives the same plan and
    takes its own entry (child processes: the affinity of the test process itself is left alone)."""
    code = ("import os, sys; sys.path.insert(0, %r); from pcrcg_amd import launcher; "
            "c = launcher.apply_rank_affinity(); print(launcher.format_cpulist(c) if c else '-', len(os.sched_getaffinity(0)))" % REPO)
    allowed = sorted(os.sched_getaffinity(0))
    outs = []
    for r in range(2):
        env = _env()
        env.update({"LOCAL_RANK": str(r), "LOCAL_WORLD_SIZE": "2"})
        env.pop(launcher.CPUS_ENV, None)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        outs.append(out.stdout.split())
    if len(allowed) >= 2:
        a, b = (set(launcher.parse_cpulist(o[0])) for o in outs)
        assert a and b and not (a & b) and (a | b) <= set(allowed)
        assert int(outs[0][1]) == len(a)                                # the set was applied to the process
    env = _env()
    env.update({"LOCAL_RANK": "0", "LOCAL_WORLD_SIZE": "2", "PCRCG_NO_AFFINITY": "1"})
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.stdout.split()[0] == "-"


def test_parent_detection(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    assert launcher.is_parent(2) and not launcher.is_parent(1)
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert not launcher.is_parent(2)                                   # under torch.distributed.run: a rank


@pytest.mark.parametrize("script", ["bench.py", os.path.join("scripts", "bench_train.py")])
def test_two_rank_dry_run_line_shape(script):
    cmd = [sys.executable, os.path.join(REPO, script), "--gpus", "2", "--launcher-dry-run", "--steps", "3", "--warmup", "1"]
    if script == "bench.py":
        cmd += ["--repeats", "2"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                                   # ONE JSON line on stdout, nothing else
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["dry_run"] is True
    assert len(line["per_rank_cpus"]) == 2 and all(line["per_rank_cpus"])
    assert line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    for k in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert k in line
    if script == "bench.py":
        assert len(line["per_rank_pairs_per_s"]) == 2
        # weak scaling: at step i rank r works on global pair i*world + r; every pair exactly once
        seeds = line["config"]["pair_seeds_first_region"]
        assert sorted(seeds[0] + seeds[1]) == list(range(6))
        # ranks' CPU sets are disjoint when the container has at least two CPUs
        a, b = (set(launcher.parse_cpulist(c)) for c in line["per_rank_cpus"])
        if len(os.sched_getaffinity(0)) >= 2:
            assert not (a & b)
    else:
        assert line["allreduce"]["elements"] > 0 and line["allreduce"]["ms"] >= 0
        assert line["replicas_identical"] is True


def test_more_ranks_than_devices_exits_nonzero():
    # no GPU in the CPU container -> 0 visible devices; on a GPU box 99 exceeds any node
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "99", "--steps", "1"], env=_env(),
                       capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode != 0
    assert "visible" in r.stderr
    assert not r.stdout.strip()


def test_eight_rank_dry_run_on_a_two_socket_node(tmp_path):
    """`--gpus 8` kept boring: eight dry-run ranks (gloo) planned on a stand-in two-socket machine -- four GPUs per NUMA
    node, 2 x 64 cores with SMT siblings -- must report eight ranks, eight disjoint CPU sets of at least four CPUs, every
    rank on its GPU's socket, and every pair seed exactly once.  (No 8-GPU hardware was available to this build: this
    covers the protocol, not a scaling number.)"""
    gpus = [(2 + i, 0, 0x05 + 0x10 * i, 0 if i < 4 else 1) for i in range(8)]
    nodes = {0: "0-63,128-191", 1: "64-127,192-255"}
    _fake_sysfs(tmp_path, gpus, nodes)
    env = _env()
    env[launcher.SYSFS_ENV] = str(tmp_path)
    env["OMP_NUM_THREADS"] = "1"
    assert launcher.visible_gpu_count(str(tmp_path)) == 8
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--launcher-dry-run", "--steps", "2",
                        "--warmup", "1", "--repeats", "1"], env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["scaling"] == "weak"
    sets = [set(launcher.parse_cpulist(c)) for c in line["per_rank_cpus"]]
    assert len(sets) == 8 and all(len(s) >= 4 for s in sets)
    assert sum(len(s) for s in sets) == len(set().union(*sets))            # disjoint
    socket_cpus = {k: set(launcher.parse_cpulist(v)) for k, v in nodes.items()}
    for rank, s in enumerate(sets):
        assert s <= socket_cpus[0 if rank < 4 else 1]
    seeds = line["config"]["pair_seeds_first_region"]
    assert sorted(x for per_rank in seeds for x in per_rank) == list(range(2 * 8))
    assert len(line["per_rank_pairs_per_s"]) == 8
