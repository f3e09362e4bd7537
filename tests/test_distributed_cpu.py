"""world_size-2 gloo test of the multi-GPU host path (runs on the CPU)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_gloo_sharding_and_aggregation():
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "tests", "dist_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "DIST_OK 2" in p.stdout


def test_bench_launches_its_own_ranks_and_shards_the_proteome():
    """`python bench.py --gpus 2` with no launcher around it: the script starts torch.distributed.run
    itself (before touching any GPU), n_gpus comes from the process group, and the strong-scaling
    shards are disjoint, complete and balanced.  --dry-run: gloo on the CPU, no compute."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run",
                        "--steps", "3", "--structures", "48"], capture_output=True, text=True, env=env,
                       timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 3
    assert out["shards_disjoint_and_complete"] is True
    assert sum(out["shard_structures"]) == 48 == out["config"]["structures_total"]
    a = out["shard_atoms"]
    assert abs(a[0] - a[1]) <= 0.02 * sum(a)


def test_bench_single_rank_dry_run_has_no_launcher():
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--structures", "12"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert out["n_gpus"] == 1 and out["shard_structures"] == [12]


def test_ranks_bind_to_their_gpus_numa_node(tmp_path):
    """Every rank pins itself to the CPUs of the NUMA node its GPU hangs off (bench.numa_bind: sysfs numa_node /
    local_cpulist of the GPU's PCI address, sched_setaffinity, no exec) before it allocates anything.  Dry run:
    two gloo ranks, a sysfs tree and two PCI addresses made up here."""
    import json
    have = sorted(os.sched_getaffinity(0))
    if len(have) < 2:
        import pytest
        pytest.skip("needs two CPUs")
    half = len(have) // 2
    plan = {"0000:05:00.0": (0, have[:half]), "0000:85:00.0": (1, have[half:])}
    for bdf, (node, cpus) in plan.items():
        d = tmp_path / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
        (d / "local_cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(RSASA_DRYRUN_SYSFS=str(tmp_path), RSASA_DRYRUN_GPU_PCI=",".join(plan))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run",
                        "--steps", "1", "--structures", "16"], capture_output=True, text=True, env=env,
                       timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert len(out["numa"]) == 2
    for rank, (bdf, (node, cpus)) in enumerate(plan.items()):
        got = out["numa"][rank]
        assert got["gpu_pci"] == bdf and got["numa_node"] == node
        assert got["affinity_after"] == cpus and got["cpus"] == len(cpus)


def test_numa_bind_leaves_single_node_machines_alone(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    before = os.sched_getaffinity(0)
    d = tmp_path / "bus" / "pci" / "devices" / "0000:01:00.0"
    d.mkdir(parents=True)
    (d / "numa_node").write_text("-1\n")
    (d / "local_cpulist").write_text("0-255\n")
    info = bench.numa_bind("0000:01:00.0", str(tmp_path))
    assert info["numa_node"] == -1 and info["cpus"] is None and os.sched_getaffinity(0) == before
    assert bench.numa_bind(None)["numa_node"] is None
    assert bench.numa_bind("0000:ff:00.0", str(tmp_path))["numa_node"] is None  # no such device
    assert bench.parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11}
