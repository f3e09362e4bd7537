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
