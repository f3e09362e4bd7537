#!/usr/bin/env python3
"""Headline benchmark: structures/sec of the Shrake-Rupley hot path on MI355X.

One "step" = one pass of the whole hot path (bounds -> cell grid -> counting
sort -> occlusion -> ResidueLevel sums) over the rank's structures.  The
workload is BASELINE.json configs[2] / configs[3]: an AlphaFold-E.-coli-like
proteome (4 363 structures, 100 sphere points, probe 1.4 A, ResidueLevel),
synthesised offline-reproducibly by bench_workloads.synthetic_proteome.

`value` is measured with the inputs and outputs resident in HBM (the contract's
definition), the way a rank of a sharded run works through its stream of
batches: step k + 1 is enqueued before step k is waited for (the library keeps
two batches in flight per context, each in its own workspace).  The
one-batch-at-a-time rate is printed as `one_at_a_time`; the SURVEY 8d
host-to-host rate (pinned SoA in host memory in, per-residue values back in
host memory) is timed in the same run and printed as `host_to_host`.

Multi-GPU (`--gpus N`): one process per GPU.  Launched by the driver through
torch.distributed.run, or - when WORLD_SIZE is not set - by this script itself,
which starts torch.distributed.run as a child process before anything touches
a GPU.  Default is STRONG scaling (reference src/main.rs:375,439: independent
structures, one per worker): the same proteome, sorted largest first, is cut
into equal-atom shards, one per rank, no data-path collective;
RCCL carries only the barrier and the final MAX(time) / SUM(units).  The weak
scaling rate (every rank its own proteome) is printed as a secondary field.

Prints ONE JSON line on rank 0 (field notes: DESIGN.md "Measurement").
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_TFLOPS = 157.3        # f32 vector peak (MI355X_MICROARCH.md)
PROBE = 1.4
N_POINTS = 100
PMC_FILE = os.path.join("profiles", "pmc_occlusion.json")


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--prewarm-steps", type=int, default=16,
                   help="batches run ahead of the warm-up steps so that the device's clocks are up (0: none)")
    p.add_argument("--structures", type=int, default=None,
                   help="structures of the proteome (default: all 4 363)")
    p.add_argument("--workload", choices=["proteome", "uniform1m"], default="proteome")
    p.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                   help="strong: one proteome sharded over the ranks (default); weak: one proteome per rank")
    p.add_argument("--n-points", type=int, default=None)
    p.add_argument("--cpu-seconds", type=float, default=15.0,
                   help="target wall time of the CPU baseline sample (0 disables it and the parity diff)")
    p.add_argument("--h2h-steps", type=int, default=10,
                   help="timed steps of the host-to-host leg (0 disables it)")
    p.add_argument("--hashed-ids-steps", type=int, default=20,
                   help="steps of the secondary measurement with unordered ids (0 = skip)")
    p.add_argument("--two-steps", type=int, default=40,
                   help="steps of the secondary one-batch-at-a-time measurement (0 = skip)")
    p.add_argument("--shard-of", type=int, default=0,
                   help="one GPU only: run rank 0's shard of an M-way strong-scaling split (what each of M GPUs would get)")
    p.add_argument("--weak-steps", type=int, default=5,
                   help="timed steps of the secondary weak-scaling measurement at N > 1 (0 disables it)")
    p.add_argument("--config5-steps", type=int, default=30,
                   help="timed steps of the secondary 1M-atom / 960-point measurement (BASELINE.json configs[4]; 0 = skip)")
    p.add_argument("--real-steps", type=int, default=10,
                   help="steps of the real_coords leg (the reference's 88-file quality set, tiled to the proteome's size; 0: skip)")
    p.add_argument("--files", type=int, default=1000,
                   help="files of the secondary directory-mode measurement (files on /dev/shm -> residue values; 0 = skip)")
    p.add_argument("--e2e-files", type=int, default=4363,
                   help="files of the files_mode.end_to_end leg (files in -> one JSON file per input out, one call; 0: skip)")
    p.add_argument("--per-call-seconds", type=float, default=1.0,
                   help="seconds per leg of the secondary drop-in measurement: rsasa_calculate_sasa_internal once per "
                        "structure from 1 and from 16 host threads (0 = skip)")
    p.add_argument("--no-ids", action="store_true", help="pass id = NULL (all atoms distinct)")
    p.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                   help="process group of a multi-rank run: nccl (= RCCL, one rank per GPU) or gloo (barrier and the "
                        "MAX / SUM on the host: lets several ranks share one GPU, which RCCL cannot - tests only)")
    p.add_argument("--device", type=int, default=None,
                   help="GPU of this rank (default: LOCAL_RANK); with --dist-backend gloo every rank may name the same one")
    p.add_argument("--verify-shards", action="store_true",
                   help="every rank compares its own shard's atoms and residues with the oracle and rank 0 prints the "
                        "per-rank verdicts (small --structures counts: the oracle runs on the rank's whole shard)")
    p.add_argument("--dry-run", action="store_true",
                   help="CPU only (gloo): launcher, sharding and aggregation without any GPU work")
    return p.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child process.
    Nothing in this process has touched a GPU (torch is not even imported yet)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def parse_cpulist(text):
    """'0-3,8,10-11' -> {0, 1, 2, 3, 8, 10, 11}"""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def numa_bind(bdf, sysfs="/sys"):
    """Pins this process - and every thread it starts from here on: the library's coding threads, the pinned
    allocations' first touch - to the CPUs of the NUMA node the GPU `bdf` (PCI address) hangs off.  No exec, no
    numactl hop (a process that has initialised the GPU must not exec).  Returns what it did for the bench line."""
    info = {"numa_node": None, "cpus": None, "cpu_list": None, "gpu_pci": bdf}
    if not bdf:
        return info
    base = os.path.join(sysfs, "bus", "pci", "devices", bdf)
    try:
        node = int(open(os.path.join(base, "numa_node")).read())
        cpulist = open(os.path.join(base, "local_cpulist")).read().strip()
    except (OSError, ValueError):
        return info
    info["numa_node"] = node
    if node < 0 or not cpulist:
        return info  # a single-node machine (or a VM that hides the topology): nothing to pin
    cpus = parse_cpulist(cpulist) & os.sched_getaffinity(0)
    if cpus:
        os.sched_setaffinity(0, cpus)
        info["cpus"], info["cpu_list"] = len(cpus), cpulist
    return info


def gpu_pci_address(torch, index):
    """PCI address of torch's device `index` (after HIP_VISIBLE_DEVICES remapping), or None."""
    try:
        pr = torch.cuda.get_device_properties(index)
        return f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:
        return None


def cgroup_cpu_quota():
    """CPUs' worth of time this process's cgroup may use per period (cgroup v2 cpu.max, v1 cfs quota), or None."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def physical_cores():
    """Distinct (socket, core) pairs of /proc/cpuinfo (None when it does not say)."""
    seen, phys = set(), None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                seen.add((phys, line.split(":")[1].strip()))
    except OSError:
        pass
    return len(seen) or None


def cpu_baseline(batch, n_points, target_seconds):
    """The oracle (a port of the reference's CPU path) on a bounded sample of the same workload, SURVEY 8d:
    (i) one thread, (ii) all host threads with structure-level parallelism (mirrors src/main.rs:375,439: one
    structure per worker, the kernel sequential inside); same inputs, wall clock, best of three after a warm-up.
    The sample is every k-th structure of the rank's list (k = 1: all of it), so it has the list's size mix -
    not its largest structures.  Returns the bench-line object, the sampled structure indices and the oracle's
    per-atom values of the all-threads sample (concatenated in that order)."""
    import numpy as np
    import bench_workloads as bw
    from oracle import pyoracle as po
    # every CPU this process may really use: the hardware threads it is allowed on, capped by its cgroup's CPU quota
    # (a container with cpu.max = 16 CPUs on a 256-thread host runs 16 threads' worth of work however many it starts:
    # 128 threads there are throttled, not faster)
    hw = min(po.max_threads(), len(os.sched_getaffinity(0)), os.cpu_count() or 1)
    quota = cgroup_cpu_quota()
    threads = max(1, min(hw, int(quota + 0.5))) if quota else hw

    def run(b, n_threads):
        t0 = time.perf_counter()
        v = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE,
                                    n_points, 8, threads=n_threads)
        return time.perf_counter() - t0, v

    def leg(n_threads, seconds):
        # a probe on every 64th structure sizes the sample for about `seconds` per run
        probe = bw.select(batch, np.arange(0, batch.n_structures, 64))
        run(probe, n_threads)  # warm up the thread pool, the workers' arenas and the library
        t_probe, _ = run(probe, n_threads)
        t_all = t_probe * batch.n_atoms / max(probe.n_atoms, 1)  # the whole list at the probe's rate
        k = max(1, int(round(t_all / max(seconds, 1e-3))))
        idx = np.arange(0, batch.n_structures, k)
        b = bw.select(batch, idx)
        best, v = run(b, n_threads)
        for _ in range(2):
            t, v = run(b, n_threads)
            best = min(best, t)
        return idx, b, best, v

    # about a third of the budget per all-threads run, a fifth of that per one-thread run
    idx, b, t, v = leg(threads, target_seconds / 3.0)
    idx1, b1, t1, _ = leg(1, target_seconds / 15.0)
    every = int(idx[1] - idx[0]) if len(idx) > 1 else 1
    every1 = int(idx1[1] - idx1[0]) if len(idx1) > 1 else 1
    one = {"value": round(b1.n_structures / t1, 3), "unit": "structures/s", "cores": 1,
           "atoms_per_s": round(b1.n_atoms / t1, 1),
           "sample": f"every k-th structure of the list, k = {every1}: {b1.n_structures} structures ({b1.n_atoms} atoms), "
                     f"best of 3 runs, {t1:.2f} s wall"}
    line = {"value": round(b.n_structures / t, 3), "unit": "structures/s", "cores": threads, "kind": "port",
            "threads_used": threads, "hardware_threads": hw, "physical_cores": physical_cores(),
            "cgroup_cpu_quota": quota,
            "atoms_per_s": round(b.n_atoms / t, 1),
            "speedup_over_one_thread": round((b.n_atoms / t) / (b1.n_atoms / t1), 2),
            "sample": (f"all {b.n_structures}" if every == 1 else f"every k-th structure of the list, k = {every}: {b.n_structures} of {batch.n_structures}")
                      + f" structures ({b.n_atoms} atoms) of rank 0's workload, {n_points} points, "
                        f"oracle/sasa_oracle.c with OpenMP over structures (one per worker, per-worker scratch arenas), "
                        f"best of 3 runs, {t:.2f} s wall",
            "one_thread": one}
    return line, idx, v


def parity(batch, idx, want_atoms, got_atoms, got_res):
    """GPU results of the timed run against the oracle's on the CPU sample (structures `idx`): atoms and residues."""
    import numpy as np
    from oracle import pyoracle as po
    so = batch.structure_offsets.astype(np.int64)
    ro = batch.residue_offsets.astype(np.int64)
    n_bad = n_cmp = r_bad = r_cmp = 0
    max_abs = r_max = 0.0
    pos = 0
    for s in idx:
        b, e = int(so[s]), int(so[s + 1])
        want = want_atoms[pos:pos + (e - b)]
        pos += e - b
        got = got_atoms[b:e]
        n_cmp += e - b
        n_bad += int(np.count_nonzero(got != want))
        if e > b:
            max_abs = max(max_abs, float(np.abs(got - want).max()))
        r0, r1 = int(np.searchsorted(ro, b)), int(np.searchsorted(ro, e))
        if r1 > r0:
            want_res = po.residue_sums(want, (ro[r0:r1 + 1] - b).astype(np.uint32))
            r_cmp += r1 - r0
            r_bad += int(np.count_nonzero(got_res[r0:r1] != want_res))
            r_max = max(r_max, float(np.abs(got_res[r0:r1] - want_res).max()))
    return {"max_abs": max_abs, "n_mismatch": n_bad, "atoms_compared": n_cmp, "residue_max_abs": r_max,
            "residue_n_mismatch": r_bad, "residues_compared": r_cmp, "tolerance": 1e-4,
            "against": "oracle/sasa_oracle.c on the cpu_baseline sample"}


def kernel_source_hash():
    """sha256 over the occlusion kernels' sources: profiles/pmc_*.json carry the hash of the build they were
    measured on, and their counter-derived fields are only printed for that build."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rustsasa_amd", "csrc")
    for f in ("occlusion.hip", "occlusion_mx.inc", "occlusion_v3.inc", "occlusion_fast.inc", "device_utils.h", "device_types.h"):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load_pmc(name):
    """A profiles/pmc_*.json file, or {} when it is missing or belongs to another build of the kernels."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}, "missing"
    if pmc.get("kernel_source_sha16") != kernel_source_hash():
        return {}, f"stale: profiles/{name} was measured on kernel sources {pmc.get('kernel_source_sha16')}, " \
                   f"this build is {kernel_source_hash()} (re-run tools/profile_round.sh + tools/publish_profiles.py)"
    return pmc, "ok"


def aggregate(dist, dev, elapsed, n_structures, n_atoms):
    """Whole-job numbers from per-rank ones: MAX of the elapsed times, SUM of the units.
    `dist` is torch.distributed (RCCL on GPUs, gloo in the CPU tests) or None for one rank."""
    import torch
    units = torch.tensor([float(n_structures), float(n_atoms)], device=dev, dtype=torch.float64)
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
    return float(el.item()), float(units[0].item()), float(units[1].item())


def make_workload(workload, structures, n_points, rank, world=1, scaling="weak", shard_of=0):
    """The rank's batch of independent structures.
    weak:   every rank gets its own proteome (seed + rank);
    strong: ONE proteome (seed of rank 0), largest structures first, equal-atom shards."""
    import numpy as np
    import bench_workloads as bw
    structures = structures or bw.PROTEOME_STRUCTURES
    if workload == "proteome":
        n_points = n_points or N_POINTS
        if scaling == "strong":
            full = bw.synthetic_proteome(structures, seed=bw.PROTEOME_SEED)
            sizes = np.diff(full.structure_offsets.astype(np.int64))
            parts = bw.shard_largest_first(sizes, shard_of or world)
            batch = bw.select(full, parts[rank])
            batch.shard_indices = parts[rank]
            name = (f"synthetic AlphaFold-E.coli-like proteome, {full.n_structures} structures "
                    f"sharded largest-first over {shard_of or world} GPU(s), {n_points} points, probe {PROBE}, "
                    f"ResidueLevel" + (f"; rank 0's shard of that {shard_of}-way split, alone on one GPU" if shard_of else ""))
        else:
            batch = bw.synthetic_proteome(structures, seed=bw.PROTEOME_SEED + rank)
            name = (f"synthetic AlphaFold-E.coli-like proteome, {batch.n_structures} structures/GPU, "
                    f"{n_points} points, probe {PROBE}, ResidueLevel")
    else:
        n_points = n_points or 960
        batch = bw.synthetic_uniform(1_000_000, seed=5 + rank)
        name = f"synthetic 1M-atom structure, {n_points} points, probe {PROBE}, AtomLevel"
    return batch, n_points, name


class DeviceRun:
    """One rank's batch resident in HBM + the step that runs the whole hot path on it."""

    def __init__(self, ctx, batch, n_points, dev, with_ids, stream):
        import numpy as np
        import torch
        self.ctx, self.batch, self.n_points, self.stream = ctx, batch, n_points, stream

        def dv(a):
            return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

        self.x, self.y, self.z, self.r = dv(batch.x), dv(batch.y), dv(batch.z), dv(batch.radius)
        self.ids = dv(batch.ids.view(np.int64)) if with_ids else None
        self.res_off = dv(batch.residue_offsets.view(np.int32))
        # two sets of outputs: two batches are in flight in the timed region
        self.outs = [(torch.empty(batch.n_atoms, dtype=torch.float32, device=dev),
                      torch.empty(batch.n_residues, dtype=torch.float32, device=dev)) for _ in range(2)]
        self.out_atom, self.out_res = self.outs[0]

    def enqueue(self, counts=None, k=0):
        self.ctx.enqueue_device(self.x, self.y, self.z, self.r, self.ids, self.batch.structure_offsets,
                                self.outs[k][0], self.res_off, self.outs[k][1], counts, PROBE,
                                self.n_points, stream=self.stream)

    def step(self, counts=None):
        self.enqueue(counts)
        self.ctx.wait()


def bw_batch_with_ids(batch, ids):
    """The same batch with another id column."""
    import bench_workloads as bw
    import numpy as np
    return bw.Batch(batch.x, batch.y, batch.z, batch.radius, np.ascontiguousarray(ids, dtype=np.uint64),
                    batch.structure_offsets, batch.residue_offsets)


def timed(dist, steps, fn):
    """EXACTLY `steps` calls of fn between barrier + synchronize on both sides."""
    import torch
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    return time.perf_counter() - t0


def config5_leg(ctx, dev, steps, with_ids):
    """1M atoms in one structure, 960 points, AtomLevel (BASELINE.json configs[4]): the default stepping (two batches in
    flight), kernel times from the library's HIP events.  Timing only: parity at this size is the GPU tests' job."""
    import numpy as np
    import torch
    import bench_workloads as bw
    b = bw.synthetic_uniform(1_000_000, seed=5)
    run = DeviceRun(ctx, b, 960, dev, with_ids, None)
    run.enqueue(k=0)
    for i in range(1, 4):
        run.enqueue(k=i % 2)
        ctx.wait()
    ctx.wait()
    ctx.enable_timing(True)
    occl, grid = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.enqueue(k=0)
    for i in range(1, steps):
        run.enqueue(k=i % 2)
        ctx.wait()
        t = ctx.timings()
        occl.append(t["occlusion_ms"])
    ctx.wait()
    occl.append(ctx.timings()["occlusion_ms"])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ctx.enable_timing(False)
    for _ in range(3):  # grid build alone: one batch at a time
        ctx.enable_timing(True)
        run.step()
        grid.append(ctx.timings()["grid_build_ms"])
        ctx.enable_timing(False)
    total = float(run.outs[0][0].sum().item())
    del run
    return {"workload": "synthetic 1M-atom structure, 960 points, probe 1.4, AtomLevel (BASELINE.json configs[4])",
            "steps": steps, "ms_per_step": round(el / steps * 1e3, 4), "occlusion_ms": round(float(np.mean(occl)), 4),
            "grid_build_ms": round(float(np.min(grid)), 4), "atoms_per_s": round(b.n_atoms * steps / el, 1),
            "total_sasa": round(total, 1)}


def real_coords_leg(ctx, dev, steps, n_points):
    """The occlusion kernel on REAL, diverse coordinates: the 87 readable structures of the reference's quality set
    (tests/golden/freesasa_set.tar.xz; the reader's selection, ProtOr radii, whole complexes, the reader's hashed ids)
    tiled under rigid motions to the proteome batch's size (real_coords.py).  Device-resident stepping, two batches in
    flight, kernel times from the library's HIP events; K from the kernel's own candidate counts; the grouping figures
    (atoms per prologue, union overflow) from the CPU simulation of the kernel's rule on every fourth structure."""
    import numpy as np
    import torch
    import bench_workloads as bw
    import real_coords as rc
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sim_groups as sg
    base = rc.quality_set_batch()
    b = rc.tiled(base, 11_700_000)
    run = DeviceRun(ctx, b, n_points, dev, True, None)
    counts = torch.zeros(b.n_atoms, dtype=torch.int32, device=dev)
    run.step(counts)   # (also the warm-up: workspace growth, the id tables switched on by ids in no order)
    k = counts.cpu().numpy().astype(np.int64)
    run.step()
    run.enqueue(k=0)
    for i in range(1, 3):
        run.enqueue(k=i % 2)
        ctx.wait()
    ctx.wait()
    ctx.enable_timing(True)
    occl, n_def, cells = [], 0, 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.enqueue(k=0)
    for i in range(1, steps):
        run.enqueue(k=i % 2)
        ctx.wait()
        occl.append(ctx.timings()["occlusion_ms"])
    ctx.wait()
    t = ctx.timings()
    occl.append(t["occlusion_ms"])
    n_def, cells = int(t["n_deferred"]), int(t["n_cells"])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ctx.enable_timing(False)
    grid = []
    for _ in range(3):
        ctx.enable_timing(True)
        run.step()
        grid.append(ctx.timings()["grid_build_ms"])
        ctx.enable_timing(False)
    total = float(run.outs[0][0].sum().item())
    del run
    alg = 20.0 * b.n_atoms + 16.0 * float(k.sum())
    occ_ms = float(np.mean(occl))
    sizes = np.diff(b.structure_offsets.astype(np.int64))
    return {"workload": f"the reference's quality set (tests/quality.rs:200-258; {base.n_structures} real structures, {base.n_atoms} atoms: "
                        f"reader's selection, ProtOr radii, hashed ids, whole complexes) x {b.n_structures // base.n_structures} rigid motions, "
                        f"{n_points} points, probe {PROBE}",
            "structures": b.n_structures, "atoms": b.n_atoms, "atoms_per_structure_median": int(np.median(sizes)),
            "atoms_per_structure_max": int(sizes.max()), "steps": steps,
            "ms_per_step": round(el / steps * 1e3, 4), "structures_per_s": round(b.n_structures * steps / el, 1),
            "atoms_per_s": round(b.n_atoms * steps / el, 1),
            "occlusion_ms": round(occ_ms, 4), "grid_build_ms": round(float(np.min(grid)), 4),
            "algorithmic_bytes": alg, "frac": round(alg / (occ_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "candidates_per_atom": round(float(k.mean()), 2), "candidates_max": int(k.max()),
            "n_deferred": n_def, "grid_cells_per_atom": round(cells / b.n_atoms, 2),
            "ids_dropped_by_the_device_check": ctx.ids_dropped() > 0,
            "grouping": sg.shipped_grouping(bw.select(base, np.arange(0, base.n_structures, 4))),
            "total_sasa": round(total, 1)}


def files_leg(n_files, e2e_files=4363):
    """Directory mode (reference src/main.rs:342-480): n synthetic PDB files on /dev/shm -> per-residue values through
    the C++ host API's process_files (sasa_host_cli, its own process: parse threads + GPU batches), three calls in one
    process; then the same structures as AlphaFold-style mmCIF files (`mmcif`: the format BASELINE.json's config names).
    The files are written here (untimed)."""
    import shutil
    import tempfile
    import numpy as np
    import bench_workloads as bw
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_files as bf
    cli = os.path.join(ROOT, "rustsasa_amd", "lib", "sasa_host_cli")
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    doms = bf.load_domains()

    def one(ext):
        d = tempfile.mkdtemp(prefix="rsasa_bench_files_", dir=base)
        try:
            rng = np.random.default_rng(bw.PROTEOME_SEED)
            sizes = np.clip(rng.lognormal(np.log(2000.0), 0.75, n_files), 150, 25000).astype(int)
            paths, atoms = [], 0
            for i, n_t in enumerate(sizes):
                p = os.path.join(d, f"s{i:05d}.{ext}")
                atoms += bf.write_structure(p, int(n_t), rng, doms, cif=ext == "cif")
                paths.append(p)
            lst = os.path.join(d, "files.txt")
            open(lst, "w").write("\n".join(paths) + "\n")
            nbytes = sum(os.path.getsize(p) for p in paths)
            p = subprocess.run([cli, "files", "residue", lst, "--threads", "0", "--batch", "0", "--workers", "0",
                                "--devices", "1", "--calls", "3"], capture_output=True, text=True, timeout=600)
            if p.returncode != 0:
                return {"error": p.stderr[-300:]}
            calls = json.loads(p.stdout)["calls_s"]
            return {"files": n_files, "atoms": int(atoms), "bytes_on_disk": int(nbytes),
                    "files_per_s": round(n_files / calls[0], 1),
                    "files_per_s_later_calls": round(n_files / min(calls[1:]), 1) if len(calls) > 1 else None,
                    "calls_s": [round(c, 4) for c in calls]}
        finally:
            shutil.rmtree(d, ignore_errors=True)

    r = one("pdb")
    if "error" in r:
        return r

    def e2e(fmt):
        # files in -> one JSON file per input out, ONE call over the whole 4 363-file set (tools/bench_files.py --end-to-end:
        # a child process that writes the set with a pool of processes and runs the C++ driver; this process holds the GPU)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_files.py"), "--end-to-end", "--files", str(e2e_files),
                            "--format", fmt, "--repeat", "1"], capture_output=True, text=True, timeout=900)
        if p.returncode != 0:
            return {"error": (p.stdout + p.stderr)[-300:]}
        return json.loads(p.stdout.strip().splitlines()[-1])

    if e2e_files > 0:
        r["end_to_end"] = e2e("pdb")
        if "error" not in r["end_to_end"]:
            r["end_to_end"]["mmcif"] = e2e("cif")
    r.update({"host_threads": os.cpu_count(), "mmcif": one("cif"),
              "note": "files_per_s: the first process_files call of a fresh process (HIP start-up inside); later calls "
                      "of the same process: files_per_s_later_calls; PDB text on /dev/shm, parse + selection + GPU + "
                      "ResidueLevel results on the host; mmcif: the same structures as AlphaFold-style mmCIF files"})
    return r


def per_call_leg(batch, n_points, seconds):
    """The literal drop-in path (reference src/lib.rs:249-254 called per file from every rayon
    worker, src/main.rs:375,439): `rsasa_calculate_sasa_internal` - AoS atoms in, per-atom values out, host buffers -
    once per structure, over the proteome's size mix (every 16th structure of the list), from 1 and from 16 host
    threads with a context each.  A child process (rustsasa_amd/lib/bench_per_call): no interpreter between the calls."""
    import tempfile
    import numpy as np
    import bench_workloads as bw
    import rustsasa_amd
    exe = os.path.join(ROOT, "rustsasa_amd", "lib", "bench_per_call")
    if not os.path.exists(exe):
        return {"error": "rustsasa_amd/lib/bench_per_call not built (make -C rustsasa_amd/csrc)"}
    b = bw.select(batch, np.arange(0, batch.n_structures, 16))
    atoms = rustsasa_amd.make_atoms(b.x, b.y, b.z, b.radius, b.ids)
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    fd, path = tempfile.mkstemp(prefix="rsasa_per_call_", suffix=".bin", dir=base)
    try:
        with os.fdopen(fd, "wb") as f:
            f.write(np.uint32(b.n_structures).tobytes())
            f.write(b.structure_offsets.astype(np.uint32).tobytes())
            f.write(atoms.tobytes())
        p = subprocess.run([exe, path, str(n_points), str(seconds), "1", "16", "c16", "c64", "s64"], capture_output=True, text=True, timeout=300)
        if p.returncode != 0:
            return {"error": (p.stdout + p.stderr)[-300:]}
        legs = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    finally:
        os.unlink(path)
    sizes = np.diff(b.structure_offsets.astype(np.int64))
    return {"entry_point": "rsasa_calculate_sasa_internal (AoS rsasa_atom_t in, per-atom f32 out, pageable host buffers)",
            "structures_in_rotation": b.n_structures, "atoms_per_structure_median": int(np.median(sizes)),
            "atoms_per_structure_max": int(sizes.max()), "n_points": n_points,
            "threads_1": next((x for x in legs if x.get("threads") == 1 and x.get("mode") == "alone"), None),
            "threads_16": next((x for x in legs if x.get("threads") == 16 and x.get("mode") == "alone"), None),
            "combined": {"threads_16": next((x for x in legs if x.get("threads") == 16 and x.get("mode") == "combined"), None),
                         "threads_64": next((x for x in legs if x.get("threads") == 64 and x.get("mode") == "combined"), None),
                         "threads_64_one_shared_context": next((x for x in legs if x.get("mode") == "combined_shared_context"), None),
                         "definition": "the same calls with rsasa_context_set_call_combining(ctx, 0) on every context: calls "
                                       "that arrive together are merged into batch launches inside the library (csrc/combine.cpp)"},
            "definition": "one structure per call, one context per host thread, all on this GPU; the threads draw "
                          "structures from one shared counter for the leg's duration; every call timed (p50 / p99)"}


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.dry_run or args.dist_backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()  # n_gpus is what RCCL says, not what --gpus asked for
    if args.dry_run:
        return dry_run(args, dist, rank, world)
    if args.device is not None:
        local_rank = args.device  # (gloo runs: ranks may share a GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # the tensors the process group reduces: on the GPU for RCCL, on the host for gloo
    red_dev = torch.device("cpu") if (dist and args.dist_backend == "gloo") else dev
    scaling = args.scaling if args.workload == "proteome" else "weak"
    shard_of = args.shard_of if (world == 1 and args.workload == "proteome" and scaling == "strong") else 0
    batch, n_points, name = make_workload(args.workload, args.structures, args.n_points, rank, world,
                                          scaling, shard_of)
    # The CPU baseline (rank 0) runs FIRST, before this process binds itself to its GPU's NUMA node: its OpenMP worker
    # threads are created here, with the whole machine's affinity mask, and keep it (sched_setaffinity below changes the
    # calling thread and the threads it starts afterwards, not the ones that exist).  Its values are kept for the parity
    # diff of the timed run's output.
    cpu_base = None
    if rank == 0 and args.cpu_seconds > 0:
        affinity_cpus = len(os.sched_getaffinity(0))
        cpu_line, cmp_idx, cpu_want = cpu_baseline(batch, n_points, args.cpu_seconds)
        cpu_line["affinity_cpus"] = affinity_cpus
        cpu_line["ran"] = "before the GPU legs and before the NUMA binding of this process"
        cpu_base = (cpu_line, cmp_idx, cpu_want)
    # NUMA: before any pinned allocation and before the library starts its threads
    numa = numa_bind(gpu_pci_address(torch, local_rank))

    import rustsasa_amd
    # The host-to-host leg's pinned arrays and its first calls (which allocate the library's staging and sub-batch
    # buffers) come first (round 3 saw 6.2-6.7 instead of 5.8 ms per batch when they were allocated after the gigabytes
    # of the device-resident run; with the process bound to the GPU's NUMA node - numa_bind above - placement no
    # longer depends on the order, the order is kept for comparability).
    h2h_arrays = None
    if args.h2h_steps > 0:
        def pin(a):
            return torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()

        want_atoms = args.workload != "proteome"
        h2h_arrays = (pin(batch.x), pin(batch.y), pin(batch.z), pin(batch.radius),
                      None if args.no_ids else pin(batch.ids), pin(batch.residue_offsets),
                      pin(np.zeros(batch.n_residues, np.float32)),
                      pin(np.zeros(batch.n_atoms, np.float32)) if want_atoms else None)
        # a second set of outputs: two host batches are in flight in the stream leg
        h2h_out2 = (pin(np.zeros(batch.n_residues, np.float32)),
                    pin(np.zeros(batch.n_atoms, np.float32)) if want_atoms else None)
    ctx = rustsasa_amd.Context(local_rank)

    def h2h_step():
        hx, hy, hz, hr, hid, hro, hres, hatm = h2h_arrays
        ctx.calculate_sasa_batch(hx, hy, hz, hr, hid, batch.structure_offsets, PROBE, n_points,
                                 residue_offsets=hro, want_atoms=hatm is not None, atom_out=hatm, res_out=hres)

    def h2h_enqueue(k):
        hx, hy, hz, hr, hid, hro, hres, hatm = h2h_arrays
        res, atm = (hres, hatm) if k % 2 == 0 else h2h_out2
        ctx.host_batch_enqueue(hx, hy, hz, hr, hid, batch.structure_offsets, PROBE, n_points,
                               residue_offsets=hro, want_atoms=atm is not None, atom_out=atm, res_out=res)

    def h2h_stream(steps):
        """`steps` host batches as a rank of a sharded run works through them: batch k + 1 is enqueued before batch k
        is waited for (rsasa_host_batch_enqueue / _wait)."""
        h2h_enqueue(0)
        for k in range(1, steps):
            h2h_enqueue(k)
            ctx.host_batch_wait()
        ctx.host_batch_wait()

    if h2h_arrays:
        for _ in range(3):
            h2h_step()
        h2h_stream(4)  # (the stream's two worker contexts allocate their workspaces and staging here)
    stream = None  # the context's own launch streams (one per batch in flight); its HIP events time the kernels on them
    run = DeviceRun(ctx, batch, n_points, dev, not args.no_ids, stream)

    # candidate counts K_i (deterministic for a given input) for the algorithmic byte count
    kcount = torch.zeros(batch.n_atoms, dtype=torch.int32, device=dev)
    run.step(kcount)
    k_sum = int(kcount.to(torch.int64).sum().item())
    del kcount
    algorithmic_bytes = 20 * batch.n_atoms + 16 * k_sum  # 16 + 16*K + 4 per atom

    # The device's clocks first.  After an idle stretch - and the second workspace's allocation inside the first overlapped
    # enqueue below is one of tens of milliseconds - an occlusion launch takes up to a quarter longer and about ten launches
    # (35 ms of continuous work) to come back: 3.89, 3.69, 3.54, 3.44, 3.43, 3.23, 3.27, 3.13, 3.17, 3.07 ms and level from
    # there (tools/experiments/occlusion_durations.sh).  W warm-up steps of 3.4 ms do not cover that for small W, so the
    # stepping first runs `--prewarm-steps` batches that are neither warm-up nor timed steps (reported as `prewarm_steps`):
    # the W warm-up steps and the K timed steps then see the clocks a rank sees after its first tenth of a second.
    if args.prewarm_steps > 0 and args.warmup > 0:
        run.enqueue(k=0)
        for i in range(1, args.prewarm_steps):
            run.enqueue(k=i % 2)
            ctx.wait()
        ctx.wait()

    # warm-up in the timed region's stepping
    if args.warmup > 0:
        run.enqueue(k=0)
        for i in range(1, args.warmup):
            run.enqueue(k=i % 2)
            ctx.wait()
        ctx.wait()

    # ---- the timed region: HBM-resident inputs and outputs, two batches in flight ----
    # Step k + 1 is enqueued before step k is waited for (rsasa_batch_wait returns the OLDEST batch); the
    # context runs them in two workspaces on two streams, so the small kernels at the start of a batch and
    # the thin tail of its occlusion kernel overlap with the neighbour's.  Exactly `steps` batches run
    # between the two barriers.  Kernel times: the library's HIP events on each batch's launch stream.
    ctx.enable_timing(True)
    occl_ms, grid_ms, agg_ms, cells = [], [], [], [0]

    def collect():
        t = ctx.timings()
        cells[0] = int(t["n_cells"])
        occl_ms.append(t["occlusion_ms"])
        grid_ms.append(t["grid_build_ms"])
        agg_ms.append(t["aggregate_ms"])

    step_done = []  # wall clock when each step's results were complete

    def timed_region():
        run.enqueue(k=0)
        for i in range(1, args.steps):
            run.enqueue(k=i % 2)
            ctx.wait()
            step_done.append(time.perf_counter())
            collect()
        ctx.wait()
        step_done.append(time.perf_counter())
        collect()

    elapsed = timed(dist, 1, timed_region)
    ctx.enable_timing(False)
    elapsed, total_structures, total_atoms = aggregate(dist, red_dev, elapsed, batch.n_structures,
                                                       batch.n_atoms)
    got_atoms = run.outs[(args.steps - 1) % 2][0].cpu().numpy()
    got_res = run.outs[(args.steps - 1) % 2][1].cpu().numpy()
    outputs_equal = args.steps < 2 or bool(torch.equal(run.outs[0][1], run.outs[1][1]) and
                                             torch.equal(run.outs[0][0], run.outs[1][0]))

    # ---- every rank's shard against the oracle (tests: the N > 1 path with the real engine on every rank) ----
    shard_parity = None
    if args.verify_shards:
        from oracle import pyoracle as po
        want = po.calculate_sasa_batch(batch.x, batch.y, batch.z, batch.radius, batch.ids, batch.structure_offsets,
                                       PROBE, n_points, 8, threads=0)
        want_res = po.residue_sums(want, batch.residue_offsets)
        mine = {"rank": rank, "device": local_rank, "structures": batch.n_structures, "atoms": batch.n_atoms,
                "shard_indices": [int(i) for i in getattr(batch, "shard_indices", np.arange(batch.n_structures))],
                "atoms_equal_oracle": bool(np.array_equal(got_atoms, want)),
                "residues_equal_oracle": bool(np.array_equal(got_res, want_res)),
                "total_sasa": float(np.sum(got_atoms, dtype=np.float64))}
        shard_parity = [None] * world
        if dist:
            dist.all_gather_object(shard_parity, mine)
        else:
            shard_parity = [mine]

    # ---- SURVEY 8d's definition: pinned host SoA in, per-residue values back on the host ----
    h2h = None
    if args.h2h_steps > 0:
        hres = h2h_arrays[6]
        for _ in range(max(3, 2 * args.h2h_steps)):  # (untimed: the rate of host calls settles over their first twenty or so)
            h2h_step()
        s_el = timed(dist, args.h2h_steps, h2h_step)   # one synchronous call after the other
        s_el, _, _ = aggregate(dist, red_dev, s_el, batch.n_structures, batch.n_atoms)
        sync_equal = bool(np.array_equal(hres, got_res))
        hres[:] = 0.0
        h2h_out2[0][:] = 0.0
        n_stream = max(2, args.h2h_steps)
        h2h_stream(n_stream)  # (untimed: the stream settles over its first ten or so batches)
        # (three repetitions, the median reported and all three listed: the stream's rate depends on how the two workers'
        # uploads fall against each other's kernels, and single runs of ten batches scatter by several percent)
        h_runs = []
        for _ in range(3):
            h_el = timed(dist, 1, lambda: h2h_stream(n_stream))
            h_el, h_structs, _ = aggregate(dist, red_dev, h_el, batch.n_structures, batch.n_atoms)
            h_runs.append(h_el)
        h_el = sorted(h_runs)[1]
        stream_leg = {"ms_per_step": round(h_el / n_stream * 1e3, 4), "value": round(h_structs * n_stream / h_el, 2),
                  "steps": n_stream, "ms_per_step_runs": [round(t / n_stream * 1e3, 4) for t in h_runs],
                  "definition": "a STREAM of host batches: batch k + 1 enqueued before batch k is waited for "
                                "(rsasa_host_batch_enqueue / _wait: two worker contexts, the calls taking turns on the link)"}
        one = {"ms_per_step": round(s_el / args.h2h_steps * 1e3, 4), "value": round(h_structs * args.h2h_steps / s_el, 2),
               "steps": args.h2h_steps,
               "definition": "rsasa_calculate_sasa_batch, each call waited for before the next"}
        best = stream_leg if stream_leg["value"] >= one["value"] else one
        h2h = {"value": best["value"], "unit": "structures/s", "ms_per_step": best["ms_per_step"], "steps": best["steps"],
               "mode": "stream" if best is stream_leg else "one_call_at_a_time",
               "definition": "SURVEY 8d: pre-parsed SoA in pinned host memory -> per-residue values in pinned host "
                             "memory (H2D, all kernels, D2H).  Both ways of calling are timed - a stream of host batches "
                             "(batch k + 1 enqueued before batch k is waited for: two worker contexts with hardware "
                             "queues of their own, two sub-batches per call, the uploads taking the link in batch order) "
                             "and one call after the other (a call's sub-batches pipelined over a copy-in, a compute and "
                             "a copy-out stream) - and the faster one on this box is the value (`mode`).  The stream is "
                             "bound by the link (DESIGN.md 6)",
               "residues_equal_hbm_run": bool(sync_equal and np.array_equal(hres, got_res)
                                              and np.array_equal(h2h_out2[0], got_res)),
               "stream": stream_leg, "one_call_at_a_time": one}
        if h2h_arrays[4] is not None and rank == 0 and world == 1:
            # the same stream with the ids as SASAOptions::process passes them (options.rs:183: 64-bit hashes, in no order): the
            # host cannot prove hashes different with one comparison per atom, so 32-bit folds cross the link and the id
            # rule stays in the kernels
            hid_sorted = h2h_arrays[4].copy()
            h2h_arrays[4][:] = hid_sorted * np.uint64(0x9E3779B97F4A7C15)
            try:
                n0 = ctx.ids_dropped()
                h2h_stream(n_stream)
                hr = []
                for _ in range(3):
                    hr.append(timed(dist, 1, lambda: h2h_stream(n_stream)))
                hm = sorted(hr)[1]
                h2h["ids_as_hashes"] = {
                    "ms_per_step": round(hm / n_stream * 1e3, 4), "value": round(batch.n_structures * n_stream / hm, 2),
                    "steps": n_stream, "ms_per_step_runs": [round(t / n_stream * 1e3, 4) for t in hr],
                    "residues_equal_hbm_run": bool(np.array_equal(hres, got_res) and np.array_equal(h2h_out2[0], got_res)),
                    "sub_batches_run_without_ids": ctx.ids_dropped() - n0,
                    "definition": "the stream of host batches with the id column replaced by 64-bit hashes in no order (pinned): "
                                  "what SASAOptions::process and process_files hand to the hot path"}
            finally:
                h2h_arrays[4][:] = hid_sorted

    # ---- secondary: one batch at a time (enqueue, wait, enqueue, ...): what a caller with a single batch sees ----
    two = None
    if args.two_steps > 0:
        ctx.enable_timing(True)
        seq_occl, seq_grid = [], []

        def seq_step():
            run.step()
            t = ctx.timings()
            seq_occl.append(t["occlusion_ms"])
            seq_grid.append(t["grid_build_ms"])

        t_el = timed(dist, args.two_steps, seq_step)
        ctx.enable_timing(False)
        t_el, t_structs, _ = aggregate(dist, red_dev, t_el, batch.n_structures, batch.n_atoms)
        two = {"value": round(t_structs * args.two_steps / t_el, 2), "unit": "structures/s",
               "ms_per_step": round(t_el / args.two_steps * 1e3, 4), "steps": args.two_steps,
               "occlusion_kernel_ms": round(float(np.mean(seq_occl)), 4),
               "grid_build_kernel_ms": round(float(np.mean(seq_grid)), 4),
               "definition": "the same steps one batch at a time: each step is waited for before the next is enqueued"}

    # ---- secondary (rank 0, one GPU): the same batch with ids in no order, as SASAOptions::process makes them ----
    hashed = None
    if rank == 0 and world == 1 and not args.no_ids and args.hashed_ids_steps > 0:
        hb = bw_batch_with_ids(batch, batch.ids * np.uint64(0x9E3779B97F4A7C15))  # (odd multiplier: a bijection, still all different)
        hrun = DeviceRun(ctx, hb, n_points, dev, True, stream)
        n0 = ctx.ids_dropped()
        hrun.enqueue(k=0)
        for i in range(1, 4):
            hrun.enqueue(k=i % 2)
            ctx.wait()
        ctx.wait()
        ctx.enable_timing(True)
        h_occl = []

        def hashed_region():
            hrun.enqueue(k=0)
            for i in range(1, args.hashed_ids_steps):
                hrun.enqueue(k=i % 2)
                ctx.wait()
                h_occl.append(ctx.timings()["occlusion_ms"])
            ctx.wait()
            h_occl.append(ctx.timings()["occlusion_ms"])

        hh_el = timed(dist, 1, hashed_region)
        ctx.enable_timing(False)
        hashed = {"value": round(batch.n_structures * args.hashed_ids_steps / hh_el, 2), "unit": "structures/s",
                  "ms_per_step": round(hh_el / args.hashed_ids_steps * 1e3, 4), "steps": args.hashed_ids_steps,
                  "occlusion_kernel_ms": round(float(np.mean(h_occl)), 4),
                  "ids_dropped_batches": ctx.ids_dropped() - n0,
                  "atoms_equal_main_run": bool(np.array_equal(hrun.outs[0][0].cpu().numpy(), got_atoms)),
                  "definition": "the timed region's stepping with the ids replaced by 64-bit hashes (all different, in no "
                                "order: what SASAOptions::process passes, options.rs:183): one comparison per atom does not "
                                "prove these different; a hash table per structure in LDS does (k_ids_distinct, 0.08 ms per "
                                "batch), and the batches run without ids as well"}
        del hrun

    # ---- secondary: weak scaling (every rank its own proteome) ----
    weak = None
    if world > 1 and scaling == "strong" and args.weak_steps > 0:
        del run
        wb, _, _ = make_workload("proteome", args.structures, n_points, rank, world, "weak")
        wrun = DeviceRun(ctx, wb, n_points, dev, not args.no_ids, stream)
        for _ in range(2):
            wrun.step()
        w_el = timed(dist, args.weak_steps, wrun.step)
        w_el, w_structs, _ = aggregate(dist, red_dev, w_el, wb.n_structures, wb.n_atoms)
        weak = {"value": round(w_structs * args.weak_steps / w_el, 2), "unit": "structures/s",
                "ms_per_step": round(w_el / args.weak_steps * 1e3, 4), "steps": args.weak_steps,
                "structures_per_gpu": wb.n_structures}
        del wrun

    # ---- secondary (rank 0, one GPU): BASELINE.json configs[4], 1M atoms in one structure x 960 points, timing only ----
    config5 = None
    if rank == 0 and world == 1 and args.workload == "proteome" and args.config5_steps > 0 and not shard_of:
        config5 = config5_leg(ctx, dev, args.config5_steps, not args.no_ids)
    # ---- secondary (rank 0, one GPU): the same kernels on real, diverse coordinates (the reference's quality set, tiled) ----
    real = None
    if rank == 0 and world == 1 and args.workload == "proteome" and args.real_steps > 0 and not shard_of:
        try:
            real = real_coords_leg(ctx, dev, args.real_steps, n_points)
        except Exception as e:  # noqa: BLE001 (a leg of its own: the line is printed without it)
            real = {"error": repr(e)[:300]}
    # ---- secondary (rank 0, one GPU): directory mode, files on disk -> per-residue values (C++ process_files) ----
    files_mode = None
    if rank == 0 and world == 1 and args.workload == "proteome" and args.files > 0 and not shard_of:
        files_mode = files_leg(args.files, args.e2e_files)

    # ---- secondary (rank 0, one GPU): the drop-in call per structure, from 1 and 16 host threads ----
    per_call = None
    if rank == 0 and world == 1 and args.workload == "proteome" and args.per_call_seconds > 0 and not shard_of:
        per_call = per_call_leg(batch, n_points, args.per_call_seconds)

    if config5:
        pmcu, _ = load_pmc("pmc_uniform1m.json")
        if pmcu.get("alu_busy"):
            config5["alu_busy"] = pmcu["alu_busy"]  # (counters of a profiled single launch of this build: profiles/)
    if rank == 0:
        occl = float(np.mean(occl_ms))
        pmc, pmc_state = load_pmc("pmc_occlusion.json")
        full_batch = world == 1 and batch.n_structures == 4363
        if args.workload == "proteome":
            achieved = algorithmic_bytes / (occl * 1e-3) / 1e9
            roofline = {"bound": "valu", "kernel": "k_occlusion", "achieved": round(achieved, 2),
                        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                        "frac_of": "SURVEY 8d algorithmic bytes (20 N + 16 sum K) / kernel time / 8 TB/s: "
                                   "the north_star figure; the kernel itself is bound by the vector + matrix ALU "
                                   "(alu_busy), its HBM-side traffic is `traffic`",
                        "traffic": pmc.get("hbm_bytes_per_launch") if full_batch else None,
                        "traffic_source": PMC_FILE + " (rocprofv3 --pmc passes of one launch of this workload, "
                                                     "see profiles/README.md): " + (pmc_state if full_batch else
                                                                                    "not this workload"),
                        "algorithmic_bytes_per_launch": algorithmic_bytes,
                        "kernel_ms": round(occl, 4)}
            if pmc.get("alu_busy") and full_batch:
                # counters of ONE profiled launch of this build (rocprofv3 --pmc, --steps 1: no neighbouring batch), with
                # that launch's own cycle count: a modelled occupancy, both variants printed (profiles/README.md)
                roofline["alu_busy"] = pmc["alu_busy"]
                roofline["insts_per_launch"] = {"vector": pmc.get("valu_insts_per_launch"), "scalar": pmc.get("salu_insts_per_launch")}
        else:
            # config 5 is bound by the vector ALUs: no HBM fraction is claimed.  Utilisation =
            # vector instructions the occlusion launch executed (rocprofv3 SQ_INSTS_VALU, profiles/) x 64
            # lanes x 2 flop, as if every one were a full-wave FMA, against the f32 vector peak.
            pmcu, _ = load_pmc("pmc_uniform1m.json")
            vi = pmcu.get("valu_insts_per_launch") if (world == 1 and batch.n_atoms == 1_000_000 and n_points == 960) else None
            achieved = vi * 128.0 / (occl * 1e-3) / 1e12 if vi else None
            roofline = {"bound": "valu", "kernel": "k_occlusion",
                        "achieved": round(achieved, 3) if achieved else None,
                        "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / VALU_PEAK_TFLOPS, 4) if achieved else None,
                        "frac_of": "vector-ALU utilisation: executed vector instructions (profiles/pmc_uniform1m.json) "
                                   "x 64 lanes x 2 flop / kernel time / f32 vector peak; matrix instructions count "
                                   "as one instruction each although they hold the pipe for 8-32 cycles",
                        "valu_insts_per_launch": vi, "mfma_insts_per_launch": pmcu.get("mfma_insts_per_launch") if vi else None,
                        "point_tests_upper_bound_per_launch": int(n_points) * int(k_sum),
                        "traffic": None, "kernel_ms": round(occl, 4)}
        line = {
            "metric": "structures/sec on AF2 E. coli proteome (100 pts, 1.4 A probe)",
            "value": round(total_structures * args.steps / elapsed, 2),
            "unit": "structures/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_steps": args.prewarm_steps if args.warmup > 0 else 0,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": name, "inputs": "resident in HBM (SoA + offsets), outputs left in HBM",
                       "stepping": "two batches in flight (step k + 1 enqueued before step k is waited for)",
                       "outputs_of_both_workspaces_equal": outputs_equal,
                       "structures_total": int(total_structures), "atoms_total": int(total_atoms),
                       "structures_rank0": batch.n_structures, "atoms_rank0": batch.n_atoms,
                       "residues_rank0": batch.n_residues,
                       "candidates_per_atom": round(k_sum / max(batch.n_atoms, 1), 2),
                       "grid_cells_rank0": cells[0],
                       "atoms_per_s": round(total_atoms * args.steps / elapsed, 1),
                       "ids": not args.no_ids,
                       "ids_note": "one 64-bit id per atom is passed, as the reference's Atom carries one; the engine checks "
                                   "whether the ids of every structure are all different (these rise within every "
                                   "structure, like serials and indices: one comparison per atom in k_bounds or in the "
                                   "host's coding threads; ids in no order: `ids_as_hashes`) and runs such batches "
                                   "without them - same values (parity below is against the oracle WITH the ids); "
                                   "rsasa_context_ids_dropped counts the (sub-)batches",
                       "ids_dropped_batches": ctx.ids_dropped(),
                       "numa_node": numa["numa_node"], "cpus": numa["cpus"], "cpu_list": numa["cpu_list"],
                       "gpu_pci": numa["gpu_pci"],
                       "parallelism": f"{world} rank(s), one per GPU, independent shards, no data-path "
                                      f"collective"},
            "roofline": roofline,
            "kernel_ms": {"grid_build": round(float(np.mean(grid_ms)), 4),
                          "occlusion": round(occl, 4),
                          "residue_sums": round(float(np.mean(agg_ms)), 4),
                          "note": "HIP events on each batch's launch stream over the timed region; a batch's grid build "
                                  "is enqueued beside the previous batch's occlusion kernel and mostly waits for it (its "
                                  "wall time here includes that wait: alone it takes one_at_a_time.grid_build_kernel_ms), "
                                  "occlusion kernels run one after the other"},
        }
        if len(step_done) > 2:
            d = np.diff(np.array(step_done)) * 1e3  # (the first step also carries the second one's enqueue: left out)
            line["ms_per_step_min"] = round(float(d.min()), 4)
            line["ms_per_step_median"] = round(float(np.median(d)), 4)
            line["ms_per_step_max"] = round(float(d.max()), 4)
        if h2h:
            line["value_host_to_host"] = h2h["value"]  # SURVEY 8d's definition (host SoA in, residue values out)
            line["host_to_host"] = h2h
        if config5:
            line["config5"] = config5
        if real:
            line["real_coords"] = real
        if files_mode:
            line["files_mode"] = files_mode
        if per_call:
            line["per_call"] = per_call
        if two:
            line["one_at_a_time"] = two
        if hashed:
            line["ids_as_hashes"] = hashed
        if shard_of:
            line["config"]["shard_of"] = shard_of
        if shard_parity:
            all_idx = sorted(i for sp in shard_parity for i in sp["shard_indices"])
            line["shard_parity"] = [{k: v for k, v in sp.items() if k != "shard_indices"} for sp in shard_parity]
            line["shards_disjoint_and_complete"] = (all_idx == list(range(int(total_structures)))
                                                    if scaling == "strong" else None)
            line["config"]["dist_backend"] = args.dist_backend if dist else None
        if weak:
            line["weak_scaling"] = weak
        if cpu_base:  # (rank 0, on its own shard at N > 1)
            line["cpu_baseline"], cmp_idx, want = cpu_base
            line["parity"] = parity(batch, cmp_idx, want, got_atoms, got_res)
        print(json.dumps(line, default=lambda o: o.item() if hasattr(o, "item") else str(o)), flush=True)

    ctx.close()
    if dist:
        dist.destroy_process_group()


def dry_run(args, dist, rank, world):
    """No GPU: the launcher, the sharder and the aggregation on gloo ranks (tests/test_distributed_cpu.py)."""
    import numpy as np
    import torch
    dev = torch.device("cpu")
    batch, n_points, name = make_workload("proteome", args.structures or 64, args.n_points, rank, world,
                                          args.scaling)
    # the NUMA binding of a real run, against a sysfs tree and PCI addresses the test provides
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    bdfs = [b for b in os.environ.get("RSASA_DRYRUN_GPU_PCI", "").split(",") if b]
    numa = numa_bind(bdfs[local_rank] if local_rank < len(bdfs) else None, os.environ.get("RSASA_DRYRUN_SYSFS", "/sys"))
    numa["affinity_after"] = sorted(os.sched_getaffinity(0))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    elapsed = time.perf_counter() - t0 + 1e-3 * (rank + 1)
    elapsed, total_structures, total_atoms = aggregate(dist, dev, elapsed, batch.n_structures, batch.n_atoms)
    shards = [None] * world
    mine = (rank, [int(i) for i in getattr(batch, "shard_indices", np.arange(batch.n_structures))],
            batch.n_atoms, float(batch.x.sum(dtype=np.float64)), numa)
    if dist:
        dist.all_gather_object(shards, mine)
    else:
        shards = [mine]
    if rank == 0:
        all_idx = sorted(i for s in shards for i in s[1])
        n_total = args.structures or 64
        complete = all_idx == list(range(n_total)) if args.scaling == "strong" else None
        print(json.dumps({"metric": "structures/sec on AF2 E. coli proteome (100 pts, 1.4 A probe)",
                          "dry_run": True, "value": round(total_structures * args.steps / elapsed, 2),
                          "unit": "structures/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "scaling": args.scaling,
                          "config": {"workload": name, "structures_total": int(total_structures),
                                     "atoms_total": int(total_atoms)},
                          "shards_disjoint_and_complete": complete,
                          "numa": [s[4] for s in shards],
                          "shard_atoms": [s[2] for s in shards],
                          "shard_structures": [len(s[1]) for s in shards]}), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
