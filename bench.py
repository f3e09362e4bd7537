#!/usr/bin/env python3
"""Headline benchmark: structures/sec of the Shrake-Rupley hot path on MI355X.

One "step" = one pass of the whole hot path (bounds -> cell grid -> counting
sort -> occlusion -> ResidueLevel sums) over the rank's structures.  The
workload is BASELINE.json configs[2] / configs[3]: an AlphaFold-E.-coli-like
proteome (4 363 structures, 100 sphere points, probe 1.4 A, ResidueLevel),
synthesised offline-reproducibly by bench_workloads.synthetic_proteome.

`value` is measured with the inputs and outputs resident in HBM (the contract's
definition); the SURVEY 8d host-to-host rate (pinned SoA in host memory in,
per-residue values back in host memory) is timed in the same run and printed as
`host_to_host`.

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
VALU_PEAK_INSTS = 1024 * 2.4e9 / 2.0   # wave64 VALU instructions/s: 1024 SIMD-32s, 2 cycles each, 2.4 GHz
VALU_PEAK_TFLOPS = 157.3        # f32 vector peak (MI355X_MICROARCH.md)
PROBE = 1.4
N_POINTS = 100
PMC_FILE = os.path.join("profiles", "pmc_occlusion.json")


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=5)
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
    p.add_argument("--two-steps", type=int, default=40,
                   help="steps of the secondary two-batches-in-flight measurement (0 = skip)")
    p.add_argument("--weak-steps", type=int, default=5,
                   help="timed steps of the secondary weak-scaling measurement at N > 1 (0 disables it)")
    p.add_argument("--no-ids", action="store_true", help="pass id = NULL (all atoms distinct)")
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


def cpu_baseline(batch, n_points, target_seconds):
    """The oracle (a port of the reference's CPU path) on a bounded sample of the same workload.
    Returns the bench-line object and the oracle's per-atom values of the sample."""
    from oracle import pyoracle as po
    threads = min(po.max_threads(), os.cpu_count() or 1)

    def run(n_struct):
        e = int(batch.structure_offsets[n_struct])
        t0 = time.perf_counter()
        v = po.calculate_sasa_batch(batch.x[:e], batch.y[:e], batch.z[:e], batch.radius[:e],
                                    batch.ids[:e], batch.structure_offsets[:n_struct + 1], PROBE,
                                    n_points, 8, threads=threads)
        return time.perf_counter() - t0, v

    probe_n = min(batch.n_structures, max(threads * 4, 16))
    run(min(probe_n, 8))  # warm up the thread pool and page in the library
    t_probe, v = run(probe_n)
    n = int(min(batch.n_structures, max(probe_n, probe_n * target_seconds / max(t_probe, 1e-6))))
    t = t_probe
    if n > probe_n:
        t, v = run(n)
    else:
        n = probe_n
    atoms = int(batch.structure_offsets[n])
    line = {"value": round(n / t, 3), "unit": "structures/s", "cores": threads, "kind": "port",
            "sample": f"first {n} of {batch.n_structures} structures ({atoms} atoms) of rank 0's "
                      f"workload, {n_points} points, oracle/sasa_oracle.c with OpenMP over "
                      f"structures, {t:.1f} s wall"}
    return line, v, n


def parity(batch, n_struct, want_atoms, got_atoms, got_res):
    """GPU results of the timed run against the oracle's on the CPU sample: atoms and residues."""
    import numpy as np
    from oracle import pyoracle as po
    e = int(batch.structure_offsets[n_struct])
    ga = got_atoms[:e]
    n_res = int(np.searchsorted(batch.residue_offsets, e, side="right") - 1)
    want_res = po.residue_sums(want_atoms, batch.residue_offsets[:n_res + 1])
    d = np.abs(ga - want_atoms)
    dr = np.abs(got_res[:n_res] - want_res)
    return {"max_abs": float(d.max()) if e else 0.0, "n_mismatch": int(np.count_nonzero(ga != want_atoms)),
            "atoms_compared": e, "residue_max_abs": float(dr.max()) if n_res else 0.0,
            "residue_n_mismatch": int(np.count_nonzero(got_res[:n_res] != want_res)),
            "residues_compared": n_res, "tolerance": 1e-4,
            "against": "oracle/sasa_oracle.c on the cpu_baseline sample"}


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


def make_workload(workload, structures, n_points, rank, world=1, scaling="weak"):
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
            parts = bw.shard_largest_first(sizes, world)
            batch = bw.select(full, parts[rank])
            batch.shard_indices = parts[rank]
            name = (f"synthetic AlphaFold-E.coli-like proteome, {full.n_structures} structures "
                    f"sharded largest-first over {world} GPU(s), {n_points} points, probe {PROBE}, "
                    f"ResidueLevel")
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
        self.out_atom = torch.empty(batch.n_atoms, dtype=torch.float32, device=dev)
        self.out_res = torch.empty(batch.n_residues, dtype=torch.float32, device=dev)

    def enqueue(self, counts=None):
        self.ctx.enqueue_device(self.x, self.y, self.z, self.r, self.ids, self.batch.structure_offsets,
                                self.out_atom, self.res_off, self.out_res, counts, PROBE,
                                self.n_points, stream=self.stream)

    def step(self, counts=None):
        self.enqueue(counts)
        self.ctx.wait()

    def twin(self, ctx, stream):
        """The same resident inputs behind another context (own workspace, stream and outputs)."""
        import copy
        import torch
        t = copy.copy(self)
        t.ctx, t.stream = ctx, stream
        t.out_atom, t.out_res = torch.empty_like(self.out_atom), torch.empty_like(self.out_res)
        return t


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
        if args.dry_run:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()  # n_gpus is what RCCL says, not what --gpus asked for
    if args.dry_run:
        return dry_run(args, dist, rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import rustsasa_amd
    scaling = args.scaling if args.workload == "proteome" else "weak"
    batch, n_points, name = make_workload(args.workload, args.structures, args.n_points, rank, world,
                                          scaling)
    ctx = rustsasa_amd.Context(local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    run = DeviceRun(ctx, batch, n_points, dev, not args.no_ids, stream)

    # candidate counts K_i (deterministic for a given input) for the algorithmic byte count
    kcount = torch.zeros(batch.n_atoms, dtype=torch.int32, device=dev)
    run.step(kcount)
    k_sum = int(kcount.to(torch.int64).sum().item())
    del kcount
    algorithmic_bytes = 20 * batch.n_atoms + 16 * k_sum  # 16 + 16*K + 4 per atom

    for _ in range(args.warmup):
        run.step()

    # ---- the timed region: HBM-resident inputs and outputs ----
    ctx.enable_timing(True)
    occl_ms, grid_ms, agg_ms, cells = [], [], [], [0]

    def timed_step():
        run.step()
        t = ctx.timings()
        cells[0] = int(t["n_cells"])
        occl_ms.append(t["occlusion_ms"])
        grid_ms.append(t["grid_build_ms"])
        agg_ms.append(t["aggregate_ms"])

    elapsed = timed(dist, args.steps, timed_step)
    ctx.enable_timing(False)
    elapsed, total_structures, total_atoms = aggregate(dist, dev, elapsed, batch.n_structures,
                                                       batch.n_atoms)
    got_atoms = run.out_atom.cpu().numpy()
    got_res = run.out_res.cpu().numpy()

    # ---- SURVEY 8d's definition: pinned host SoA in, per-residue values back on the host ----
    h2h = None
    if args.h2h_steps > 0:
        def pin(a):
            return torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()

        hx, hy, hz, hr = pin(batch.x), pin(batch.y), pin(batch.z), pin(batch.radius)
        hid = None if args.no_ids else pin(batch.ids)
        hro = pin(batch.residue_offsets)
        hres = pin(np.zeros(batch.n_residues, np.float32))
        want_atoms = args.workload != "proteome"
        hatm = pin(np.zeros(batch.n_atoms, np.float32)) if want_atoms else None

        def h2h_step():
            ctx.calculate_sasa_batch(hx, hy, hz, hr, hid, batch.structure_offsets, PROBE, n_points,
                                     residue_offsets=hro, want_atoms=want_atoms, atom_out=hatm,
                                     res_out=hres)

        for _ in range(5):  # (the first calls allocate the sub-batch slots and may regrow the cell array)
            h2h_step()
        h_el = timed(dist, args.h2h_steps, h2h_step)
        h_el, h_structs, _ = aggregate(dist, dev, h_el, batch.n_structures, batch.n_atoms)
        h2h = {"value": round(h_structs * args.h2h_steps / h_el, 2), "unit": "structures/s",
               "ms_per_step": round(h_el / args.h2h_steps * 1e3, 4), "steps": args.h2h_steps,
               "definition": "SURVEY 8d: pre-parsed SoA in pinned host memory -> per-residue values "
                             "in pinned host memory (H2D, all kernels, D2H), sub-batches pipelined "
                             "over a copy-in, a compute and a copy-out stream",
               "residues_equal_hbm_run": bool(np.array_equal(hres, got_res))}

    # ---- secondary: two batches in flight (a second context: own workspace and stream), batch k + 1
    # enqueued before batch k is waited for.  The host's enqueue time and the small kernels of one
    # batch then hide behind the other's occlusion kernel; `value` stays the one-batch-at-a-time rate
    # whose kernel times the roofline is computed from.
    two = None
    if args.two_steps > 0:
        ctx2 = rustsasa_amd.Context(local_rank)
        s2 = torch.cuda.Stream()
        runs = [run, run.twin(ctx2, s2.cuda_stream)]
        for r_ in runs:
            for _ in range(3):
                r_.step()

        def two_region():
            k_steps = args.two_steps
            runs[0].enqueue()
            for i in range(1, k_steps):
                runs[i % 2].enqueue()
                runs[(i - 1) % 2].ctx.wait()
            runs[(k_steps - 1) % 2].ctx.wait()

        t_el = timed(dist, 1, two_region)
        t_el, t_structs, _ = aggregate(dist, dev, t_el, batch.n_structures, batch.n_atoms)
        same = bool(torch.equal(runs[0].out_res, runs[1].out_res))
        two = {"value": round(t_structs * args.two_steps / t_el, 2), "unit": "structures/s",
               "ms_per_step": round(t_el / args.two_steps * 1e3, 4), "steps": args.two_steps,
               "definition": "the same steps with two batches in flight: two contexts (workspaces, streams) "
                             "alternate, batch k + 1 is enqueued before batch k is waited for",
               "outputs_equal": same}
        del runs
        ctx2.close()

    # ---- secondary: weak scaling (every rank its own proteome) ----
    weak = None
    if world > 1 and scaling == "strong" and args.weak_steps > 0:
        del run
        wb, _, _ = make_workload("proteome", args.structures, n_points, rank, world, "weak")
        wrun = DeviceRun(ctx, wb, n_points, dev, not args.no_ids, stream)
        for _ in range(2):
            wrun.step()
        w_el = timed(dist, args.weak_steps, wrun.step)
        w_el, w_structs, _ = aggregate(dist, dev, w_el, wb.n_structures, wb.n_atoms)
        weak = {"value": round(w_structs * args.weak_steps / w_el, 2), "unit": "structures/s",
                "ms_per_step": round(w_el / args.weak_steps * 1e3, 4), "steps": args.weak_steps,
                "structures_per_gpu": wb.n_structures}
        del wrun

    if rank == 0:
        occl = float(np.mean(occl_ms))
        pmc = {}
        try:
            pmc = json.load(open(os.path.join(ROOT, PMC_FILE)))
        except Exception:
            pmc = {}
        if args.workload == "proteome":
            achieved = algorithmic_bytes / (occl * 1e-3) / 1e9
            roofline = {"bound": "valu", "kernel": "k_occlusion", "achieved": round(achieved, 2),
                        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                        "frac_of": "SURVEY 8d algorithmic bytes (20 N + 16 sum K) / kernel time / 8 TB/s: "
                                   "the north_star figure; the kernel itself is bound by vector-instruction "
                                   "issue (valu_issue), its HBM-side traffic is `traffic`",
                        "traffic": pmc.get("hbm_bytes_per_launch") if world == 1 and
                        batch.n_structures == 4363 else None,
                        "traffic_source": PMC_FILE + " (rocprofv3 --pmc passes of this workload, "
                                                     "see profiles/README.md)",
                        "algorithmic_bytes_per_launch": algorithmic_bytes,
                        "kernel_ms": round(occl, 4)}
            vi = pmc.get("valu_insts_per_launch")
            if vi and world == 1 and batch.n_structures == 4363:
                rate = vi / (occl * 1e-3)
                roofline["valu_issue"] = {"insts_per_launch": vi, "achieved_insts_per_s": round(rate, 1),
                                          "peak_insts_per_s": VALU_PEAK_INSTS,
                                          "frac": round(rate / VALU_PEAK_INSTS, 4), "source": PMC_FILE}
        else:
            # config 5 is bound by the vector ALUs: no HBM fraction is claimed.  Utilisation =
            # vector instructions the occlusion launch executed (rocprofv3 SQ_INSTS_VALU, profiles/) x 64
            # lanes x 2 flop, as if every one were a full-wave FMA, against the f32 vector peak.
            pmcu = {}
            try:
                pmcu = json.load(open(os.path.join(ROOT, "profiles", "pmc_uniform1m.json")))
            except Exception:
                pmcu = {}
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
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": name, "inputs": "resident in HBM (SoA + offsets), outputs left in HBM",
                       "structures_total": int(total_structures), "atoms_total": int(total_atoms),
                       "structures_rank0": batch.n_structures, "atoms_rank0": batch.n_atoms,
                       "residues_rank0": batch.n_residues,
                       "candidates_per_atom": round(k_sum / max(batch.n_atoms, 1), 2),
                       "grid_cells_rank0": cells[0],
                       "atoms_per_s": round(total_atoms * args.steps / elapsed, 1),
                       "ids": not args.no_ids,
                       "parallelism": f"{world} rank(s), one per GPU, independent shards, no data-path "
                                      f"collective"},
            "roofline": roofline,
            "kernel_ms": {"grid_build": round(float(np.mean(grid_ms)), 4),
                          "occlusion": round(occl, 4),
                          "residue_sums": round(float(np.mean(agg_ms)), 4)},
        }
        if h2h:
            line["host_to_host"] = h2h
        if two:
            line["two_in_flight"] = two
        if weak:
            line["weak_scaling"] = weak
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"], want, n_cmp = cpu_baseline(batch, n_points, args.cpu_seconds)
            line["parity"] = parity(batch, n_cmp, want, got_atoms, got_res)
        print(json.dumps(line), flush=True)

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
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    elapsed = time.perf_counter() - t0 + 1e-3 * (rank + 1)
    elapsed, total_structures, total_atoms = aggregate(dist, dev, elapsed, batch.n_structures, batch.n_atoms)
    shards = [None] * world
    mine = (rank, [int(i) for i in getattr(batch, "shard_indices", np.arange(batch.n_structures))],
            batch.n_atoms, float(batch.x.sum(dtype=np.float64)))
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
                          "shard_atoms": [s[2] for s in shards],
                          "shard_structures": [len(s[1]) for s in shards]}), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
