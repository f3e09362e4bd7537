#!/usr/bin/env python3
"""Headline benchmark: structures/sec of the Shrake-Rupley hot path on MI355X.

One "step" = one pass of the whole hot path (bounds -> cell grid -> counting
sort -> occlusion -> ResidueLevel sums) over one batch of synthetic structures
that is already resident in HBM.  The workload is BASELINE.json configs[2]:
an AlphaFold-E.-coli-like proteome (4 363 structures, 100 sphere points,
probe 1.4 A, ResidueLevel), synthesised offline-reproducibly by
bench_workloads.synthetic_proteome (seed 20260807 + rank).

Multi-GPU (driver: torchrun, one rank per GPU): structures are independent, so
each rank processes its own proteome-sized batch with no data-path collective
(weak scaling); RCCL is used only for the barrier and the final max/sum gather.

Prints ONE JSON line on rank 0 (see the field notes in DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench_workloads as bw  # noqa: E402
import rustsasa_amd  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PROBE = 1.4
N_POINTS = 100


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--structures", type=int, default=bw.PROTEOME_STRUCTURES,
                   help="structures per GPU (default: the full 4 363-structure proteome)")
    p.add_argument("--workload", choices=["proteome", "uniform1m"], default="proteome")
    p.add_argument("--n-points", type=int, default=None)
    p.add_argument("--cpu-seconds", type=float, default=15.0,
                   help="target wall time of the CPU baseline sample (0 disables it)")
    p.add_argument("--no-ids", action="store_true", help="pass id = NULL (all atoms distinct)")
    return p.parse_args()


def cpu_baseline(batch, n_points, target_seconds):
    """The oracle (a port of the reference's CPU path) on a bounded sample of the same workload."""
    from oracle import pyoracle as po
    threads = min(po.max_threads(), os.cpu_count() or 1)

    def run(n_struct):
        e = int(batch.structure_offsets[n_struct])
        t0 = time.perf_counter()
        po.calculate_sasa_batch(batch.x[:e], batch.y[:e], batch.z[:e], batch.radius[:e],
                                batch.ids[:e], batch.structure_offsets[:n_struct + 1], PROBE,
                                n_points, 8, threads=threads)
        return time.perf_counter() - t0

    probe_n = min(batch.n_structures, max(threads * 4, 16))
    run(min(probe_n, 8))  # warm up the thread pool and page in the library
    t_probe = run(probe_n)
    n = int(min(batch.n_structures, max(probe_n, probe_n * target_seconds / max(t_probe, 1e-6))))
    t = run(n) if n > probe_n else t_probe
    atoms = int(batch.structure_offsets[n])
    return {"value": round(n / t, 3), "unit": "structures/s", "cores": threads, "kind": "port",
            "sample": f"first {n} of {batch.n_structures} structures ({atoms} atoms) of the same "
                      f"workload, {n_points} points, oracle/sasa_oracle.c with OpenMP over "
                      f"structures, {t:.1f} s wall"}


def aggregate(dist, dev, elapsed, n_structures, n_atoms):
    """Whole-job numbers from per-rank ones: MAX of the elapsed times, SUM of the units.
    `dist` is torch.distributed (RCCL on GPUs, gloo in the CPU tests) or None for one rank."""
    units = torch.tensor([float(n_structures), float(n_atoms)], device=dev, dtype=torch.float64)
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
    return float(el.item()), float(units[0].item()), float(units[1].item())


def make_workload(workload, structures, n_points, rank):
    """Per-rank batch of independent structures (weak scaling: every rank gets its own)."""
    if workload == "proteome":
        n_points = n_points or N_POINTS
        batch = bw.synthetic_proteome(structures, seed=bw.PROTEOME_SEED + rank)
        name = (f"synthetic AlphaFold-E.coli-like proteome, {batch.n_structures} structures/GPU, "
                f"{n_points} points, probe {PROBE}, ResidueLevel")
    else:
        n_points = n_points or 960
        batch = bw.synthetic_uniform(1_000_000, seed=5 + rank)
        name = f"synthetic 1M-atom structure, {n_points} points, probe {PROBE}, AtomLevel"
    return batch, n_points, name


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- workload (independent structures; each rank gets its own batch) ----
    batch, n_points, name = make_workload(args.workload, args.structures, args.n_points, rank)

    def dv(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    x, y, z, r = dv(batch.x), dv(batch.y), dv(batch.z), dv(batch.radius)
    ids = None if args.no_ids else dv(batch.ids.view(np.int64))
    res_off = dv(batch.residue_offsets.view(np.int32))
    out_atom = torch.empty(batch.n_atoms, dtype=torch.float32, device=dev)
    out_res = torch.empty(batch.n_residues, dtype=torch.float32, device=dev)
    kcount = torch.zeros(batch.n_atoms, dtype=torch.int32, device=dev)

    ctx = rustsasa_amd.Context(local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    def step(counts=None):
        ctx.enqueue_device(x, y, z, r, ids, batch.structure_offsets, out_atom, res_off, out_res,
                           counts, PROBE, n_points, stream=stream)
        ctx.wait()

    # candidate counts K_i (deterministic for a given input) for the algorithmic byte count
    step(kcount)
    k_sum = int(kcount.to(torch.int64).sum().item())
    algorithmic_bytes = 20 * batch.n_atoms + 16 * k_sum  # 16 + 16*K + 4 per atom

    for _ in range(args.warmup):
        step()

    ctx.enable_timing(True)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    occl_ms, grid_ms, agg_ms = [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        t = ctx.timings()
        n_cells = int(t["n_cells"])
        occl_ms.append(t["occlusion_ms"])
        grid_ms.append(t["grid_build_ms"])
        agg_ms.append(t["aggregate_ms"])
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    elapsed, total_structures, total_atoms = aggregate(dist, dev, elapsed, batch.n_structures,
                                                       batch.n_atoms)

    if rank == 0:
        occl = float(np.mean(occl_ms))
        achieved = algorithmic_bytes / (occl * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_occlusion.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "structures/sec on AF2 E. coli proteome (100 pts, 1.4 A probe)",
            "value": round(total_structures * args.steps / elapsed, 2),
            "unit": "structures/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": name, "structures_per_gpu": batch.n_structures,
                       "atoms_per_gpu": batch.n_atoms, "residues_per_gpu": batch.n_residues,
                       "candidates_per_atom": round(k_sum / max(batch.n_atoms, 1), 2),
                       "grid_cells_per_gpu": n_cells,
                       "atoms_per_s": round(total_atoms * args.steps / elapsed, 1),
                       "ids": not args.no_ids, "parallelism": f"{world} x independent shards"},
            "roofline": {"bound": "hbm", "kernel": "k_occlusion", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": algorithmic_bytes,
                         "kernel_ms": round(occl, 4)},
            "kernel_ms": {"grid_build": round(float(np.mean(grid_ms)), 4),
                          "occlusion": round(occl, 4),
                          "residue_sums": round(float(np.mean(agg_ms)), 4)},
        }
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(batch, n_points, args.cpu_seconds)
        print(json.dumps(line), flush=True)

    ctx.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
