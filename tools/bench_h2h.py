#!/usr/bin/env python3
"""Host-to-host rate (pinned SoA in, residue values out) of the proteome batch for several sub-batch counts."""
import os, sys, time
os.environ.setdefault("RSASA_TUNING", "1")  # (the library reads its RSASA_* measurement switches only then)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench_workloads as bw
import rustsasa_amd
b = bw.synthetic_proteome(seed=bw.PROTEOME_SEED)
if os.environ.get("H2H_SORTED"):  # bench.py's order: largest structures first
    b = bw.select(b, bw.shard_largest_first(np.diff(b.structure_offsets.astype(np.int64)), 1)[0])
pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()
x, y, z, r, ids, ro = pin(b.x), pin(b.y), pin(b.z), pin(b.radius), pin(b.ids), pin(b.residue_offsets)
out = pin(np.zeros(b.n_residues, np.float32))
with rustsasa_amd.Context(0) as ctx:
    for n in sys.argv[1:] or ["8"]:
        os.environ["RSASA_SUB_BATCHES"] = n
        for _ in range(3):
            ctx.calculate_sasa_batch(x, y, z, r, ids, b.structure_offsets, 1.4, 100, residue_offsets=ro, want_atoms=False, res_out=out)
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.calculate_sasa_batch(x, y, z, r, ids, b.structure_offsets, 1.4, 100, residue_offsets=ro, want_atoms=False, res_out=out)
        dt = (time.perf_counter() - t0) / 10
        print(f"sub-batches {n}: {dt * 1e3:.3f} ms per proteome batch, {b.n_structures / dt:.0f} structures/s")
