#!/usr/bin/env python3
"""Latency of one calculate_sasa_internal-style call (host buffers in, host buffers out) for the
single-structure BASELINE configs, and the PCIe-inclusive rate of the host batch call."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_workloads as bw  # noqa: E402
import rustsasa_amd  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

out = {}
with rustsasa_amd.Context(0) as ctx:
    for name in ("1jcd.pdb", "151L_H3.pdb", "example.cif"):
        xyz, r, _, ids = bw.fixture_soa(name)
        x, y, z = (np.ascontiguousarray(xyz[:, k]).astype(np.float32) for k in range(3))
        for n_points in (100, 960):
            for _ in range(5):
                got = ctx.calculate_sasa_soa(x, y, z, r, ids, 1.4, n_points)
            t0 = time.perf_counter()
            reps = 200
            for _ in range(reps):
                got = ctx.calculate_sasa_soa(x, y, z, r, ids, 1.4, n_points)
            gpu_ms = (time.perf_counter() - t0) / reps * 1e3
            t0 = time.perf_counter()
            for _ in range(10):
                want = po.calculate_sasa_internal(x, y, z, r, ids, 1.4, n_points, 8)
            cpu_ms = (time.perf_counter() - t0) / 10 * 1e3
            out[f"{name}:{n_points}"] = {"atoms": len(x), "gpu_ms_per_call": round(gpu_ms, 4),
                                         "cpu_1thread_ms": round(cpu_ms, 3),
                                         "max_abs_diff": float(np.max(np.abs(got - want)))}
    b = bw.synthetic_proteome(seed=bw.PROTEOME_SEED)
    for _ in range(2):
        ctx.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, 1.4, 100,
                                 residue_offsets=b.residue_offsets, want_atoms=False)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, 1.4, 100,
                                 residue_offsets=b.residue_offsets, want_atoms=False)
    dt = (time.perf_counter() - t0) / 5
    out["proteome_host_buffers"] = {"structures_per_s": round(b.n_structures / dt, 1),
                                    "ms_per_batch": round(dt * 1e3, 2),
                                    "note": "pageable host SoA in, residue values out (PCIe inclusive)"}
    # MD-trajectory mode: one topology, many frames; only xyz crosses PCIe per frame
    xyz, r, res_off, ids = bw.fixture_soa("example.cif")
    rng = np.random.default_rng(4)
    n_frames = 2000
    frames = (xyz[None, :, :] + rng.normal(scale=0.3, size=(n_frames, xyz.shape[0], 3))).astype(np.float32)
    for _ in range(2):
        ctx.calculate_sasa_trajectory(frames, r, ids, 1.4, 100, residue_offsets=res_off, want_atoms=False)
    t0 = time.perf_counter()
    for _ in range(3):
        _, res = ctx.calculate_sasa_trajectory(frames, r, ids, 1.4, 100, residue_offsets=res_off, want_atoms=False)
    dt = (time.perf_counter() - t0) / 3
    out["trajectory_example_cif"] = {"atoms": int(xyz.shape[0]), "frames": n_frames,
                                     "frames_per_s": round(n_frames / dt, 1), "ms_total": round(dt * 1e3, 2),
                                     "note": "frame-major xyz on the host in, per-residue values out"}
print(json.dumps(out, indent=1))
