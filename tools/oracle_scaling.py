#!/usr/bin/env python3
"""Thread scaling of the CPU oracle's batch path on this host (what bench.py's cpu_baseline runs).

    python tools/oracle_scaling.py [structures]

Prints atoms/s for 1, 2, 4, ... up to every hardware thread, best of three, on the first `structures`
(default: all 4 363) of the bench workload, every k-th structure for the slow legs."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_workloads as bw  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def physical_cores():
    seen = set()
    phys = core = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
                seen.add((phys, core))
    except OSError:
        pass
    return len(seen) or None


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else bw.PROTEOME_STRUCTURES
    full = bw.synthetic_proteome(n)
    sizes = np.diff(full.structure_offsets.astype(np.int64))
    order = np.argsort(-sizes, kind="stable")
    hw = po.max_threads()
    out = {"hardware_threads": hw, "physical_cores": physical_cores(), "legs": []}
    t = 1
    threads = []
    while t < hw:
        threads.append(t)
        t *= 2
    threads.append(hw)
    for th in threads:
        # about the same wall time per leg: every k-th structure of the largest-first order
        k = max(1, int(round(hw / th / 4)))
        b = bw.select(full, order[::k])
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, 1.4, 100, 8, threads=th)
            best = min(best, time.perf_counter() - t0)
        out["legs"].append({"threads": th, "structures": b.n_structures, "atoms": b.n_atoms, "seconds": round(best, 3),
                            "atoms_per_s": round(b.n_atoms / best, 1), "structures_per_s": round(b.n_structures / best, 2)})
        print(out["legs"][-1], flush=True)
    base = out["legs"][0]["atoms_per_s"]
    for leg in out["legs"]:
        leg["speedup_vs_1"] = round(leg["atoms_per_s"] / base, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
