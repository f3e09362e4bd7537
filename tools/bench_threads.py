#!/usr/bin/env python3
"""Per-structure drop-in use from several host threads (what rayon workers would do through the
Rust shim of INTEGRATION.md): every thread owns a context and calls calculate_sasa_soa on one
structure at a time.  Prints structures/s for 1, 2, 4, 8, 16 threads."""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_workloads as bw  # noqa: E402
import rustsasa_amd  # noqa: E402

xyz, r, _, ids = bw.fixture_soa("example.cif")
x, y, z = (np.ascontiguousarray(xyz[:, k]).astype(np.float32) for k in range(3))
out = {}
for n_threads in (1, 2, 4, 8, 16):
    ctxs = [rustsasa_amd.Context(0) for _ in range(n_threads)]
    calls = 2000

    def work(c):
        for _ in range(calls):
            c.calculate_sasa_soa(x, y, z, r, ids, 1.4, 100)

    for c in ctxs:
        c.calculate_sasa_soa(x, y, z, r, ids, 1.4, 100)
    ths = [threading.Thread(target=work, args=(c,)) for c in ctxs]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    out[f"{n_threads}_threads"] = round(n_threads * calls / dt, 1)
    for c in ctxs:
        c.close()
print(json.dumps({"structure": "example.cif (2622 atoms), 100 points", "structures_per_s": out}))
