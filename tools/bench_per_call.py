#!/usr/bin/env python3
"""The per-structure drop-in call from several host threads, alone and with call combining (bench.py's `per_call` leg by
itself; GPU box):

    python tools/bench_per_call.py [seconds per leg] [legs ...]      legs: 16 c16 c64 s16 ... (see csrc/host/bench_per_call.cpp)
    PER_CALL_COMBINE_WAIT_US=20 RSASA_TUNING=1 RSASA_COMBINE_LANES=3 python tools/bench_per_call.py 1 c16 c64
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_workloads as bw  # noqa: E402
import rustsasa_amd  # noqa: E402


def main():
    seconds = sys.argv[1] if len(sys.argv) > 1 else "1"
    legs = sys.argv[2:] or ["1", "16", "c16", "c64", "s16", "s64"]
    batch = bw.synthetic_proteome()
    sizes = np.diff(batch.structure_offsets.astype(np.int64))
    batch = bw.select(batch, np.argsort(-sizes, kind="stable"))
    b = bw.select(batch, np.arange(0, batch.n_structures, 16))
    atoms = rustsasa_amd.make_atoms(b.x, b.y, b.z, b.radius, b.ids)
    keep = os.environ.get("PER_CALL_BIN")  # write the structures there and stop (for a profiler run of the executable itself)
    if keep:
        fd, path = os.open(keep, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644), keep
    else:
        fd, path = tempfile.mkstemp(prefix="rsasa_per_call_", suffix=".bin", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        with os.fdopen(fd, "wb") as f:
            f.write(np.uint32(b.n_structures).tobytes())
            f.write(b.structure_offsets.astype(np.uint32).tobytes())
            f.write(atoms.tobytes())
        exe = os.path.join(ROOT, "rustsasa_amd", "lib", "bench_per_call")
        if keep:
            print(exe, path, "100", seconds, *legs)
            return
        p = subprocess.run([exe, path, "100", seconds] + legs, capture_output=True, text=True)
        print(p.stdout, p.stderr[-500:])
    finally:
        if not keep:
            os.unlink(path)


if __name__ == "__main__":
    main()
