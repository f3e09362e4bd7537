#!/usr/bin/env python3
"""Per-atom work statistics of k_occlusion_v3 on the bench workload (ablation build).

RSASA_DEBUG_STOP=10..19 makes the kernel return a per-atom statistic through the
neighbour-count output (results stay correct).  Prints the means; used together with
tools/isa_blocks.py to attribute instruction counts to stages.
"""
import os
os.environ.setdefault("RSASA_TUNING", "1")  # (the library reads its RSASA_* measurement switches only then)
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench_workloads as bw  # noqa: E402

STATS = {10: "T  (atoms in the culled runs)", 11: "sweep iterations", 12: "nA (near candidates, summed over flushes)",
         13: "S  (survivors after phase A)", 14: "phase-B tile steps", 15: "flushes", 16: "remainder steps",
         17: "phase-A trips", 18: "tiled phase-B passes", 19: "generic phase-B candidates", 0: "K  (candidates)"}


def main():
    import rustsasa_amd
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    b = bw.synthetic_proteome(n, seed=bw.PROTEOME_SEED)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    x, y, z, r = t(b.x), t(b.y), t(b.z), t(b.radius)
    ids = t(b.ids.view(np.int64))
    for sel, label in STATS.items():
        os.environ["RSASA_DEBUG_STOP"] = str(sel)
        with rustsasa_amd.Context(0) as ctx:
            out = torch.empty(b.n_atoms, dtype=torch.float32, device=dev)
            k = torch.zeros(b.n_atoms, dtype=torch.int32, device=dev)
            ctx.enqueue_device(x, y, z, r, ids, b.structure_offsets, out, None, None, k, 1.4, 100,
                               stream=torch.cuda.current_stream().cuda_stream)
            ctx.wait()
            v = k.cpu().numpy().view(np.uint32).astype(np.float64)
        q = np.percentile(v, [50, 90, 99, 100])
        print(f"{sel:3d} {label:45s} mean {v.mean():8.3f}  p50 {q[0]:6.0f} p90 {q[1]:6.0f} p99 {q[2]:6.0f} max {q[3]:6.0f}"
              f"  frac>0 {np.mean(v > 0):.3f}")


if __name__ == "__main__":
    main()
