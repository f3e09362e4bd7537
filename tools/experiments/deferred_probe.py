"""How many atoms k_occlusion_mx leaves to the general kernel, by workload and point count (100 points at W = 8 have 4
remainder points - the certainty band of DESIGN.md 3 -, 96 have none): python tools/experiments/deferred_probe.py  (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rustsasa_amd, bench_workloads as bw, real_coords as rc
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
with rustsasa_amd.Context(0) as ctx:
    for name, b in (("synthetic proteome", bw.synthetic_proteome()), ("quality set x 26", rc.tiled(rc.quality_set_batch(), 11_700_000))):
        x, y, z, r, ids = t(b.x), t(b.y), t(b.z), t(b.radius), t(b.ids.view(np.int64))
        out = torch.empty(b.n_atoms, dtype=torch.float32, device=dev)
        for n_points in (100, 96, 128):
            for with_ids in (True, False):
                for _ in range(2):
                    ctx.enable_timing(True)
                    ctx.enqueue_device(x, y, z, r, ids if with_ids else None, b.structure_offsets, out, None, None, None, 1.4, n_points)
                    ctx.wait()
                    tm = ctx.timings()
                    ctx.enable_timing(False)
                print(f"{name:20s} {n_points:4d} points, ids {'passed' if with_ids else 'none  '}: deferred {int(tm['n_deferred']):7d} of {b.n_atoms} "
                      f"({100.0 * tm['n_deferred'] / b.n_atoms:.3f} %), occlusion {tm['occlusion_ms']:.3f} ms")
