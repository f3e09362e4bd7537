"""Every atom of three batches (the proteome, the quality set x 3, a dense 200 k-atom structure) of a library variant under
rustsasa_amd/lib/variants/<name>/ against the oracle: python tools/experiments/parity_of_variant.py <name>   (GPU box)"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ["RSASA_TUNING"] = "1"
import numpy as np
import rustsasa_amd._capi as c
c.LIB_PATH = os.path.join(os.getcwd(), "rustsasa_amd", "lib", "variants", sys.argv[1], "librustsasa_amd.so")
import rustsasa_amd, bench_workloads as bw, real_coords as rc
from oracle import pyoracle as po
import torch
dev = torch.device("cuda:0")
def run(ctx, b):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = torch.full((b.n_atoms,), -1.0, dtype=torch.float32, device=dev)
    k = torch.zeros(b.n_atoms, dtype=torch.int32, device=dev)
    ctx.enqueue_device(t(b.x), t(b.y), t(b.z), t(b.radius), t(b.ids.view(np.int64)), b.structure_offsets, out, None, None, k, 1.4, 100)
    ctx.wait()
    return out.cpu().numpy(), k.cpu().numpy()
with rustsasa_amd.Context(0) as ctx:
    for name, b in (("proteome", bw.synthetic_proteome()), ("real x3", rc.tiled(rc.quality_set_batch(), 1_370_000, seed=7)), ("uniform 200k", bw.synthetic_uniform(200_000, seed=4))):
        want, _, wk = None, None, None
        want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, 1.4, 100, 8, threads=0)
        got, k = run(ctx, b)
        print(name, os.environ.get("RSASA_ATOMS_PER_WAVE"), "atoms differ:", int((got != want).sum()), "of", b.n_atoms, "K mean", float(k.mean()))
