import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
import rustsasa_amd, bench_workloads as bw
b = bw.synthetic_proteome(512, seed=3)
small = bw.synthetic_proteome(20, seed=4)   # > 32768 atoms: the matrix-core path
print("atoms", b.n_atoms, small.n_atoms)
for order in ("big first", "small then big"):
    ctx = rustsasa_amd.Context(0)
    t = []
    def call(x):
        t0 = time.perf_counter()
        ctx.calculate_sasa_batch(x.x, x.y, x.z, x.radius, x.ids, x.structure_offsets, 1.4, 100, residue_offsets=x.residue_offsets)
        t.append(1e3 * (time.perf_counter() - t0))
    if order == "small then big": call(small)
    call(b); call(b); call(b)
    print(order, [round(v, 1) for v in t])
    ctx.close()
