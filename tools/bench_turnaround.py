#!/usr/bin/env python3
"""Host-side cost of one device-resident batch call: time inside enqueue_device, inside wait, and the
step period, with and without the context's event timing."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench_workloads as bw
import rustsasa_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else bw.PROTEOME_STRUCTURES
b = bw.synthetic_proteome(n, seed=bw.PROTEOME_SEED)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
x, y, z, r = t(b.x), t(b.y), t(b.z), t(b.radius)
ids = t(b.ids.view(np.int64))
ro = t(b.residue_offsets.view(np.int32))
out = torch.zeros(b.n_atoms, dtype=torch.float32, device=dev)
res = torch.zeros(b.n_residues, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
with rustsasa_amd.Context(0) as ctx:
    for timing in (False, True):
        ctx.enable_timing(timing)
        for _ in range(5):
            ctx.enqueue_device(x, y, z, r, ids, b.structure_offsets, out, ro, res, None, 1.4, 100, stream=st); ctx.wait()
        te = tw = 0.0
        K = 50
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K):
            a = time.perf_counter()
            ctx.enqueue_device(x, y, z, r, ids, b.structure_offsets, out, ro, res, None, 1.4, 100, stream=st)
            c = time.perf_counter()
            ctx.wait()
            d = time.perf_counter()
            te += c - a; tw += d - c
        tot = time.perf_counter() - t0
        print(f"structures {b.n_structures} timing {timing}: period {tot / K * 1e3:.4f} ms, in enqueue {te / K * 1e3:.4f} ms, in wait {tw / K * 1e3:.4f} ms")
