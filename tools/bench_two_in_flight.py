#!/usr/bin/env python3
"""What two batches in flight would give: two contexts (two workspaces, two streams) alternate on the
same device-resident batch; batch k + 1 is enqueued before batch k is waited for.  An experiment for
DESIGN.md (the product API holds one batch per context)."""
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
outs = [torch.zeros(b.n_atoms, dtype=torch.float32, device=dev) for _ in range(2)]
ress = [torch.zeros(b.n_residues, dtype=torch.float32, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
ctxs = [rustsasa_amd.Context(0) for _ in range(2)]


def enq(k):
    ctxs[k].enqueue_device(x, y, z, r, ids, b.structure_offsets, outs[k], ro, ress[k], None, 1.4, 100,
                           stream=streams[k].cuda_stream)


K = 100
for mode in ("one in flight", "two in flight"):
    for k in range(2):
        for _ in range(3):
            enq(k); ctxs[k].wait()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == "one in flight":
        for i in range(K):
            enq(0); ctxs[0].wait()
    else:
        enq(0)
        for i in range(1, K):
            enq(i % 2)
            ctxs[(i - 1) % 2].wait()
        ctxs[(K - 1) % 2].wait()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{b.n_structures} structures, {mode}: {dt * 1e3:.4f} ms per batch, {b.n_structures / dt:.0f} structures/s")
assert torch.equal(outs[0], outs[1])
