#!/usr/bin/env python3
"""Every atom and residue of the bench workload (11.7 M atoms) against the oracle - a one-off check
(the test suite compares a slice and random structures of it)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench_workloads as bw, rustsasa_amd
from oracle import pyoracle as po
b = bw.synthetic_proteome(seed=bw.PROTEOME_SEED)
dev = torch.device('cuda:0')
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
x, y, z, r = t(b.x), t(b.y), t(b.z), t(b.radius)
ids = t(b.ids.view(np.int64)); ro = t(b.residue_offsets.view(np.int32))
out = torch.empty(b.n_atoms, dtype=torch.float32, device=dev); res = torch.empty(b.n_residues, dtype=torch.float32, device=dev)
k = torch.zeros(b.n_atoms, dtype=torch.int32, device=dev)
with rustsasa_amd.Context(0) as ctx:
    ctx.enqueue_device(x, y, z, r, ids, b.structure_offsets, out, ro, res, k, 1.4, 100, stream=torch.cuda.current_stream().cuda_stream)
    ctx.wait()
t0 = time.time()
want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, 1.4, 100, 8, threads=128)
print('oracle s', round(time.time() - t0, 1))
got = out.cpu().numpy()
print('atoms', b.n_atoms, 'mismatching atoms', int(np.sum(got != want)), 'max abs diff', float(np.max(np.abs(got - want))))
wr = po.residue_sums(want, b.residue_offsets)
print('residues', b.n_residues, 'mismatching', int(np.sum(res.cpu().numpy() != wr)))
