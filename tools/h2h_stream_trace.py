#!/usr/bin/env python3
"""Timeline of a stream of host batches (RSASA_H2H_TRACE=1): proteome batches through rsasa_host_batch_enqueue / _wait,
the host-side phases of every call and the device-side start and end of every sub-batch's uploads and kernels on one
clock (stderr).  usage: tools/h2h_stream_trace.py [batches] [repetitions]"""
import os, sys, time
os.environ.setdefault("RSASA_TUNING", "1")  # (the library reads its RSASA_* measurement switches only then)
os.environ["RSASA_H2H_TRACE"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench_workloads as bw
import rustsasa_amd
b = bw.synthetic_proteome(seed=bw.PROTEOME_SEED)
b = bw.select(b, bw.shard_largest_first(np.diff(b.structure_offsets.astype(np.int64)), 1)[0])
pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()
x, y, z, r, ids, ro = pin(b.x), pin(b.y), pin(b.z), pin(b.radius), pin(b.ids), pin(b.residue_offsets)
outs = [pin(np.zeros(b.n_residues, np.float32)) for _ in range(2)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
with rustsasa_amd.Context(0) as ctx:
    def enq(k):
        ctx.host_batch_enqueue(x, y, z, r, ids, b.structure_offsets, 1.4, 100, residue_offsets=ro, want_atoms=False, res_out=outs[k % 2])
    for rep in range(REPS):
        print(f"==== round {rep}", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        enq(0)
        for k in range(1, N):
            enq(k)
            ctx.host_batch_wait()
        ctx.host_batch_wait()
        print(f"==== {N} batches in {(time.perf_counter() - t0) * 1e3:.2f} ms", file=sys.stderr, flush=True)
