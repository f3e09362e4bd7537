#!/usr/bin/env python3
"""A STREAM of host batches (pinned SoA in, residue values out): T host threads, each with a context of its own on the
same GPU, call rsasa_calculate_sasa_batch back to back on the proteome batch - call k + 1's uploads cross the link
while call k's last sub-batches compute and download.  Prints ms per batch for T = 1, 2, 3."""
import os, sys, time, threading
os.environ.setdefault("RSASA_TUNING", "1")  # (the library reads its RSASA_* measurement switches only then)
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
api = "--api" in sys.argv  # the stream API (rsasa_host_batch_enqueue / _wait) instead of host threads with a context each
args = [a for a in sys.argv[1:] if a != "--api"]
steps = int(args[0]) if args else 12
if api:
    with rustsasa_amd.Context(0) as ctx:
        depth = int(os.environ.get("H2H_DEPTH", "3"))  # host batches in flight
        outs = [pin(np.zeros(b.n_residues, np.float32)) for _ in range(depth)]
        def enq(k):
            ctx.host_batch_enqueue(x, y, z, r, ids, b.structure_offsets, 1.4, 100, residue_offsets=ro, want_atoms=False, res_out=outs[k % depth])
        for rep in range(int(os.environ.get("H2H_REPS", "2"))):
            t0 = time.perf_counter()
            for k in range(min(depth - 1, steps)):
                enq(k)
            for k in range(depth - 1, steps):
                enq(k)
                ctx.host_batch_wait()
            ctx.host_batch_wait_all()
            dt = (time.perf_counter() - t0) / steps
            print(f"stream API, {steps} batches: {dt * 1e3:.3f} ms per proteome batch, {b.n_structures / dt:.0f} structures/s, outputs equal {all(np.array_equal(outs[0], o) for o in outs)}", flush=True)
    sys.stdout.flush()
    sys.exit(0)
for T in (1, 2, 3):
    ctxs = [rustsasa_amd.Context(0) for _ in range(T)]
    outs = [pin(np.zeros(b.n_residues, np.float32)) for _ in range(T)]
    def call(t):
        ctxs[t].calculate_sasa_batch(x, y, z, r, ids, b.structure_offsets, 1.4, 100, residue_offsets=ro, want_atoms=False, res_out=outs[t])
    for t in range(T):
        for _ in range(3):
            call(t)
    def work(t):
        for _ in range(steps):
            call(t)
    ths = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    dt = (time.perf_counter() - t0) / (T * steps)
    same = all(np.array_equal(outs[0], o) for o in outs)
    print(f"{T} context(s) / thread(s): {dt * 1e3:.3f} ms per proteome batch, {b.n_structures / dt:.0f} structures/s, outputs equal {same}", flush=True)
    for c in ctxs: c.close()
