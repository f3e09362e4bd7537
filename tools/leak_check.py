#!/usr/bin/env python3
"""Memory growth check on the GPU box: repeats the host-batch, device-batch (two in flight), host-batch-stream
(rsasa_host_batch_enqueue / _wait, three queued) and per-structure entry points, creates and destroys contexts that have
run a stream (their two worker contexts and threads go with them), and prints the process RSS and the free device memory before and after.  usage: tools/leak_check.py [rounds]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench_workloads as bw
import rustsasa_amd


def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024.0


def free_mb():
    return torch.cuda.mem_get_info()[0] / 2**20


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
b = bw.synthetic_proteome(600, seed=bw.PROTEOME_SEED)
one = bw.synthetic_proteome(1, seed=3)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
with rustsasa_amd.Context(0) as ctx:
    x, y, z, r, ids = t(b.x), t(b.y), t(b.z), t(b.radius), t(b.ids.view(np.int64))
    ro = t(b.residue_offsets.view(np.int32))
    outs = [(torch.empty(b.n_atoms, dtype=torch.float32, device=dev), torch.empty(b.n_residues, dtype=torch.float32, device=dev)) for _ in range(2)]

    def cycle(n):
        for i in range(n):
            ctx.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, residue_offsets=b.residue_offsets, want_atoms=False)
            ctx.enqueue_device(x, y, z, r, ids, b.structure_offsets, outs[0][0], ro, outs[0][1])
            ctx.enqueue_device(x, y, z, r, ids, b.structure_offsets, outs[1][0], ro, outs[1][1])
            ctx.wait_all()
            for _ in range(20):
                ctx.calculate_sasa_soa(one.x, one.y, one.z, one.radius, one.ids)
            for k in range(3):
                ctx.host_batch_enqueue(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, residue_offsets=b.residue_offsets, want_atoms=k == 1)
            ctx.host_batch_wait_all()
            if i % 10 == 0:
                with rustsasa_amd.Context(0) as c2:   # a context that has run a stream: created and destroyed
                    c2.host_batch_enqueue(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, want_atoms=True)
                    c2.host_batch_wait()

    cycle(10)  # buffers reach their sizes
    torch.cuda.synchronize()
    r0, f0 = rss_mb(), free_mb()
    cycle(rounds)
    torch.cuda.synchronize()
    r1, f1 = rss_mb(), free_mb()
print(f"{rounds} rounds: host RSS {r0:.1f} -> {r1:.1f} MB ({r1 - r0:+.1f}), free device memory {f0:.0f} -> {f1:.0f} MB ({f1 - f0:+.0f})")
sys.exit(0 if abs(r1 - r0) < 64 and abs(f1 - f0) < 64 else 1)
