#!/usr/bin/env python3
"""Per-atom statistics of k_occlusion_mx on the bench workload from the statistics builds
   make -C rustsasa_amd/csrc OUT=../lib/variants/statN/librustsasa_amd.so EXTRA=-DMX_STAT=N ...  (N = 1: survivors of
phase A, 2: near candidates, 3: candidate tiles); each reports its quantity through the neighbour-count output."""
import os
os.environ.setdefault("RSASA_TUNING", "1")  # (the library reads its RSASA_* measurement switches only then)
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import rustsasa_amd._capi as c
c.LIB_PATH = %r
import torch, bench_workloads as bw, rustsasa_amd
b = bw.synthetic_proteome(800, seed=bw.PROTEOME_SEED)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
x, y, z, r = t(b.x), t(b.y), t(b.z), t(b.radius)
ids = t(b.ids.view(np.int64))
with rustsasa_amd.Context(0) as ctx:
    out = torch.empty(b.n_atoms, dtype=torch.float32, device=dev)
    k = torch.zeros(b.n_atoms, dtype=torch.int32, device=dev)
    ctx.enqueue_device(x, y, z, r, ids, b.structure_offsets, out, None, None, k, 1.4, 100, stream=torch.cuda.current_stream().cuda_stream)
    ctx.wait()
v = k.cpu().numpy().astype(np.float64)
q = np.percentile(v, [50, 90, 99, 100])
print(%r, "mean %%.2f p50 %%d p90 %%d p99 %%d max %%d  frac==0 %%.3f  frac>16 %%.3f frac>32 %%.3f  1..4: %%s  5..8: %%.3f  9..16: %%.3f" %% (v.mean(), q[0], q[1], q[2], q[3], np.mean(v == 0), np.mean(v > 16), np.mean(v > 32), [round(float(np.mean(v == i)), 3) for i in (1, 2, 3, 4)], np.mean((v >= 5) & (v <= 8)), np.mean((v >= 9) & (v <= 16))))
"""
names = {1: "S (survivors of phase A, fused-rule points)", 2: "nA (near candidates)", 3: "candidate tiles", 5: "remainder points alive after the filter"}
for n, label in names.items():
    lib = os.path.join(ROOT, "rustsasa_amd", "lib", "variants", f"stat{n}", "librustsasa_amd.so")
    if os.path.exists(lib):
        subprocess.run([sys.executable, "-c", CHILD % (ROOT, lib, label)], cwd=ROOT)
