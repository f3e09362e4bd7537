#!/usr/bin/env python3
"""Per-phase time of k_sort_window on the bench batch (throw-away build).

    make -C rustsasa_amd/csrc OUT=../lib/variants/sortprof/librustsasa_amd.so EXTRA=-DRSASA_SORT_PROF ../lib/variants/sortprof/librustsasa_amd.so
    python tools/sort_prof.py [shard_of]          (on the GPU box)

Thread 0 of every workgroup stamps s_memrealtime (10 ns ticks) behind the barriers of the kernel; the sums over all
workgroups of one launch are printed per phase, as a share and as microseconds per workgroup."""
import ctypes as C
import os
os.environ.setdefault("RSASA_TUNING", "1")  # (the library reads its RSASA_* measurement switches only then)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustsasa_amd._capi as capi  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "rustsasa_amd", "lib", "variants", "sortprof", "librustsasa_amd.so")
import torch  # noqa: E402
import bench  # noqa: E402
import rustsasa_amd  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 0
workload = sys.argv[2] if len(sys.argv) > 2 else "proteome"
dev = torch.device("cuda:0")
batch, n_points, _ = bench.make_workload(workload, None, None, 0, 1, "strong", m)
lib = C.CDLL(capi.LIB_PATH)
lib.rsasa_debug_sort_prof.argtypes = [C.POINTER(C.c_ulonglong)]
names = ["load xyz, cells of the slots", "count (LDS atomics)", "scan", "cell starts out", "positions (LDS atomics)",
         "records in, staged", "records out"]
with rustsasa_amd.Context(0) as ctx:
    run = bench.DeviceRun(ctx, batch, n_points, dev, True, None)
    for _ in range(3):
        run.step()
    out = (C.c_ulonglong * 16)()
    lib.rsasa_debug_sort_prof(out)
    steps = 5
    for _ in range(steps):
        run.step()
    torch.cuda.synchronize()
    lib.rsasa_debug_sort_prof(out)
    wg = out[8] / steps
    tot = sum(out[k] for k in range(7))
    print(f"{workload} shard_of={m}: {batch.n_structures} structures, {batch.n_atoms} atoms; workgroups per launch {wg:.0f}, "
          f"cells per workgroup {out[9] / out[8]:.0f}, atoms binned per workgroup {out[10] / out[8]:.0f}, atoms looked at {out[11] / out[8]:.0f}")
    for k in range(7):
        print(f"  {names[k]:32s} {100.0 * out[k] / tot:5.1f} %   {out[k] * 0.01 / out[8]:7.2f} us per workgroup")
    print(f"  total {tot * 0.01 / out[8]:.2f} us per workgroup; x workgroups / 512 resident = {tot * 0.01 / out[8] * wg / 512:.1f} us per launch")
