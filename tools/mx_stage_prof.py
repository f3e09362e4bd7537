#!/usr/bin/env python3
"""Cycles per atom and stage of k_occlusion_mx on the bench batch (diagnostic build; GPU box).

    make -C rustsasa_amd/csrc OUT=../lib/variants/stageprof/librustsasa_amd.so EXTRA=-DMX_STAGE_PROF ../lib/variants/stageprof/librustsasa_amd.so
    python tools/mx_stage_prof.py [shard_of] [workload]

Every wave stamps s_memtime at the stage boundaries of its atom loop (occlusion_mx.inc, MX_STAMP); lane 0 adds the
deltas to per-wave LDS words, the wave adds those to a device buffer of their own at its end.  Printed: the share of a
wave's lifetime each stage takes and shader cycles per atom (WAVE time: seven waves share a SIMD, so the launch spends
about a seventh of these per atom and SIMD).  A stamp costs the wave about as much as eight instructions; the stamped
build's kernel time is printed beside the plain one."""
import ctypes as C
import os
os.environ.setdefault("RSASA_TUNING", "1")  # (the library reads its RSASA_* measurement switches only then)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustsasa_amd._capi as capi  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "rustsasa_amd", "lib", "variants", "stageprof", "librustsasa_amd.so")
import torch  # noqa: E402
import bench  # noqa: E402
import rustsasa_amd  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 0
workload = sys.argv[2] if len(sys.argv) > 2 else "proteome"
dev = torch.device("cuda:0")
batch, n_points, _ = bench.make_workload(workload, None, None, 0, 1, "strong", m)
lib = C.CDLL(capi.LIB_PATH)
lib.rsasa_debug_mx_prof.argtypes = [C.POINTER(C.c_ulonglong)]
names = {0: "wave start (tables, atoms, group starts)", 1: "group prologue (runs, scan, union gathers)",
         2: "atom header + sweep", 3: "prep", 4: "phase A (filter)", 5: "phase B (exact)", 6: "result word, loop",
         7: "rare paths", 11: "results out"}
with rustsasa_amd.Context(0) as ctx:
    ctx.enable_timing(True)
    run = bench.DeviceRun(ctx, batch, n_points, dev, True, None)
    for _ in range(3):
        run.step()
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    lib.rsasa_debug_mx_prof(out)
    steps = 5
    ms = []
    for _ in range(steps):
        run.step()
        ms.append(ctx.timings()["occlusion_ms"])
    torch.cuda.synchronize()
    lib.rsasa_debug_mx_prof(out)
    atoms, waves, groups, gatoms, gchunks = out[12], out[13], out[8], out[9], out[10]
    tot = sum(out[k] for k in names)
    print(f"{workload} shard_of={m}: {batch.n_structures} structures, {batch.n_atoms} atoms x {steps} launches; waves {waves}, "
          f"atoms stamped {atoms}, groups {groups} ({gatoms / max(groups, 1):.2f} atoms, {gchunks / max(groups, 1):.2f} chunks each); "
          f"occlusion kernel of the stamped build {sum(ms) / len(ms):.3f} ms")
    print(f"  wave cycles per atom, all stages: {tot / atoms:.0f}   (per wave: {tot / waves:.0f})")
    for k, n in names.items():
        print(f"  {n:44s} {100.0 * out[k] / tot:5.1f} %   {out[k] / atoms:8.1f} cycles per atom")
