#!/usr/bin/env python3
"""A/B of the window-aligned wave blocks (-DMX_ALIGNED build under lib/variants/al): the shipped library
with 64 / 60 / 56 atoms per wave (every wave block cuts a group) against the aligned build with windows of 60 / 56 atoms and
a tail of 4 / 8 (a wave follows its last group past its window), interleaved on one box.  Also checks the aligned build's
values against the shipped library's (bit-equal)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = ("import sys; sys.path.insert(0, %r); import rustsasa_amd._capi as c; c.LIB_PATH = %r; "
         "import bench; sys.argv = ['bench.py'] + %r; bench.main()")
ARGS = ["--steps", "10", "--warmup", "2", "--cpu-seconds", "0", "--h2h-steps", "0", "--two-steps", "0", "--config5-steps", "0",
        "--files", "0", "--per-call-seconds", "0", "--real-steps", "0", "--hashed-ids-steps", "0"]
BASE = os.path.join(ROOT, "rustsasa_amd", "lib", "librustsasa_amd.so")
AL = os.path.join(ROOT, "rustsasa_amd", "lib", "variants", "al", "librustsasa_amd.so")
legs = [("shipped, 64 per wave", BASE, None), ("shipped, 60 per wave", BASE, "60"), ("shipped, 56 per wave", BASE, "56"),
        ("aligned, window 60 + tail 4", AL, "60"), ("aligned, window 56 + tail 8", AL, "56"), ("aligned, window 48 + tail 16", AL, "48")]
res = {n: [] for n, _, _ in legs}
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for name, lib, apw in legs:
        env = dict(os.environ, RSASA_TUNING="1")
        if apw:
            env["RSASA_ATOMS_PER_WAVE"] = apw
        p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, lib, ARGS)], capture_output=True, text=True, cwd=ROOT, env=env)
        if p.returncode != 0:
            print(name, "FAILED", p.stderr[-300:])
            continue
        d = json.loads(p.stdout.strip().split("\n")[-1])
        res[name].append((d["kernel_ms"]["occlusion"], d["ms_per_step"], d.get("total_sasa"), (d.get("parity") or {}).get("atoms_differ")))
for name, v in res.items():
    if v:
        print(f"{name:32s} occlusion ms min {min(x[0] for x in v):.4f} all {[round(x[0], 3) for x in v]} step min {min(x[1] for x in v):.4f} total_sasa {v[0][2]}")
