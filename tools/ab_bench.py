#!/usr/bin/env python3
"""A/B timing of engine library variants on the bench workload.

    make -C rustsasa_amd/csrc OUT=../lib/variants/NAME/librustsasa_amd.so EXTRA=-DSOMETHING ../lib/variants/NAME/librustsasa_amd.so
    python tools/ab_bench.py [--rounds 3] [bench.py args]        (on the GPU box)

Runs bench.py once per variant and round (interleaved, so drift hits all variants alike) in
child processes whose loader points at the variant, and prints the occlusion-kernel time.
"""
import glob
import json
import os
os.environ.setdefault("RSASA_TUNING", "1")  # (the library reads its RSASA_* measurement switches only then)
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = ("import sys; sys.path.insert(0, %r); import rustsasa_amd._capi as c; c.LIB_PATH = %r; "
         "import bench; sys.argv = ['bench.py'] + %r; bench.main()")


def main():
    args = sys.argv[1:]
    rounds = 3
    if args[:1] == ["--rounds"]:
        rounds, args = int(args[1]), args[2:]
    args = args or ["--steps", "10", "--warmup", "2", "--cpu-seconds", "0", "--h2h-steps", "0", "--two-steps", "0",
                    "--config5-steps", "0", "--files", "0", "--per-call-seconds", "0", "--real-steps", "0", "--hashed-ids-steps", "0"]
    libs = {"base": os.path.join(ROOT, "rustsasa_amd", "lib", "librustsasa_amd.so")}
    for p in sorted(glob.glob(os.path.join(ROOT, "rustsasa_amd", "lib", "variants", "*", "librustsasa_amd.so"))):
        libs[os.path.basename(os.path.dirname(p))] = p
    res = {k: [] for k in libs}
    for _ in range(rounds):
        for name, path in libs.items():
            p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, path, args)], capture_output=True, text=True,
                               cwd=ROOT)
            if p.returncode != 0:
                print(name, "FAILED", p.stderr[-400:])
                continue
            d = json.loads(p.stdout.strip().split("\n")[-1])
            one = d.get("one_at_a_time") or {}
            res[name].append((d["kernel_ms"]["occlusion"], d["ms_per_step"], d["kernel_ms"]["grid_build"],
                              one.get("grid_build_kernel_ms", float("nan")), one.get("ms_per_step", float("nan"))))
    for name, v in res.items():
        if v:
            print(f"{name:24s} occlusion ms min {min(x[0] for x in v):.4f}  all {[round(x[0], 3) for x in v]}  step ms min {min(x[1] for x in v):.4f}  grid ms min {min(x[2] for x in v):.4f}"
                  f"  alone: grid ms min {min(x[3] for x in v):.4f} step ms min {min(x[4] for x in v):.4f}")


if __name__ == "__main__":
    main()
