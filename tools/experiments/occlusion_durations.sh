#!/bin/bash
# Every k_occlusion_mx launch of a profiled bench run: its duration, the gap before it, and the kernels that start while it runs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/occ_trace; mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/occ_trace -- python3 bench.py --steps 30 --warmup 3 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --config5-steps 0 --files 0 --per-call-seconds 0 --real-steps 0 --hashed-ids-steps 0 > gpurun_out/occ_trace.log 2>&1
python3 - <<'P'
import csv, glob, re
kt = glob.glob("gpurun_out/occ_trace/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(kt)):
    if "rsasa" in r["Kernel_Name"]:
        m = re.search(r"k_\w+", r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:24]))
ev.sort()
occ = [e for e in ev if "occlusion_mx" in e[2] and e[1] - e[0] > 1_000_000]
prev = None
for s, e, n in occ:
    inside = [(n2, (s2 - s) / 1e3, (e2 - s2) / 1e3) for s2, e2, n2 in ev if s < s2 < e and n2 != n]
    print(f"occlusion {(e - s) / 1e3:8.1f} us  gap before {((s - prev) / 1e3 if prev else 0):7.1f}  started inside: " + ", ".join(f"{a}@{b:.0f}({c:.0f})" for a, b, c in inside))
    prev = e
P
