#!/bin/bash
# Kernel timeline of the timed region's stepping (two batches in flight): every kernel of three steps in the middle of the
# run with its queue, start and end - where a batch's grid build runs relative to the neighbour's occlusion kernel.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/steps_trace; mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/steps_trace -- python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --config5-steps 0 --files 0 --per-call-seconds 0 --real-steps 0 --hashed-ids-steps 0 > gpurun_out/steps_trace.log 2>&1
python3 - <<'P'
import csv, glob, re
kt = glob.glob("gpurun_out/steps_trace/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(kt)):
    if "rsasa" in r["Kernel_Name"]:
        m = re.search(r"k_\w+", r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:24], r.get("Queue_Id", "?")))
ev.sort()
occ = [e for e in ev if "occlusion_mx" in e[2] and e[1] - e[0] > 1_000_000]
t0 = occ[len(occ) // 2][0]
t1 = occ[len(occ) // 2 + 3][0]
prev_occ_end = None
for s, e, name, q in ev:
    if s < t0 - 400_000 or s > t1:
        continue
    note = ""
    if "occlusion_mx" in name and e - s > 1_000_000:
        if prev_occ_end:
            note = f"   idle since the previous occlusion kernel {(s - prev_occ_end) / 1e3:7.1f} us"
        prev_occ_end = e
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  queue {q:>2}  {name}{note}")
P
tail -1 gpurun_out/steps_trace.log | cut -c1-200
