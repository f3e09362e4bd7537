#!/bin/bash
# Kernel timeline of a STREAM of host batches (rsasa_host_batch_enqueue / _wait), several repetitions: every
# k_occlusion_mx launch with the idle time before it, so that a slow repetition can be told from a fast one.
# (kernel trace only: rocprofv3's memory-copy trace crashes at exit with the library's worker threads)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/h2hs_k; mkdir -p gpurun_out
export H2H_REPS=${2:-6}
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/h2hs_k -- python3 tools/bench_h2h_stream.py --api ${1:-8} > gpurun_out/h2hs_k.log 2>&1
grep "stream API" gpurun_out/h2hs_k.log
python3 - <<'P'
import csv, glob, re
kt = glob.glob("gpurun_out/h2hs_k/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(kt)):
    if "rsasa" in r["Kernel_Name"]:
        m = re.search(r"k_\w+", r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:24], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
ev.sort()
t0 = ev[0][0]
prev_end = None; prev_any = None
for s, e, name, q, st in ev:
    if "occlusion_mx" in name:
        gap = (s - prev_end) / 1e3 if prev_end else 0
        # kernels of other kinds that ran in the gap
        print(f"{(s-t0)/1e3:10.1f} us  k_occlusion_mx {(e-s)/1e3:7.1f} us  queue {q:>3} stream {st:>3}  idle before (since the previous occlusion) {gap:8.1f} us" + ("   <<<<" if gap > 1500 else ""))
        prev_end = e
P
