#!/bin/bash
# Timeline of a STREAM of host batches (rsasa_host_batch_enqueue / _wait): uploads and occlusion kernels from rocprofv3's
# kernel and memory-copy traces, every k_occlusion_mx launch with the idle time before it and what the link did then.
# usage: tools/h2h_stream_timeline.sh [batches] [repetitions]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/h2hs_trace; mkdir -p gpurun_out
export H2H_REPS=${2:-4}
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/h2hs_trace -- python3 tools/bench_h2h_stream.py --api ${1:-12} > gpurun_out/h2hs_trace.log 2>&1
grep "stream API" gpurun_out/h2hs_trace.log
python3 - <<'P'
import csv, glob, re
kt = glob.glob("gpurun_out/h2hs_trace/**/*kernel_trace.csv", recursive=True)
mt = glob.glob("gpurun_out/h2hs_trace/**/*memory_copy_trace.csv", recursive=True)
if not kt or not mt:
    raise SystemExit("no trace files (the profiler died before it wrote them)")
ev = []
for r in csv.DictReader(open(kt[0])):
    if "rsasa" in r["Kernel_Name"]:
        m = re.search(r"k_\w+", r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", m.group(0) if m else r["Kernel_Name"][:24], r.get("Stream_Id", "?")))
for r in csv.DictReader(open(mt[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", "?"))[:30], r.get("Stream_Id", "?")))
ev.sort()
t0 = ev[0][0]
def union(iv, a, b):
    iv = sorted((max(s, a), min(e, b)) for s, e in iv if e > a and s < b)
    tot = 0; cur = None
    for s, e in iv:
        if cur is None: cur = [s, e]
        elif s > cur[1]: tot += cur[1] - cur[0]; cur = [s, e]
        else: cur[1] = max(cur[1], e)
    return tot + (cur[1] - cur[0] if cur else 0)
h2d = [(e[0], e[1]) for e in ev if e[2] == "C" and "HOST_TO_DEVICE" in e[3].upper()]
prev_end = None
for s, e, kind, name, st in ev:
    if kind == "K" and "occlusion_mx" in name:
        if prev_end is not None:
            gap = s - prev_end
            busy = union(h2d, prev_end, s) if gap > 0 else 0
            print(f"{(s-t0)/1e3:10.1f} us  k_occlusion_mx {(e-s)/1e3:7.1f} us  stream {st:>3}  idle before {gap/1e3:8.1f} us, uploads busy in it {busy/1e3:8.1f} us" + ("   <<<<" if gap > 1500e3 else ""))
        prev_end = max(prev_end or 0, e)
# upload rate while kernels run / do not run: the big copies
big = [(s, e) for s, e in h2d if e - s > 100e3]
print(f"{len(h2d)} uploads, {len(big)} longer than 100 us: median {sorted(e - s for s, e in big)[len(big)//2]/1e3:.1f} us")
P
tail -3 gpurun_out/h2hs_trace.log
