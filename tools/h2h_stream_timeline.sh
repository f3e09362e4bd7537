#!/bin/bash
# Timeline of a STREAM of host batches (rsasa_host_batch_enqueue / _wait): kernels and memory copies of the last ~11 ms
# (two proteome batches) from rocprofv3 traces: is the link busy all the time, are the kernels?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/h2hs_trace
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/h2hs_trace -- python3 tools/bench_h2h_stream.py --api 12 > gpurun_out/h2hs_trace.log 2>&1
python3 - <<'P'
import csv, glob, re
kt = glob.glob("gpurun_out/h2hs_trace/**/*kernel_trace.csv", recursive=True)[0]
mt = glob.glob("gpurun_out/h2hs_trace/**/*memory_copy_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(kt)):
    if "rsasa" in r["Kernel_Name"]:
        m = re.search(r"k_\w+", r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", m.group(0) if m else r["Kernel_Name"][:24]))
for r in csv.DictReader(open(mt)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", "?"))[:30]))
ev.sort()
t_end = ev[-1][1]; w0 = t_end - 16_000_000; w1 = t_end - 5_000_000   # a steady-state window: not the stream's last batch
sel = [e for e in ev if e[0] >= w0 and e[1] <= w1]
t0 = sel[0][0]
def union(iv):
    iv = sorted(iv); tot = 0; cur_s, cur_e = iv[0]
    for s, e in iv[1:]:
        if s > cur_e: tot += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    return tot + cur_e - cur_s
h2d = [(e[0], e[1]) for e in sel if e[2] == "C" and "HOST_TO_DEVICE" in e[3].upper()]
d2h = [(e[0], e[1]) for e in sel if e[2] == "C" and "DEVICE_TO_HOST" in e[3].upper()]
ker = [(e[0], e[1]) for e in sel if e[2] == "K"]
occ = [(e[0], e[1]) for e in sel if e[2] == "K" and "occlusion_mx" in e[3]]
span = (sel[-1][1] - t0) / 1e6
print(f"window {span:.2f} ms: H2D busy {union(h2d)/1e6:.2f} ms in {len(h2d)} copies, D2H busy {union(d2h)/1e6:.2f} ms, some kernel running {union(ker)/1e6:.2f} ms, k_occlusion_mx running {union(occ)/1e6:.2f} ms (sum of its durations {sum(e-s for s,e in occ)/1e6:.2f} ms, {len(occ)} launches)")
prev = None
for e in sel:
    if e[2] == "C" and "HOST_TO_DEVICE" in e[3].upper():
        gap = (e[0] - prev) / 1e3 if prev else 0
        prev = e[1]
        if gap > 40 or (e[1] - e[0]) > 200000: print(f"{(e[0]-t0)/1e3:9.1f} us  H2D {(e[1]-e[0])/1e3:8.1f} us  gap since previous H2D {gap:7.1f}")
    elif e[2] == "K" and "occlusion_mx" in e[3]:
        print(f"{(e[0]-t0)/1e3:9.1f} us  k_occlusion_mx {(e[1]-e[0])/1e3:8.1f} us   ends {(e[1]-t0)/1e3:9.1f}")
P
tail -3 gpurun_out/h2hs_trace.log
