#!/bin/bash
# Timeline of one host-to-host proteome call: kernels and memory copies (rocprofv3 traces), summarised per call.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/h2h_trace
export H2H_SORTED=1  # bench.py's order: largest structures first
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/h2h_trace -- python3 tools/bench_h2h.py ${1:-8} > /dev/null 2>&1
python3 - <<'P'
import csv, glob, re
kt = glob.glob("gpurun_out/h2h_trace/**/*kernel_trace.csv", recursive=True)[0]
mt = glob.glob("gpurun_out/h2h_trace/**/*memory_copy_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(kt)):
    if "rsasa" in r["Kernel_Name"]:
        m = re.search(r"k_\w+", r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", m.group(0) if m else r["Kernel_Name"][:24]))
rows = list(csv.DictReader(open(mt)))
print("copy columns:", list(rows[0].keys()))
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", "?"))[:30]))
ev.sort()
# last full call: find the last k_init_acc groups; take the last 3 ms*... simply the last 9 ms window
t_end = ev[-1][1]; w0 = t_end - 9_000_000
sel = [e for e in ev if e[0] >= w0]
t0 = sel[0][0]
busy_c = sum(e[1] - e[0] for e in sel if e[2] == "C" and "HOST_TO_DEVICE" in e[3].upper().replace("MEMORY_COPY_", "")) / 1e6
print("window ms", (t_end - t0) / 1e6, "H2D busy ms", busy_c)
prev_c = None
for e in sel:
    if e[2] == "C":
        gap = (e[0] - prev_c) / 1e3 if prev_c else 0
        prev_c = e[1]
        print(f"{(e[0]-t0)/1e3:9.1f} us  copy {e[3]:28s} {(e[1]-e[0])/1e3:8.1f} us  gap since previous copy {gap:7.1f}")
    elif e[2] == "K" and (e[1] - e[0]) > 15000:
        print(f"{(e[0]-t0)/1e3:9.1f} us  kernel {e[3]:26s} {(e[1]-e[0])/1e3:8.1f} us   ends {(e[1]-t0)/1e3:9.1f}")
P
