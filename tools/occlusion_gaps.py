#!/usr/bin/env python3
"""Timeline around the ends of the occlusion kernels of a two-batches-in-flight run, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 12 --warmup 3 --cpu-seconds 0 --h2h-steps 0 --two-steps 0
    tools/occlusion_gaps.py DIR
For each occlusion kernel: its duration, the idle gap before the next one, and every other kernel that overlaps the
interval [end - 0.5 ms, next start] as (name, queue, start, end) in microseconds relative to the kernel's end."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
occ = [k for k in ks if "k_occlusion_mx" in k[2]]
print("kernels", len(ks), "occlusion", len(occ))
def short(n):
    n = n.split("(")[0]
    return n.split("::")[-1][:22]
for i in range(max(1, len(occ) - 4), len(occ)):
    a, b = occ[i - 1], occ[i]
    near = [k for k in ks if k[1] > a[1] - 500_000 and k[0] < b[0] and "k_occlusion_mx" not in k[2]]
    print("occ on queue %s: %.3f ms, gap to the next (queue %s) %.3f ms" % (a[3], (a[1] - a[0]) / 1e6, b[3], (b[0] - a[1]) / 1e6))
    for k in near:
        print("      %-24s q%s  %8.0f .. %8.0f us" % (short(k[2]), k[3], (k[0] - a[1]) / 1e3, (k[1] - a[1]) / 1e3))
