"""Timeline of merged batches (call combining) from a rocprofv3 kernel trace of csrc/host/bench_per_call: per batch - the
kernels k_unpack_atoms -> k_sort_window -> k_occlusion_* on one queue - kernel durations and the gaps between them.
usage (GPU box): PER_CALL_BIN=/tmp/pc.bin python tools/bench_per_call.py 1 c16; rocprofv3 --kernel-trace --output-format csv -d out --
rustsasa_amd/lib/bench_per_call /tmp/pc.bin 100 1 c16; python tools/experiments/combine_timeline.py out/**/*kernel_trace.csv"""
import csv, glob, sys
import numpy as np
rows = []
for f in glob.glob(sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Queue_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort(key=lambda t: (t[0], t[1]))
per_q = {}
for q, s, e, n in rows:
    per_q.setdefault(q, []).append((s, e, n))
gaps = {"unpack->sort": [], "sort->occ": [], "batch->batch": []}
dur = {"unpack": [], "sort": [], "occ_fast": [], "occ_mx": [], "v3": [], "batch span": []}
for q, ks in per_q.items():
    i = 0
    last_end = None
    while i < len(ks):
        if "k_unpack_atoms" not in ks[i][2]:
            i += 1
            continue
        u = ks[i]
        j = i + 1
        if j + 1 >= len(ks) or "k_sort_window" not in ks[j][2]:
            i += 1
            continue
        srt, occ = ks[j], ks[j + 1]
        dur["unpack"].append(u[1] - u[0]); dur["sort"].append(srt[1] - srt[0])
        (dur["occ_mx"] if "k_occlusion_mx" in occ[2] else dur["occ_fast"]).append(occ[1] - occ[0])
        gaps["unpack->sort"].append(srt[0] - u[1]); gaps["sort->occ"].append(occ[0] - srt[1])
        end = occ[1]
        k = j + 2
        while k < len(ks) and "k_unpack_atoms" not in ks[k][2]:
            if "k_occlusion" in ks[k][2]:
                end = ks[k][1]
                if "v3" in ks[k][2]:
                    dur["v3"].append(ks[k][1] - ks[k][0])
            k += 1
        dur["batch span"].append(end - u[0])
        if last_end is not None:
            gaps["batch->batch"].append(u[0] - last_end)
        last_end = end
        i = k
print("queues with batches:", sum(1 for q, ks in per_q.items() if any("k_unpack" in k[2] for k in ks)))
for name, v in list(dur.items()) + list(gaps.items()):
    if v:
        v = np.array(v) / 1e3
        print(f"{name:14s} n {len(v):6d}  median {np.median(v):7.1f} us  mean {v.mean():7.1f}  p90 {np.percentile(v, 90):7.1f}")
