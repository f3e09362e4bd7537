#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per-dispatch totals
(summed over XCDs/SEs as rocprofv3 reports them), for the largest dispatch of each kernel."""
import collections
import csv
import glob
import sys


def main(paths, match="occlusion"):
    for pat in paths:
        for f in sorted(glob.glob(pat, recursive=True)):
            agg = collections.defaultdict(lambda: collections.defaultdict(float))
            meta = {}
            for r in csv.DictReader(open(f)):
                if match not in r["Kernel_Name"]:
                    continue
                d = r["Dispatch_Id"]
                agg[d][r["Counter_Name"]] += float(r["Counter_Value"])
                meta[d] = (r["Kernel_Name"][:70], r["Grid_Size"], r["VGPR_Count"], r["SGPR_Count"],
                           r["LDS_Block_Size"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            if not agg:
                continue
            # the full-size dispatch that ran longest (a batch with ids launches both instantiations of the matrix-core
            # kernel, and the one the device's id check does not ask for returns at once)
            d = max(agg, key=lambda k: (int(meta[k][1]), meta[k][5], int(k)))
            name, grid, vgpr, sgpr, lds, dur = meta[d]
            print(f"{f}\n  dispatch {d}: {name} grid={grid} vgpr={vgpr} sgpr={sgpr} lds={lds} dur_ns={dur}")
            waves = agg[d].get("SQ_WAVES")
            for k, v in sorted(agg[d].items()):
                per = f"  per-wave {v / waves:10.1f}" if waves else ""
                print(f"    {k:28s} {v:18.0f}{per}")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--match=")]
    m = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--match=")]
    main(args, m[0] if m else "occlusion")
