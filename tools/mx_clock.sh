#!/bin/bash
# usage (GPU box): tools/mx_clock.sh   (needs the -DMX_CLOCK variant: make OUT=../lib/variants/clock/librustsasa_amd.so EXTRA=-DMX_CLOCK ...)
# In-kernel shader clock of k_occlusion_mx on the bench batch: delta s_memtime / delta s_memrealtime x 100 MHz per wave,
# after a few seconds of back-to-back launches (MI355X guide, DVFS give-back item 6).  Diagnostic build only.
tools/run_variant.sh clock --steps 400 --warmup 5 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 > gpurun_out/mx_clock_raw.txt 2>&1
python3 - <<'PY'
import re, statistics
rows = [tuple(map(int, m.groups())) for m in re.finditer(r"MXCLK block (\d+) cycles (\d+) ticks100MHz (\d+) atoms (\d+)", open("gpurun_out/mx_clock_raw.txt").read())]
late = rows[len(rows) // 2:]   # the second half of the launches: the clock has settled
ghz = [c / t * 0.1 for _, c, t, _ in late if t]
cyc = [c / a for _, c, _, a in late if a == 64]
print("waves stamped", len(rows), "used", len(late))
print("in-kernel clock GHz: median %.3f  min %.3f  max %.3f" % (statistics.median(ghz), min(ghz), max(ghz)))
print("wave cycles per atom (64-atom waves): median %.0f" % statistics.median(cyc))
PY
tail -1 gpurun_out/mx_clock_raw.txt | cut -c1-400
