#!/bin/bash
# Per-kernel times of the grid build on the bench batch (rocprofv3 kernel trace), plus two bench lines.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/gp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gp -- python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 "$@" > /dev/null 2>&1
python3 - <<'P'
import csv, glob
f = glob.glob("gpurun_out/gp/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "at::" in r["Name"]: continue
    print(r["Name"][:78].ljust(78), r["Calls"], round(float(r["AverageNs"]) / 1000, 1))
P
for i in 1 2; do python3 bench.py --cpu-seconds 0 --h2h-steps 0 --two-steps 0 "$@" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'])"; done
