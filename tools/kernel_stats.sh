#!/bin/bash
# rocprofv3 --kernel-trace --stats of a short bench run (any bench.py arguments), the engine's kernels with calls and
# average / minimum / maximum duration.  usage (GPU box): tools/kernel_stats.sh [bench.py args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kstats; mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --config5-steps 0 --files 0 --per-call-seconds 0 --real-steps 0 "$@" > gpurun_out/kstats.log 2>&1
f=$(find gpurun_out/kstats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("rsasa::(anonymous namespace)::", "").replace("rsasa::", "")
    if n.startswith(("void k_", "k_")):
        print(f'{n[:64]:64s} calls {int(r["Calls"]):4d}  avg {float(r["AverageNs"])/1e3:9.1f} us  min {int(r["MinNs"])/1e3:9.1f}  max {int(r["MaxNs"])/1e3:9.1f}')
P
tail -1 gpurun_out/kstats.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('value', d['value'], 'ms/step', d['ms_per_step'], 'ids_as_hashes', d.get('ids_as_hashes', {}).get('ms_per_step'), d.get('ids_as_hashes', {}).get('ids_dropped_batches'))"
