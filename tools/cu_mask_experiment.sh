#!/bin/bash
# usage (GPU box): tools/cu_mask_experiment.sh   -> gpurun_out/cu_mask.txt
# RSASA_GRID_CUS=N sets N compute units aside for the grid builds (a CU-masked stream) and masks the launch streams to
# the others: does hiding the grid build behind the occlusion kernel pay for the CUs that kernel loses?  
export RSASA_TUNING=1  # (the library reads its RSASA_* measurement switches only then)
out=gpurun_out/cu_mask.txt
: > $out
args="--steps 40 --warmup 5 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --config5-steps 0 --files 0 --per-call-seconds 0 --real-steps 0 --hashed-ids-steps 0"
run() {  # label, env assignments..., -- bench args
    label=$1; shift
    line=$(env "$@" python3 bench.py $args $EXTRA 2>/dev/null | tail -1)
    echo "$line" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-34s occlusion %.4f ms  grid build %.4f ms  step %.4f ms (min %.4f median %.4f)' % ('$label', d['kernel_ms']['occlusion'], d['kernel_ms']['grid_build'], d['ms_per_step'], d.get('ms_per_step_min', 0), d.get('ms_per_step_median', 0)))" >> $out
}
for EXTRA in "" "--shard-of 8"; do
    echo "== bench.py $args $EXTRA" >> $out
    run "no reserved CUs" RSASA_NUMA=1
    run "8 reserved, mask bits 0..7" RSASA_GRID_CUS=8 RSASA_GRID_CU_STRIDE=1
    run "8 reserved, every 32nd bit" RSASA_GRID_CUS=8 RSASA_GRID_CU_STRIDE=32
    run "16 reserved, every 16th bit" RSASA_GRID_CUS=16 RSASA_GRID_CU_STRIDE=16
    run "16 reserved, mask bits 0..15" RSASA_GRID_CUS=16 RSASA_GRID_CU_STRIDE=1
    run "32 reserved, every 8th bit" RSASA_GRID_CUS=32 RSASA_GRID_CU_STRIDE=8
done
cat $out
