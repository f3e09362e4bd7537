#!/bin/bash
# Timing ablation of k_occlusion_fast: builds library variants that stop after the flat map (1),
# the run-start mask (11), the flat positions (12), the sweep's gathers (13),
# the sweep (2), the prep pass (3) or phase A (4) -- results are WRONG in those builds, only the
# kernel time is read -- and times them next to the full kernel with tools/ab_bench.py.
#   local:   tools/ablate_fast.sh build        GPU box:  python tools/ab_bench.py --rounds 2
cd "$(dirname "$0")/../rustsasa_amd/csrc" || exit 1
for v in ${STOPS:-1 11 12 13 2 3 4}; do
  make OUT=../lib/variants/stop_$v/librustsasa_amd.so EXTRA=-DFAST_STOP=$v ../lib/variants/stop_$v/librustsasa_amd.so 2>&1 | grep -E "error" &
done
wait
