#!/bin/bash
# usage (GPU box): tools/mx_ablate.sh [rounds]   -> gpurun_out/mx_ablate.txt
# What each stage of k_occlusion_mx costs the LAUNCH: the kernel's time with the atom loop cut short behind a stage
# (-DMX_ABLATE=n builds under lib/variants/abl1..abl5, wrong results by design) against the full build, interleaved on
# one box by tools/ab_bench.py.  Differences of consecutive levels are the stages' throughput costs; the stamped build
# (tools/mx_stage_prof.py) gives their latencies.
#   for n in 1 2 3 4 5; do make -C rustsasa_amd/csrc OUT=../lib/variants/abl$n/librustsasa_amd.so EXTRA=-DMX_ABLATE=$n ../lib/variants/abl$n/librustsasa_amd.so; done
rounds=${1:-3}
python3 tools/ab_bench.py --rounds $rounds --steps 10 --warmup 2 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --config5-steps 0 --files 0 --per-call-seconds 0 --real-steps 0 --hashed-ids-steps 0 2>&1 | tee gpurun_out/mx_ablate.txt
