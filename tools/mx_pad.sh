#!/bin/bash
# usage (GPU box, from the repo root): tools/mx_pad.sh [rounds]
# The PAD method: what one more unit of a stage's work costs the LAUNCH beside everything else it runs - the
# number a telescoping ablation cannot give.  Diagnostic builds of k_occlusion_mx under lib/variants/ (same results as the
# shipped library, never shipped), timed against it by tools/ab_bench.py, interleaved on one box:
#   padg1, padg2   -DMX_PAD_GROUP=1|2   a group's prologue (run look-ups, scan, union gathers) done once / twice more per group
#   padv20         -DMX_PAD_VALU=20     20 more vector instructions per atom
#   pads20         -DMX_PAD_SALU=20     20 more scalar instructions per atom
#   sleep2         -DMX_SLEEP=2         128 cycles of pure latency per atom (s_sleep: no unit used)
# Build them first, in the build container:
#   for v in "padg1 -DMX_PAD_GROUP=1" "padg2 -DMX_PAD_GROUP=2" "padv20 -DMX_PAD_VALU=20" "pads20 -DMX_PAD_SALU=20" "sleep2 -DMX_SLEEP=2"; do set -- $v; \
#     make -C rustsasa_amd/csrc OUT=../lib/variants/$1/librustsasa_amd.so EXTRA=$2 ../lib/variants/$1/librustsasa_amd.so; done
rounds=${1:-3}
python3 tools/ab_bench.py --rounds $rounds 2>&1 | tee gpurun_out/mx_pad.txt
