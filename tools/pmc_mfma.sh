#!/bin/bash
# usage (GPU box): tools/pmc_mfma.sh <tag> [bench.py args]  -> gpurun_out/<tag>/pmc_mfma.txt
# Matrix-pipe counters of the occlusion kernel on the bench dispatch (one rocprofv3 --pmc pass, no tracing):
# do matrix and vector instructions of different waves of a SIMD execute together (COEXEC) or one after the other?
tag=${1:-pmcm}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
shift
one="python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --per-call-seconds 0 --hashed-ids-steps 0 $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $out/pmc_m -- $one > $out/pmc_m.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $out/pmc_g -- $one > $out/pmc_g.log 2>&1
python3 tools/pmc_summary.py "$out/pmc_*/**/*counter_collection.csv" > $out/pmc_mfma.txt 2>&1
rm -rf $out/pmc_m $out/pmc_g
cat $out/pmc_mfma.txt
