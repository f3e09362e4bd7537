#!/bin/bash
# usage (GPU box): tools/pmc_occupancy.sh <tag> [variant]   -> gpurun_out/<tag>/occupancy.txt
# Wave-slot utilisation of the occlusion kernel's bench dispatch: SQ_WAVE_CYCLES (quad-cycles summed over waves) x 4 against
# (GRBM_GUI_ACTIVE / 8) x 1024 SIMDs x the waves per SIMD the registers allow, and SQ_LEVEL_WAVES / SQ_BUSY_CU_CYCLES.
tag=${1:-occ}; variant=$2
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="--steps 1 --warmup 0 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --config5-steps 0 --files 0 --per-call-seconds 0 --real-steps 0 --hashed-ids-steps 0"
if [ -n "$variant" ]; then
  code="import sys; sys.path.insert(0, '.'); import rustsasa_amd._capi as c; c.LIB_PATH = 'rustsasa_amd/lib/variants/$variant/librustsasa_amd.so'; import bench; sys.argv = ['bench.py'] + '$args'.split(); bench.main()"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_o -- python3 -c "$code" > $out/pmc_o.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_g -- python3 -c "$code" > $out/pmc_g.log 2>&1
else
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_o -- python3 bench.py $args > $out/pmc_o.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_g -- python3 bench.py $args > $out/pmc_g.log 2>&1
fi
python3 tools/pmc_summary.py "$out/pmc_*/**/*counter_collection.csv" > $out/occupancy.txt 2>&1
rm -rf $out/pmc_o $out/pmc_g
cat $out/occupancy.txt
