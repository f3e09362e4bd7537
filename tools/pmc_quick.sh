#!/bin/bash
# usage (GPU box): tools/pmc_quick.sh <tag> [bench.py args, e.g. --workload uniform1m]   -> gpurun_out/<tag>/pmc.txt : instruction mix and wait/active cycles of the occlusion kernel
tag=${1:-pmcq}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
shift
one="python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --per-call-seconds 0 --hashed-ids-steps 0 $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $out/pmc_a -- $one > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_b -- $one > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $out/pmc_c -- $one > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD --output-format csv -d $out/pmc_d -- $one > /dev/null 2>&1
python3 tools/pmc_summary.py "$out/pmc_*/**/*counter_collection.csv" > $out/pmc.txt 2>&1
rm -rf $out/pmc_a $out/pmc_b $out/pmc_c $out/pmc_d
cat $out/pmc.txt
