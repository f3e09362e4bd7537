#!/bin/bash
# usage (GPU box): tools/pmc_round5.sh <tag> [bench.py args]   -> gpurun_out/<tag>/pmc5.txt
# The counters the round-4 review asked for, for the occlusion kernel's bench dispatch (separate --pmc passes, no tracing):
# scalar-unit cycles, LDS issue stalls and bank conflicts, instruction-cache requests / hits / misses and fetches, beside the
# instruction mix, the wait / active split and the matrix pipe's busy and co-execution cycles of the SAME build.
tag=${1:-pmc5}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
shift
one="python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --config5-steps 0 --files 0 --per-call-seconds 0 --real-steps 0 --hashed-ids-steps 0 $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $out/pmc_a -- $one > $out/pmc_a.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_b -- $one > $out/pmc_b.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC --output-format csv -d $out/pmc_c -- $one > $out/pmc_c.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_LDS SQ_CYCLES --output-format csv -d $out/pmc_d -- $one > $out/pmc_d.log 2>&1
rocprofv3 --pmc SQ_WAVES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES --output-format csv -d $out/pmc_e -- $one > $out/pmc_e.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $out/pmc_g -- $one > $out/pmc_g.log 2>&1
python3 tools/pmc_summary.py "$out/pmc_*/**/*counter_collection.csv" > $out/pmc5.txt 2>&1
for p in a b c d e g; do tail -2 $out/pmc_$p.log | grep -i "error\|invalid\|unknown" ; done
rm -rf $out/pmc_a $out/pmc_b $out/pmc_c $out/pmc_d $out/pmc_e $out/pmc_g
cat $out/pmc5.txt
