#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_run.sh <tag> [bench args...]
# Collects instruction-mix and wait counters of the occlusion kernel in two rocprofv3 passes.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_${tag}_a -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 "$@" > gpurun_out/pmc_${tag}_a.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_${tag}_b -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 "$@" > gpurun_out/pmc_${tag}_b.log 2>&1
python3 tools/pmc_summary.py "gpurun_out/pmc_${tag}_*/**/*counter_collection.csv"
