#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# Everything profiles/ holds for one build, written under gpurun_out/<tag>/:
#   bench.json                 default `python bench.py` line (with the CPU baseline leg)
#   kernel_stats.csv           rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 3 --cpu-seconds 0`
#   bench_under_rocprof.json   the bench line of that profiled run
#   pmc.txt                    PMC summaries of the occlusion kernel (separate passes, no tracing): instruction mix, waits,
#                              matrix-pipe busy / co-execution cycles, scalar-unit cycles, LDS issue stalls and bank conflicts,
#                              instruction-cache requests / hits / misses, GRBM_GUI_ACTIVE (clock), FETCH_SIZE, WRITE_SIZE, TCC hits
#   microbench_clock.txt       tools/microbench_clock (instruction prices with observed residency); h2h_stream.txt: a stream of
#                              host batches against one call after the other
#   pmc_uniform1m.txt          the same instruction-mix, matrix-pipe and GRBM_GUI_ACTIVE passes for the 1M-atom / 960-point dispatch
#   bench_uniform1m.json, single_and_pcie.json, files_mode.json, files_mode_1500.json, files_mode_cif.json, bench_shard_of_8.json,
#   two_in_flight.txt, bench_run2.json (a second default run at the end), per_call_combined.txt (the drop-in call alone and with call
#   combining), files_end_to_end.json / _cif.json (4 363 files in -> 4 363 JSON files out)
tag=${1:-round}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > $out/bench.log 2>&1; tail -1 $out/bench.log > $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --config5-steps 0 --files 0 --per-call-seconds 0 --real-steps 0 --hashed-ids-steps 0 > $out/trace.log 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
grep '^{"metric"' $out/trace.log | tail -1 > $out/bench_under_rocprof.json
one="python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --config5-steps 0 --files 0 --per-call-seconds 0 --real-steps 0 --hashed-ids-steps 0"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $out/pmc_a -- $one > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_b -- $one > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $out/pmc_c -- $one > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_LDS SQ_CYCLES --output-format csv -d $out/pmc_d -- $one > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES --output-format csv -d $out/pmc_e -- $one > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_g -- $one > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $one > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $one > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_tcc -- $one > /dev/null 2>&1
python3 tools/pmc_summary.py "$out/pmc_*/**/*counter_collection.csv" > $out/pmc.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d $out/pmcu_a -- python3 bench.py --workload uniform1m --steps 1 --warmup 0 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --hashed-ids-steps 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $out/pmcu_c -- python3 bench.py --workload uniform1m --steps 1 --warmup 0 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --hashed-ids-steps 0 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/pmcu_g -- python3 bench.py --workload uniform1m --steps 1 --warmup 0 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --hashed-ids-steps 0 > /dev/null 2>&1
python3 tools/pmc_summary.py "$out/pmcu_*/**/*counter_collection.csv" > $out/pmc_uniform1m.txt 2>&1
rm -rf $out/pmcu_a $out/pmcu_c $out/pmcu_g
python3 bench.py --workload uniform1m --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --hashed-ids-steps 0 > $out/u1m.log 2>&1; tail -1 $out/u1m.log > $out/bench_uniform1m.json
python3 tools/bench_single.py 2>/dev/null > $out/single_and_pcie.json
python3 tools/bench_files.py --files 4363 --repeat 3 --calls 4 > $out/files.log 2>&1; tail -1 $out/files.log > $out/files_mode.json
python3 tools/bench_files.py --files 1500 --repeat 3 --calls 4 > $out/files1500.log 2>&1; tail -1 $out/files1500.log > $out/files_mode_1500.json
python3 tools/bench_files.py --files 4363 --repeat 3 --calls 4 --format cif > $out/files_cif.log 2>&1; tail -1 $out/files_cif.log > $out/files_mode_cif.json
python3 bench.py --shard-of 8 > $out/shard8.log 2>&1; tail -1 $out/shard8.log > $out/bench_shard_of_8.json
python3 tools/bench_two_in_flight.py 1 8 2>&1 | grep "^shard" > $out/two_in_flight.txt
python3 bench.py > $out/bench2.log 2>&1; tail -1 $out/bench2.log > $out/bench_run2.json
tools/microbench_clock > $out/microbench_clock.txt 2>&1
H2H_REPS=5 python3 tools/bench_h2h_stream.py --api 12 2>/dev/null | tail -3 > $out/h2h_stream.txt
H2H_SORTED=1 python3 tools/bench_h2h.py 8 2>/dev/null | tail -1 >> $out/h2h_stream.txt
python3 tools/h2h_stream_trace.py 12 3 2> $out/h2h_trace.log > /dev/null; awk '/==== round 2/,0' $out/h2h_trace.log | grep "device:\|====" > $out/h2h_stream_trace.txt
RSASA_TUNING=1 RSASA_COMBINE_TRACE=1 python3 tools/bench_per_call.py 1 1 16 c16 c64 s64 2>&1 | grep -v '^ *$' > $out/per_call_combined.txt
python3 tools/bench_files.py --end-to-end --files 4363 --repeat 3 > $out/e2e.log 2>&1; tail -1 $out/e2e.log > $out/files_end_to_end.json
python3 tools/bench_files.py --end-to-end --files 4363 --repeat 3 --format cif > $out/e2e_cif.log 2>&1; tail -1 $out/e2e_cif.log > $out/files_end_to_end_cif.json
rm -rf $out/trace $out/pmc_a $out/pmc_b $out/pmc_c $out/pmc_d $out/pmc_e $out/pmc_g $out/pmc_fetch $out/pmc_write $out/pmc_tcc
ls -la $out; cat $out/bench.json; cat $out/bench_under_rocprof.json; head -4 $out/kernel_stats.csv | cut -c1-160; cat $out/pmc.txt
