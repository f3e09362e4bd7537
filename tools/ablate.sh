#!/bin/bash
# Timing ablation of k_occlusion_v3: RSASA_DEBUG_STOP=n skips the stages after point n
# (results are wrong in those runs; only the kernel time and instruction counts are read).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in 1 6 2 3 4 5 0; do
  export RSASA_DEBUG_STOP=$s
  ms=$(python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms']['occlusion'])")
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/abl_$s -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2>&1
  echo "stop=$s occlusion_ms=$ms $(python3 tools/pmc_summary.py "gpurun_out/abl_$s/**/*counter_collection.csv" | grep -E 'INSTS' | awk '{printf "%s=%s ", $1, $NF}')"
done
