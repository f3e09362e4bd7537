#!/bin/bash
# Timing ablation of k_occlusion_v3: RSASA_DEBUG_STOP=n skips the stages after point n
# (results are wrong in those runs; only the kernel time and instruction counts are read).
# The switch exists only in the ablation library: build it first with
#   make -C rustsasa_amd/csrc ablate            (-> rustsasa_amd/lib/variants/ablate/)
export RSASA_TUNING=1  # (the library reads its RSASA_* measurement switches only then)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIB=$PWD/rustsasa_amd/lib/variants/ablate/librustsasa_amd.so
RUN="import sys; sys.path.insert(0, '.'); import rustsasa_amd._capi as c; c.LIB_PATH = '$LIB'; import bench; bench.main()"
for s in 1 6 2 3 4 5 0; do
  export RSASA_DEBUG_STOP=$s RSASA_OCCLUSION_KERNEL=3
  ms=$(python3 -c "$RUN" --steps 5 --warmup 1 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms']['occlusion'])")
  echo "stop=$s occlusion_ms=$ms"
done
