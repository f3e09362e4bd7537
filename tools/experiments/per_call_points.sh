# the per-structure call (k_occlusion_fast: batches below 32 768 atoms) at 100 points (4 remainder points with W = 8) against 104 (none)
PER_CALL_BIN=/tmp/pc.bin python tools/bench_per_call.py 1 > /dev/null
for r in 1 2; do for n in 100 104; do echo "points $n"; rustsasa_amd/lib/bench_per_call /tmp/pc.bin $n 1 1 c16 c64 2>&1 | cut -c1-150; done; done
