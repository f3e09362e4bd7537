export RSASA_TUNING=1  # (the library reads its RSASA_* measurement switches only then)
for apw in 0 64 48 32 24 16; do
  for sh in 8 4 2; do
    RSASA_ATOMS_PER_WAVE=$apw python bench.py --steps 100 --warmup 5 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --config5-steps 0 --files 0 --per-call-seconds 0 --real-steps 0 --hashed-ids-steps 0 --shard-of $sh | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('apw', $apw, 'shard', $sh, d['kernel_ms']['occlusion'], d['ms_per_step'])"
  done
done
