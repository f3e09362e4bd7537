export RSASA_TUNING=1  # (the library reads its RSASA_* measurement switches only then)
for apw in "$@"; do
  RSASA_ATOMS_PER_WAVE=$apw timeout 200 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('apw', $apw, d['kernel_ms'], d['value'])"
done
