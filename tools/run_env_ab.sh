# usage: tools/run_env_ab.sh VAR v1 v2 ...   (bench.py once per value, interleaved twice)
var=$1; shift
for round in 1 2; do
for v in "$@"; do
  env $var=$v timeout 200 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', d['kernel_ms'], 'step', d['ms_per_step'], d['value'])"
done
done
