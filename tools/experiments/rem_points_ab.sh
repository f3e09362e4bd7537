A="--steps 10 --warmup 2 --cpu-seconds 0 --h2h-steps 0 --two-steps 0 --config5-steps 0 --files 0 --per-call-seconds 0 --real-steps 0 --hashed-ids-steps 0"
for r in 1 2 3; do for n in 100 104 96; do python bench.py $A --n-points $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print($n, d['kernel_ms']['occlusion'], d['ms_per_step'])"; done; done
