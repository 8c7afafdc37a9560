for rep in 1 2; do
for b in 384 320 352 416 448; do
CN_WGRAD_BLOCKS=$b python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --probe-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocks $b', d['ms_per_step'])"
done; done
