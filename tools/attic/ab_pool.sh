for i in 1 2 3; do
CN_DLA_POOL_TWICE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --probe-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('twice ', d['ms_per_step'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --probe-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shared', d['ms_per_step'])"
done
