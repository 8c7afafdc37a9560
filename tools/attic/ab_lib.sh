# usage: ab_lib.sh <old.so> <new.so>  : interleaved bench A/B of two builds
for i in 1 2 3; do
for v in "$1" "$2"; do
CN_LIB_PATH=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --probe-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v'.split('/')[-1], d['ms_per_step'])"
done; done
