# usage: ab_prio.sh "<lib:blocks> ..." [reps] : interleaved bench over (library build, CN_WGRAD_BLOCKS) pairs; blocks "-" = default
reps=${2:-2}
for i in $(seq $reps); do
for c in $1; do
lib=${c%%:*}; blk=${c##*:}
if [ "$blk" = "-" ]; then unset CN_WGRAD_BLOCKS; else export CN_WGRAD_BLOCKS=$blk; fi
CN_LIB_PATH=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --probe-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c'.split('/')[-1], d['ms_per_step'], d['trained_offsets']['ms_per_step'])"
done; done
