# usage: ab_env.sh "<VAR=val[,VAR=val] | -> ..." [reps] : interleaved short bench runs, each under its own set of env assignments
reps=${2:-2}
for i in $(seq $reps); do for c in $1; do
( if [ "$c" != "-" ]; then for kv in ${c//,/ }; do export $kv; done; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-extras --no-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['ms_per_step'])" )
done; done
