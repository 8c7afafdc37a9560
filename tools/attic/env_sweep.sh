#!/bin/bash
# usage: tools/_ab/env_sweep.sh "A=1 B=2" "A=3" ...   each argument = one set of env assignments (use "-" for none); prints ms/step
for round in 1 2; do
for a in "$@"; do
  if [ "$a" = "-" ]; then envs=""; else envs="$a"; fi
  r=$(env $envs python bench.py --no-probe --no-extras --no-inference --no-cpu-baseline --steps 20 --warmup 5 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "round $round [$a] $r"
done
done
