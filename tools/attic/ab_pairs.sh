#!/bin/bash
# usage: ab_pairs.sh "A=1,B=2 A=3,B=4 ..." [reps]  — interleaved short bench runs, each with the comma-separated VAR=value settings
# of one item exported ("-" = nothing set)
reps=${2:-2}
for r in $(seq $reps); do for item in $1; do
  ( if [ "$item" != "-" ]; then for kv in ${item//,/ }; do export $kv; done; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-inference --no-probe 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$item', d['value'], d['ms_per_step'])" )
done; done
