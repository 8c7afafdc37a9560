#!/bin/bash
# usage: ab_quick.sh VAR "v1 v2 ..." [reps]  — interleaved short bench runs (value + ms/step only) with env VAR set to each value;
# the value "-" leaves VAR unset
var=$1; vals=$2; reps=${3:-2}
for r in $(seq $reps); do for v in $vals; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-inference --no-probe 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$var=$v', d['value'], d['ms_per_step'])"
done; done
