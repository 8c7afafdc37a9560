#!/bin/bash
# usage: bench_ab.sh VAR "v1 v2 ..." [reps]   — interleaved bench.py runs on one box with env VAR set to each value
var=$1; vals=$2; reps=${3:-2}
for r in $(seq $reps); do for v in $vals; do
  export $var=$v
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$var=$v', d['value'], d['ms_per_step'], {k.split('<')[1]: (v['tflops'], v['ms_per_step']) for k, v in d['roofline']['all_igemm']['variants'].items() if '3x3' in k and '64>' in k})"
done; done
