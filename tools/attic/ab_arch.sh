#!/bin/bash
# ab_arch.sh ARCH BATCH VAR "v1 v2" [reps] — interleaved bench runs of another architecture with env VAR set to each value
arch=$1; bs=$2; var=$3; vals=$4
for r in $(seq ${5:-2}); do for v in $vals; do
  export $var=$v
  python bench.py --arch $arch --batch $bs --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-inference --no-probe 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$arch $var=$v', d['value'], d['ms_per_step'])"
done; done
