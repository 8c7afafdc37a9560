#!/bin/bash
# interleaved A/B of CN_WGRAD_BLOCKS values on the trained-offsets regime: ab_trained.sh "128 160" [reps]
for r in $(seq ${2:-2}); do for v in $1; do
  CN_WGRAD_BLOCKS=$v python bench.py --dcn-offsets trained --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-inference --no-probe 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('trained CN_WGRAD_BLOCKS=$v', d['value'], d['ms_per_step'])"
done; done
