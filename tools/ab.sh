#!/bin/bash
# interleaved A/B of the default build against an environment switch: tools/ab.sh "CN_DISABLE_X=1" [pairs] [steps]
sw="$1"; pairs=${2:-3}; steps=${3:-40}
for i in $(seq $pairs); do
  a=$(python bench.py --no-cpu-baseline --no-extras --no-inference --no-probe --steps $steps --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(env $sw python bench.py --no-cpu-baseline --no-extras --no-inference --no-probe --steps $steps --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "default $a ms   |   $sw $b ms"
done
