#!/bin/bash
# quick round check on the GPU box: (optional) GPU tests, a bench line, a --no-probe kernel trace analysed by replay_trace.py
# usage: tools/round_check.sh <tag> [pytest -k expression | "all" | "none"]
out=$GRAFT_REPO_ROOT/gpurun_out/check_${1:-a}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ "${2:-all}" != "none" ]; then
  if [ "${2:-all}" = "all" ]; then timeout 1200 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; else timeout 1200 python -m pytest tests -m gpu -q -k "$2" > $out/pytest.log 2>&1; fi
  echo "pytest rc $?" >> $out/pytest.log
  grep -E "FAILED|ERROR|passed|failed|rc " $out/pytest.log | tail -20
fi
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --probe-detail $out/ops_by_shape.txt > $out/bench.log 2>&1
tail -1 $out/bench.log > $out/bench_line.json
python - <<PY
import json
d=json.load(open('$out/bench_line.json'))
print({k:d.get(k) for k in ('value','ms_per_step')}, 'trained', d.get('trained_offsets',{}).get('value'), 'inference', d.get('inference',{}).get('value'), 'fp32', d.get('fp32',{}).get('value'))
print(d['roofline']['entry_points_ms_per_step'])
PY
rm -rf $out/kt
timeout 600 rocprofv3 --kernel-trace -d $out/kt -o p -- python bench.py --no-cpu-baseline --no-inference --no-extras --no-probe --steps 10 --warmup 3 > $out/kt.log 2>&1
db=$(ls $out/kt/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python tools/replay_trace.py $db > $out/replay_trace.txt 2>&1
[ -n "$db" ] && python tools/rocpd_stats.py $db 200 > $out/replay_kernel_stats.txt
cat $out/replay_trace.txt
rm -rf $out/kt
