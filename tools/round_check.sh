#!/bin/bash
# quick round check on the GPU box: GPU tests, a bench line, a --no-probe kernel trace for gap_check (a REPLAYED step)
out=$GRAFT_REPO_ROOT/gpurun_out/check_${1:-a}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -5 $out/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --probe-detail $out/ops_by_shape.txt > $out/bench.log 2>&1
tail -1 $out/bench.log > $out/bench_line.json
python - <<PY
import json
d=json.load(open('$out/bench_line.json'))
print({k:d.get(k) for k in ('value','ms_per_step')}, d.get('trained_offsets',{}).get('value'), d.get('inference',{}).get('value'), d.get('fp32',{}).get('value'))
print(d['roofline']['entry_points_ms_per_step'])
PY
rm -rf $out/kt
timeout 600 rocprofv3 --kernel-trace -d $out/kt -o p -- python bench.py --no-cpu-baseline --no-inference --no-extras --no-probe --steps 10 --warmup 3 > $out/kt.log 2>&1
db=$(ls $out/kt/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python tools/gap_check.py $db > $out/gap_check.txt 2>&1
[ -n "$db" ] && python tools/rocpd_stats.py $db 40 > $out/replay_kernel_stats.txt
head -12 $out/gap_check.txt
rm -rf $out/kt
