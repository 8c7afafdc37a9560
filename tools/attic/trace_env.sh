#!/bin/bash
# replay trace under a set of env assignments: tools/attic/trace_env.sh <tag> "A=1 B=2"
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$1
mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf $out/kt
env $2 timeout 600 rocprofv3 --kernel-trace -d $out/kt -o p -- python bench.py --no-cpu-baseline --no-inference --no-extras --no-probe --steps 10 --warmup 3 > $out/kt.log 2>&1
db=$(ls $out/kt/*.db 2>/dev/null | head -1)
REPLAY_DUMP=$out/last_step_kernels.txt python tools/replay_trace.py $db 2 > $out/replay_trace.txt 2>&1
grep -v "^kernels view" $out/replay_trace.txt
rm -rf $out/kt
