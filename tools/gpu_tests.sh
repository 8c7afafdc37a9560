#!/bin/bash
# all GPU tests (no -x: report every failure)
out=$GRAFT_REPO_ROOT/gpurun_out/tests_${1:-a}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q ${@:2} > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
grep -E "FAILED|ERROR|passed|failed" $out/pytest.log | tail -30
