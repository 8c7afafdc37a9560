#!/bin/bash
# FETCH_SIZE and WRITE_SIZE of one kernel over a short bench run, in SEPARATE counter-only passes (recipe in profiles/r01_pmc_traffic.json)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
pat=$1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_t
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_t -o p -- python bench.py --no-cpu-baseline --no-probe --steps 3 --warmup 1 > /dev/null 2>&1
  python - <<PY
import sqlite3, glob
c = sqlite3.connect(glob.glob('gpurun_out/pmc_t/*.db')[0])
rows = c.execute("select dispatch_id, sum(value) from counters_collection where kernel_name like '%$pat%' and counter_name='$c' group by dispatch_id").fetchall()
print('$c', 'launches', len(rows), 'avg_kib', sum(r[1] for r in rows) / max(1, len(rows)))
PY
done
rm -rf gpurun_out/pmc_t
