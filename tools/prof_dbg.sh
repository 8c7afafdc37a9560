#!/bin/bash
# usage: prof_dbg.sh "<dbg values>" <prof_conv mode> <kernel grep>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in $1; do
  rm -rf gpurun_out/pd_$d
  CN_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pd_$d -o p -- python tools/prof_conv.py $2 > /dev/null 2>&1
  echo "== CN_DBG=$d"
  python - <<PY
import sqlite3,glob
c=sqlite3.connect(glob.glob('gpurun_out/pd_$d/*.db')[0])
rows=c.execute("select name, end-start from kernels where name like '%$3%' order by start").fetchall()
print([ (r[0][5:50].split('(')[0], round(r[1]/1000)) for r in rows])
PY
done
