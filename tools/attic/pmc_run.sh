#!/bin/bash
# usage: pmc_run.sh <prof_conv mode> <kernel grep> <counter> [<counter> ...]   (one pass, counters only + kernel trace)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mode=$1; pat=$2; shift 2
rm -rf gpurun_out/pmc_x
timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_x -o p -- python tools/attic/prof_conv.py $mode > gpurun_out/pmc_x.log 2>&1
python - <<PY
import sqlite3,glob
from collections import defaultdict
c=sqlite3.connect(glob.glob('gpurun_out/pmc_x/*.db')[0])
rows=c.execute("select dispatch_id,kernel_name,counter_name,value,end-start from counters_collection where kernel_name like '%$pat%' order by dispatch_id").fetchall()
d=defaultdict(dict)
for did,k,cn,v,dur in rows: d[(did,k.split('(')[0][5:60],dur)][cn]=d[(did,k.split('(')[0][5:60],dur)].get(cn,0)+v
for k,v in d.items(): print(k, {a:round(b) for a,b in sorted(v.items())})
PY
