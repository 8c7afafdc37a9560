#!/bin/bash
# wave-state breakdown of the four DCN kernels at 64->64 @128^2 (isolated launches): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_ws
DCN_SHAPES=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS -d gpurun_out/pmc_ws -o p -- python tools/opbench.py dcn > gpurun_out/pmc_ws.log 2>&1
python - <<PY
import sqlite3, glob
from collections import defaultdict
c = sqlite3.connect(glob.glob('gpurun_out/pmc_ws/*.db')[0])
for pat in ('dcn_fwd_b2_kernel', 'dcn_wgrad_bm_kernel', 'dcn_dx_bm_kernel', 'dcn_dom_bm_kernel'):
    rows = c.execute("select dispatch_id, counter_name, sum(value), max(end - start) from counters_collection where kernel_name like ? group by dispatch_id, counter_name", ('%' + pat + '%',)).fetchall()
    d = defaultdict(dict); dur = {}
    for did, cn, v, t in rows: d[did][cn] = v; dur[did] = t
    if not d: print(pat, 'no dispatch'); continue
    n = len(d); avg = defaultdict(float)
    for m in d.values():
        for k, v in m.items(): avg[k] += v / n
    wc = avg['SQ_WAVE_CYCLES']
    print(f"{pat}: {n} launches, {sum(dur.values()) / n / 1e3:.0f} us; share of wave cycles: waiting (s_waitcnt / barrier) {avg['SQ_WAIT_ANY'] / wc:.2f}, issue-stalled {avg['SQ_WAIT_INST_ANY'] / wc:.2f}, "
          f"issuing {avg['SQ_ACTIVE_INST_ANY'] / wc:.2f} (VALU {avg['SQ_ACTIVE_INST_VALU'] / wc:.2f}, LDS {avg['SQ_ACTIVE_INST_LDS'] / wc:.2f}); SALU insts {avg['SQ_INSTS_SALU']:.3g}, LDS insts {avg['SQ_INSTS_LDS']:.3g}, wave cycles {wc:.3g}")
PY
rm -rf gpurun_out/pmc_ws
