#!/bin/bash
# SQ counter pass over the bench step for the kernels that dominate it (round-3 VERDICT, evidence item 5b): one rocprofv3 run with
# --kernel-trace --pmc only (no other trace domain), eager launches of the same workload (--no-graph: the counters are per dispatch),
# per kernel template the per-launch averages and the derived ratios.   usage: tools/pmc_sq.sh <tag>
tag=${1:-r06}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_$tag
mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf $out/db
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $out/db -o p -- python bench.py --no-graph --no-cpu-baseline --no-probe --no-inference --no-extras --steps 2 --warmup 1 > $out/run.log 2>&1
python - <<PY > $out/pmc_sq.txt
import sqlite3, glob, re
from collections import defaultdict
dbs = glob.glob('$out/db/*.db')
c = sqlite3.connect(dbs[0])
pats = ['dcn_wgrad_bm_kernel', 'dcn_dx_bm_kernel<2>', 'dcn_dom_bm_kernel<64, true>', 'dcn_fwd_b2_kernel<false>', 'dcn_fwd_b2_kernel<true>', 'dcn_fwd_bm_kernel<2>', 'conv3x3s1_kernel<unsigned short, 128, 64, 8, 1>', 'conv3x3_ws_kernel<64, false, 0, true, 0>',
        'bn_bwd_apply_kernel<unsigned short, true>', 'bn_partial_kernel<unsigned short, 1>', 'wgrad3x3s1_kernel<128, 64, 9, true, 8, 1>', 'conv1x1_stream_kernel<16, 3',
        'conv3x3_c16r_kernel<1, 1, 2>', 'conv3x3_c16r_kernel<1, 1, 0>', 'conv3x3_c16r_kernel<2, 2, 2>', 'stem7_rows_kernel', 'dgrad3x3s2_kernel<64, 64, 8>', 'dcn_fwd_tile_kernel<128>']
print("# rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS")
print("# -- python bench.py --no-graph --no-probe --no-extras --steps 2 --warmup 1   ($tag; eager launches, both streams active; per-launch averages summed over all SEs/CUs)")
print("# MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES (cycles: 32 per 32x32x16 bf16 MFMA, summed over the chip) / (1024 SIMDs x launch time x 2.4 GHz): the matrix-pipe")
print("# utilisation against the nominal clock (blend-matrix MFMAs of the DCN kernels included, so it is above their contraction-only frac_mfma in the bench line);")
print("# VALU per MFMA = SQ_INSTS_VALU / SQ_INSTS_MFMA (wave instructions); LDS conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; LDS issue stall = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES")
for pat in pats:
    rows = c.execute("select dispatch_id, counter_name, sum(value), max(end - start) from counters_collection where kernel_name like ? group by dispatch_id, counter_name", ('%' + pat + '%',)).fetchall()
    d = defaultdict(dict); dur = {}
    for did, cn, v, t in rows:
        d[did][cn] = v; dur[did] = t
    if not d:
        print(f"{pat}: no dispatch"); continue
    n = len(d)
    avg = defaultdict(float)
    for did, m in d.items():
        for k, v in m.items():
            avg[k] += v / n
    us = sum(dur.values()) / n / 1e3
    g = lambda k: avg.get(k, 0.0)
    print(f"{pat}\n    launches {n}, avg {us:.1f} us (under the counter pass); " + ", ".join(f"{k}={g(k):.3g}" for k in sorted(avg)))
    vpm = f"{g('SQ_INSTS_VALU') / g('SQ_INSTS_MFMA'):.1f}" if g('SQ_INSTS_MFMA') else "n/a (no MFMA)"
    print(f"    MFMA pipe busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * us * 1e-6 * 2.4e9):.3f};  VALU per MFMA {vpm};  "
          f"LDS bank-conflict share of LDS cycles {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.3f};  LDS-issue-stall share of wave cycles {g('SQ_WAIT_INST_LDS') / max(g('SQ_WAVE_CYCLES'), 1):.3f}")
PY
cat $out/pmc_sq.txt | head -50
rm -rf $out/db
