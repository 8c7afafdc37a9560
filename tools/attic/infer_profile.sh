#!/bin/bash
# kernel-level breakdown of an inference batch (eval forward + decode, hipGraph): rocprofv3 kernel trace of tools/attic/infer_bench.py
out=$GRAFT_REPO_ROOT/gpurun_out/infer_prof
mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/attic/infer_bench.py --steps 30 | tail -1
rm -rf $out/kt
timeout 600 rocprofv3 --kernel-trace -d $out/kt -o p -- python tools/attic/infer_bench.py --steps 20 > $out/kt.log 2>&1
db=$(ls $out/kt/*.db | head -1)
python - <<PY > $out/infer_kernels.txt
import sqlite3, re
c = sqlite3.connect("$db")
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# the last replay: from the last stem kernel to the end
stem = [i for i, r in enumerate(rows) if "stem7_" in r[0]]
seg = rows[stem[-1]:]
t0 = seg[0][1]
print(f"last replayed batch: {len(seg)} kernels, span {(seg[-1][2] - t0) / 1e6:.3f} ms")
agg = {}
for n, s, e in seg:
    k = re.sub(r"\(.*", "", n)[:80]
    d = agg.setdefault(k, [0, 0]); d[0] += 1; d[1] += e - s
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t / 1e3:9.1f} us {n:4d}  {k}")
print("---- in order ----")
for n, s, e in seg:
    nm = re.sub(r"\(.*", "", n)[:90]
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f}  {nm}")
PY
head -45 $out/infer_kernels.txt
rm -rf $out/kt
