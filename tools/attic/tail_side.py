"""From a rocprofv3 kernel-trace db of bench.py (graph replay, --no-probe): the side-stream kernels of the last step that run after
the launch-stream chain of backward has finished (the step's tail).   python tools/attic/tail_side.py <db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else [x for x in cols if "queue" in x][0]
rows = c.execute(f"select name, start, end, {qcol} from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
seg = rows[adam[-2] + 1:adam[-1] + 1]
t0 = seg[0][1]
byq = {}
for n, s, e, q in seg:
    byq.setdefault(q, []).append((s, e, n))
for q, ks in byq.items():
    print(f"queue {q}: {len(ks)} kernels, first {ks[0][2][:40]} @ {(ks[0][0] - t0) / 1e6:.2f} ms, last {ks[-1][2][:40]} ends @ {(ks[-1][1] - t0) / 1e6:.2f} ms, running {sum(e - s for s, e, n in ks) / 1e6:.2f} ms")
# the backward chain: the queue that holds bn_bwd_apply kernels
bq = max(byq, key=lambda q: sum(1 for s, e, n in byq[q] if "bn_bwd_apply" in n))
ks = sorted(byq[bq])
gaps = [(ks[i + 1][0] - ks[i][1], i) for i in range(len(ks) - 1)]
g, i = max(gaps)
t_end = ks[i][1]
print(f"backward chain (queue {bq}) done at {(t_end - t0) / 1e6:.2f} ms; then a {g / 1e6:.2f} ms wait")
sq = max((q for q in byq if q != bq), key=lambda q: sum(1 for s, e, n in byq[q] if "wgrad" in n))
late = [(s, e, n) for s, e, n in sorted(byq[sq]) if e > t_end]
print(f"side queue {sq}: {len(late)} kernels still to finish, {sum(e - max(s, t_end) for s, e, n in late) / 1e6:.2f} ms")
for s, e, n in late:
    print(f"   {(s - t0) / 1e6:7.2f} ms  {(e - s) / 1e3:8.1f} us  {n[:90]}")
