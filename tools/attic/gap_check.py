"""From a rocprofv3 kernel-trace db of bench.py: idle time between consecutive kernels of every hardware queue in the last
replayed step, and the largest gaps with the kernels on either side.   python tools/attic/gap_check.py <db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else [x for x in cols if "queue" in x][0]
rows = c.execute(f"select name, start, end, {qcol} from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
seg = rows[adam[-2] + 1:adam[-1] + 1]
t0, t1 = seg[0][1], seg[-1][2]
print(f"step span {(t1 - t0) / 1e6:.2f} ms, {len(seg)} kernels")
byq = {}
for n, s, e, q in seg:
    byq.setdefault(q, []).append((s, e, n))
# union busy time over all queues
ev = sorted([(s, 1) for n, s, e, q in seg] + [(e, -1) for n, s, e, q in seg])
busy, depth, last = 0, 0, t0
for t, d in ev:
    if depth > 0:
        busy += t - last
    depth += d
    last = t
print(f"GPU busy (any queue) {busy / 1e6:.2f} ms = {busy / (t1 - t0) * 100:.1f} % of the span")
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    ks.sort()
    run = sum(e - s for s, e, n in ks)
    gaps = [(ks[i + 1][0] - ks[i][1], ks[i][2], ks[i + 1][2]) for i in range(len(ks) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print(f"queue {q}: {len(ks)} kernels, running {run / 1e6:.2f} ms, gaps {sum(g[0] for g in pos) / 1e6:.2f} ms "
          f"({len([g for g in pos if g[0] > 5000])} gaps > 5 us)")
    for g, a, b in sorted(pos, reverse=True)[:6]:
        print(f"    {g / 1e3:8.1f} us   {a[:50]:50s} -> {b[:50]}")
