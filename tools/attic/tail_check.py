"""From a rocprofv3 kernel-trace db of bench.py: for the last replayed step, when does each hardware queue finish relative to the
optimizer kernel (adam_kernel)?  Shows whether the side stream (weight gradients) or the main chain ends the step."""
import sqlite3, sys, glob
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else [x for x in cols if "queue" in x][0]
rows = c.execute(f"select name, start, end, {qcol} from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
i1, i0 = adam[-1], adam[-2]
seg = rows[i0 + 1:i1 + 1]
t_adam = rows[i1][1]
last = {}
busy = {}
for n, s, e, q in seg[:-1]:
    last[q] = (e, n)
    busy[q] = busy.get(q, 0) + (e - s)
print("step span ms:", (rows[i1][2] - rows[i0][2]) / 1e6)
for q, (e, n) in sorted(last.items(), key=lambda kv: kv[1][0]):
    print(f"queue {q}: last kernel ends {(t_adam - e) / 1e3:8.1f} us before adam starts; busy {busy[q] / 1e6:6.2f} ms; last = {n[:60]}")

print("last kernels before the optimizer:")
for n, s_, e, q in seg[-9:-1]:
    print(f"  q{q} start {-(t_adam - s_) / 1e3:9.1f} us  dur {(e - s_) / 1e3:8.1f} us  {n[:70]}")
