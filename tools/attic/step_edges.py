"""From a rocprofv3 --kernel-trace db of `bench.py --no-probe`: the first and the last kernels of the last replayed step (adam_kernel ->
adam_kernel) with start offsets and durations — the two ends of a step are where only ONE stream has work (forward: no weight gradients
yet; tail: the data-gradient chain of the first layers).   python tools/attic/step_edges.py <db> [n_head] [n_tail]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if r[0].startswith("adam_kernel")]
seg = rows[adam[-2] + 1:adam[-1] + 1]
t0, t1 = seg[0][1], max(r[2] for r in seg)
nh = int(sys.argv[2]) if len(sys.argv) > 2 else 25
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 40
short = lambda n: re.sub(r"^void ", "", re.sub(r"\(.*", "", n))[:78]
print(f"step span {(t1 - t0) / 1e6:.3f} ms, {len(seg)} kernels")
print("---- head")
for n, s, e in seg[:nh]:
    print(f"  +{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  {short(n)}")
print("---- tail (offsets from the END of the step)")
for n, s, e in seg[-nt:]:
    print(f"  {(s - t1) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  {short(n)}")
# which kernels feed a stand-alone forward statistics pass (bn_partial_kernel<T, 0>: the producer had no statistics hook)
prod = {}
for i, (n, s, e) in enumerate(seg):
    if "bn_partial_kernel" in n and ", 0>" in n and i > 0:
        k = short(seg[i - 1][0])
        d = prod.setdefault(k, [0, 0.0, 0.0])
        d[0] += 1; d[1] += (e - s) / 1e3
        d[2] += (seg[i + 1][2] - seg[i + 1][1]) / 1e3 if i + 1 < len(seg) and "finalize" in seg[i + 1][0] else 0.0
print("---- producers without a BN statistics hook (launches, us in bn_partial<0>, us in the finalize launch behind it)")
for k, (c_, a, b) in sorted(prod.items(), key=lambda kv: -kv[1][1]):
    print(f"  {c_:3d}  {a:8.1f} us  {b:7.1f} us  {k}")
# short kernels of the step (every launch-stream kernel boundary costs ~3.7 us of serialisation on top of its run time: +212 one-thread
# launches = +0.78 ms, measured): launches under 12 us by name
small = {}
for n, s, e in seg:
    d = (e - s) / 1e3
    if d < 12.0:
        k = short(n)
        v = small.setdefault(k, [0, 0.0])
        v[0] += 1; v[1] += d
print("---- kernels under 12 us (launches per step, total us)")
for k, (c_, a) in sorted(small.items(), key=lambda kv: -kv[1][0]):
    print(f"  {c_:4d}  {a:8.1f} us  {k}")
print(f"  total {sum(v[0] for v in small.values())} launches, {sum(v[1] for v in small.values()):.1f} us")
