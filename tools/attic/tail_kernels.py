"""From a rocprofv3 --kernel-trace db of `bench.py --no-probe`: the side-stream (weight-gradient) kernels of the last replayed step that
START after the launch stream's last kernel has ended — the step's tail — with their grids.   python tools/attic/tail_kernels.py <db>"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if r[0].startswith("adam_kernel")]
seg = rows[adam[-2] + 1:adam[-1] + 1]
t0 = min(r[1] for r in seg)
side = lambda n: any(k in n for k in ("wgrad", "colsum", "dwdeconv_bwd_weight", "zero_kernel", "sum_slabs"))
main_end = max(r[2] for r in seg if not side(r[0]) and "adam" not in r[0] and "topk" not in r[0] and "ctdet" not in r[0])
print(f"launch stream ends at +{(main_end - t0) / 1e6:.3f} ms; step ends at +{(max(r[2] for r in seg) - t0) / 1e6:.3f} ms")
short = lambda n: re.sub(r"\(.*", "", n)[:64]
tot = 0
print("side kernels running after that (start ms, duration us, workgroups, name):")
for n, s, e, gx, gy, gz, wx in seg:
    if side(n) and e > main_end:
        tot += e - max(s, main_end)
        print(f"  +{(s - t0) / 1e6:7.3f}  {(e - s) / 1e3:8.1f}  {gx * gy * gz // max(wx, 1):6d}  {short(n)}")
print(f"side-stream time after the launch stream's end: {tot / 1e6:.3f} ms")
# the side stream's backlog over time: kernel time queued before / after the launch stream ended
before = sum(e - s for n, s, e, *_ in seg if side(n) and e <= main_end)
print(f"side-stream kernel time before: {before / 1e6:.3f} ms")
# per-millisecond occupancy of the two classes (fraction of the bin each class has a kernel in flight)
import math
t1 = max(r[2] for r in seg)
nb = int(math.ceil((t1 - t0) / 1e6))
occ = [[0.0, 0.0] for _ in range(nb)]
for n, s, e, *_ in seg:
    k = 1 if side(n) else 0
    b0, b1 = int((s - t0) // 1e6), int((e - t0 - 1) // 1e6)
    for b in range(b0, min(b1, nb - 1) + 1):
        lo, hi = max(s, t0 + b * 1e6), min(e, t0 + (b + 1) * 1e6)
        occ[b][k] += max(0.0, hi - lo) / 1e6
print("ms bin: launch-stream kernel time / side kernel time (ms of kernel time started in flight per ms; >1 = overlapping kernels)")
print("  " + "  ".join(f"{b:2d}:{o[0]:.1f}/{o[1]:.1f}" for b, o in enumerate(occ)))
print("first side-stream kernels of the step (start ms, duration us, workgroups, name) and the launch-stream kernel in flight at that moment:")
sk = [r for r in seg if side(r[0])]
mk = [r for r in seg if not side(r[0])]
for n, s, e, gx, gy, gz, wx in sk[:60]:
    cur = [short(m[0])[:40] for m in mk if m[1] <= s < m[2]]
    print(f"  +{(s - t0) / 1e6:7.3f}  {(e - s) / 1e3:8.1f}  {gx * gy * gz // max(wx, 1):6d}  {short(n):50s} | {cur[:1]}")
print("launch-stream kernels by 1 ms bin (first kernel name starting in the bin):")
seen = set()
for n, s, e, *_ in mk:
    b = int((s - t0) // 1e6)
    if b not in seen and 9 <= b <= 32:
        seen.add(b)
        print(f"  +{b:2d} ms  {short(n)}")
