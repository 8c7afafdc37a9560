"""From a rocprofv3 --kernel-trace db of `bench.py --no-probe` (every timed step is a hipGraph REPLAY): for the last complete step
(adam_kernel -> adam_kernel): span, GPU-busy share (union over everything in flight), the idle gaps > 5 us of that union, which
STREAM ends the step and by how much (per-stream first start / last end / summed kernel time), and every kernel that is not ours
(at::native::*, __amd_rocclr_*) with its launches per step.      python tools/replay_trace.py <db> [steps_to_average]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print("kernels view columns:", cols)
scol = next((x for x in ("stream_id", "stream") if x in cols), None)
qcol = "queue_id" if "queue_id" in cols else None
sel = "name, start, end" + (f", {scol}" if scol else ", 0") + (f", {qcol}" if qcol else ", 0")
rows = c.execute(f"select {sel} from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if r[0].startswith("adam_kernel")]
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 5
print(f"{len(adam)} optimizer launches in the trace; analysing the last {nst} replayed steps")
short = lambda n: re.sub(r"\(.*", "", n)[:70]
for k in range(nst, 0, -1):
    seg = rows[adam[-k - 1] + 1:adam[-k] + 1]
    t0, t1 = min(r[1] for r in seg), max(r[2] for r in seg)
    ev = sorted([(r[1], 1) for r in seg] + [(r[2], -1) for r in seg])
    busy, depth, last, gaps = 0, 0, t0, []
    for t, d in ev:
        if depth > 0:
            busy += t - last
        elif t - last > 5000:
            gaps.append((t - last, last))
        depth += d
        last = t
    line = f"step -{k}: span {(t1 - t0) / 1e6:7.3f} ms, {len(seg)} kernels, GPU busy {busy / (t1 - t0) * 100:5.1f} %, {len(gaps)} idle gaps > 5 us (total {sum(g for g, _ in gaps) / 1e3:.1f} us)"
    print(line)
    if k == 1:
        for g, at in sorted(gaps, reverse=True)[:8]:
            before = max((r for r in seg if r[2] <= at + 1), key=lambda r: r[2])
            after = min((r for r in seg if r[1] >= at + g - 1), key=lambda r: r[1])
            print(f"      {g / 1e3:7.1f} us idle   {short(before[0])} -> {short(after[0])}")
        by = {}
        for n, s, e, st, q in seg:
            d = by.setdefault(st, [s, e, 0, 0, n])
            d[0] = min(d[0], s)
            if e >= d[1]:
                d[1], d[4] = e, n
            d[2] += e - s; d[3] += 1
        print(f"  per {'stream' if scol else 'trace (no stream column)'}:")
        for st, (s, e, run, n, lastn) in sorted(by.items(), key=lambda kv: -kv[1][2]):
            print(f"    stream {st}: {n:4d} kernels, kernel time {run / 1e6:7.3f} ms, first start +{(s - t0) / 1e6:6.3f} ms, last end {(e - t1) / 1e6:+7.3f} ms vs step end   (last: {short(lastn)})")
        # graph replays carry no stream id: classify by kernel name (weight-gradient family = the side stream, decode = the
        # post_forward stream, everything else = the launch stream) and ask which class ends the step, and how long each runs alone
        def cls(n):
            if any(k in n for k in ("wgrad", "colsum", "dwdeconv_bwd_weight", "zero_kernel", "sum_slabs")):
                return "side (weight gradients)"
            if any(k in n for k in ("topk", "ctdet_stage2", "pose_assemble")):
                return "decode (post_forward)"
            if "adam" in n:
                return "optimizer"
            return "launch stream"
        per = {}
        for n, s, e, st, q in seg:
            d = per.setdefault(cls(n), [s, e, 0, 0])
            d[0] = min(d[0], s); d[1] = max(d[1], e); d[2] += e - s; d[3] += 1
        print("  by kernel class (names; a replayed graph has no stream ids):")
        for k, (s, e, run, n) in sorted(per.items(), key=lambda kv: kv[1][1]):
            print(f"    {k:26s} {n:4d} kernels, kernel time {run / 1e6:7.3f} ms, first start +{(s - t0) / 1e6:7.3f} ms, last end +{(e - t0) / 1e6:7.3f} ms ({(e - t1) / 1e6:+.3f} vs step end)")
        # hardware queues: which kernel classes share a queue (kernels of one queue run strictly one after the other)
        perq = {}
        for n, s, e, st, q in seg:
            d = perq.setdefault(q, {})
            c_ = d.setdefault(cls(n), [0, 0, s, e])
            c_[0] += 1; c_[1] += e - s; c_[2] = min(c_[2], s); c_[3] = max(c_[3], e)
        print("  hardware queues (kernel class: launches, kernel time, first start .. last end):")
        for q, d in sorted(perq.items()):
            for k, (n, run, s, e) in sorted(d.items(), key=lambda kv: -kv[1][1]):
                print(f"    queue {q}: {k:26s} {n:4d} kernels {run / 1e6:7.3f} ms   +{(s - t0) / 1e6:7.3f} .. +{(e - t0) / 1e6:7.3f} ms")
        # how long does each class run with nothing of the OTHER main class in flight?
        marks = sorted([(s, 1, cls(n)) for n, s, e, st, q in seg] + [(e, -1, cls(n)) for n, s, e, st, q in seg])
        depth, last, alone = {}, t0, {}
        for t, d, k in marks:
            live = tuple(sorted(c_ for c_, v in depth.items() if v > 0))
            alone[live] = alone.get(live, 0) + (t - last)
            depth[k] = depth.get(k, 0) + d
            last = t
        print("  time by the set of classes in flight:")
        for live, t in sorted(alone.items(), key=lambda kv: -kv[1])[:8]:
            print(f"    {t / 1e6:7.3f} ms  {' + '.join(live) if live else '(idle)'}")
        import os
        dump = os.environ.get("REPLAY_DUMP")
        if dump:
            with open(dump, "w") as f:
                for n, s, e, st, q in seg:
                    f.write(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f}  q{q}  {short(n)}\n")
        foreign = {}
        for n, s, e, st, q in seg:
            if "at::native" in n or "rocclr" in n or n.startswith("void at::"):
                d = foreign.setdefault(short(n)[:110], [0, 0])
                d[0] += 1; d[1] += e - s
        print("  kernels that are not the library's in this step:" + (" none" if not foreign else ""))
        for n, (cnt, t) in sorted(foreign.items(), key=lambda kv: -kv[1][0]):
            print(f"    {cnt:4d} launches {t / 1e3:8.1f} us  {n}")
