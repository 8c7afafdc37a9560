"""Summarise a rocprofv3 rocpd sqlite (kernel-trace) into a per-kernel table: calls, total, average, share."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(cur.execute(f"select {name_col}, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'share':>7s}")
for n, c, t, mn, mx in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    short = re.sub(r"\(.*", "", n)[:90]
    print(f"{short:90s} {c:7d} {t/1e6:10.3f} {t/c/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*t/tot:6.2f}%")
print(f"TOTAL kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} launches")
