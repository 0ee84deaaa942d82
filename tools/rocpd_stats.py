"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel table (calls, total, avg, %)."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
print(f"total kernel time {tot/1e6:.3f} ms over {nsteps:g} steps -> {tot/1e6/nsteps:.3f} ms/step")
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'%':>6s}")
for n, c, t, a, mn, mx in rows[:40]:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    print(f"{n[:90]:90s} {c:7d} {t/1e6:10.3f} {a/1e3:9.2f} {mn/1e3:8.2f} {mx/1e3:9.2f} {100*t/tot:6.2f}")
