"""rocpd kernel-trace database -> which hardware queue every stream's kernels ran on, plus the per-kernel table.
   python tools/trace_queues.py <p_results.db> <steps>"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("columns:", cols)
qcol = next((c for c in cols if c == "queue_id"), None) or next((c for c in cols if "queue" in c), None)
scol = next((c for c in cols if c == "stream_id"), None) or next((c for c in cols if "stream" in c), None)
ncol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
if qcol and scol:
    print(f"\n(queue, stream) -> launches, total ms, three most frequent kernels   [{qcol}, {scol}]")
    for q, s, n, t in cur.execute(f"select {qcol}, {scol}, count(*), sum(end-start) from kernels group by {qcol}, {scol} order by 3 desc"):
        top = cur.execute(f"select {ncol}, count(*) from kernels where {qcol}=? and {scol}=? group by {ncol} order by 2 desc limit 3", (q, s)).fetchall()
        print(f"  queue {q} stream {s}: {n:6d} launches {t / 1e6:9.2f} ms  " + "; ".join(f"{re.sub(r'.*::', '', a)[:40]} x{b}" for a, b in top))
rows = cur.execute(f"select {ncol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {ncol} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"\ntotal kernel time {tot / 1e6:.3f} ms over {nsteps:g} steps -> {tot / 1e6 / nsteps:.3f} ms/step")
for n, c, t, a, mn, mx in rows[:24]:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    print(f"{n[:80]:80s} {c:7d} {t / 1e6:10.3f} ms {a / 1e3:9.2f} us avg {mn / 1e3:8.2f} min {mx / 1e3:9.2f} max")
