"""Text timeline of ONE steady-state step of a rocprofv3 --kernel-trace run (rocpd database): per hardware queue, the busy fraction of every
1-ms bin (0-9, '.' = idle), the queue's busy time, and the kernels that run in the step's last milliseconds.
    python tools/step_timeline.py DB [bin_ms=1.0]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
binms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = [c for c in cols if "queue" in c][0]
rows = cur.execute(f"select start, end, {name_col}, {qcol} from kernels order by start").fetchall()
ad = [r[1] for r in rows if "adamw_kernel" in str(r[2])]
a = len(ad) * 2 // 3
lo, hi = ad[a], ad[a + 1]
rows = [r for r in rows if r[1] > lo and r[0] < hi]
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", str(n))[:44]
nb = int((hi - lo) / 1e6 / binms) + 1
print(f"step between optimiser launches {a} and {a + 1}: {(hi - lo) / 1e6:.2f} ms, {len(rows)} kernels, bins of {binms:g} ms")
qs = sorted({r[3] for r in rows})
for q in qs:
    busy = [0.0] * nb
    tot = 0.0
    for s, e, n, qq in rows:
        if qq != q:
            continue
        s, e = max(s, lo), min(e, hi)
        tot += (e - s) / 1e6
        b0, b1 = int((s - lo) / 1e6 / binms), int((e - lo) / 1e6 / binms)
        for b in range(b0, min(b1, nb - 1) + 1):
            bs, be = lo + b * binms * 1e6, lo + (b + 1) * binms * 1e6
            busy[b] += max(0.0, min(e, be) - max(s, bs)) / 1e6 / binms
    line = "".join("." if x < 0.05 else str(min(9, int(x * 10))) for x in busy)
    print(f"queue {q:>3}: busy {tot:6.2f} ms  {line}")
tail_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 1.5
head_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
print(f"last {tail_ms:g} ms of the step (times relative to the end of this step's AdamW kernel):")
for s, e, n, q in rows:
    if e > hi - tail_ms * 1e6:
        print(f"   queue {q:>3}  {(s - hi) / 1e3:9.1f} .. {(e - hi) / 1e3:8.1f} us   {short(n)}")
if head_ms > 0:
    print(f"first {head_ms:g} ms of the step (times relative to the end of the previous step's AdamW kernel):")
    for s, e, n, q in rows:
        if s < lo + head_ms * 1e6:
            print(f"   queue {q:>3}  {(s - lo) / 1e3:9.1f} .. {(e - lo) / 1e3:8.1f} us   {short(n)}")
