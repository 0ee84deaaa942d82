"""Timeline around the n-th launch of a kernel in the last full step of a rocprofv3 rocpd database: every kernel that intersects
[start - before, end + after] with its stream, start offset and duration.   python3 tools/trace_window.py DB NAME_REGEX [n] [before_us] [after_us]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end, stream_id from kernels order by start").fetchall()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
seg = rows[ad[-2] + 1: ad[-1] + 1]
pat = re.compile(sys.argv[2])
hits = [r for r in seg if pat.search(r[0])]
n = int(sys.argv[3]) if len(sys.argv) > 3 else len(hits) // 2
before, after = (float(sys.argv[4]) if len(sys.argv) > 4 else 300.0) * 1e3, (float(sys.argv[5]) if len(sys.argv) > 5 else 500.0) * 1e3
h = hits[n]
lo, hi = h[1] - before, h[2] + after
short = lambda s: re.sub(r"\(anonymous namespace\)::|void |at::native::|_ZN12_GLOBAL__N_1\d+", "", s)[:44]
print(f"{len(hits)} launches match; launch {n}: {short(h[0])} at {(h[1] - seg[0][1]) / 1e6:.2f} ms of the step, {(h[2] - h[1]) / 1e3:.1f} us, stream {h[3]}")
for r in seg:
    if r[2] > lo and r[1] < hi:
        print(f"  {'*' if r is h else ' '} stream {r[3]:3d}  +{(r[1] - h[1]) / 1e3:9.1f} us  {(r[2] - r[1]) / 1e3:8.1f} us  {short(r[0])}")
