"""Idle gaps of the GPU inside one training step, from a rocprofv3 rocpd database (kernel trace of bench.py): the union of all
kernels' [start, end) intervals, the largest holes in it and the kernels on either side.   python3 tools/trace_gaps.py DB [step]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end, stream_id from kernels order by start").fetchall()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(ad) - 2
seg = rows[ad[k] + 1: ad[k + 1] + 1]
t0, t1 = seg[0][1], max(r[2] for r in seg)
gaps, cur_e, last = [], seg[0][2], seg[0]
for r in seg[1:]:
    if r[1] > cur_e:
        gaps.append((r[1] - cur_e, cur_e - t0, last[0], r[0]))
    if r[2] > cur_e:
        cur_e, last = r[2], r
busy = (t1 - t0) - sum(g[0] for g in gaps)
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)[:48]
print(f"step {k}: wall {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {sum(g[0] for g in gaps) / 1e6:.2f} ms in {len(gaps)} gaps; kernels {len(seg)}")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"  {g[0] / 1e3:8.1f} us at {g[1] / 1e6:7.2f} ms   after {short(g[2]):48s} before {short(g[3])}")
