"""Idle time of the GPU inside the steady-state steps of a rocprofv3 --kernel-trace run (rocpd database): the union of all kernel intervals
over every queue, the gaps in it, and what ran right before / after the largest gaps.
    python tools/idle_gaps.py DB [skip_fraction=0.5] [min_gap_us=15]
    python tools/idle_gaps.py DB 0.5 8 3      # ... and the kernels around the three largest gaps
The window is whole steps between two optimiser launches of the middle of the run (pass a trace of a plain
`bench.py --steps N --no-kernel-timing --no-other-configs` run).  CAVEAT (round 4): under rocprofv3 the host enqueues slower (the step takes
61.4 ms instead of 57.5-59) and the ~1.5 ms per step found idle -- 0.8-1.0 ms between the last weight-gradient reduction and the optimiser,
start-of-step concatenations -- is about that difference: without the tracer the host runs a step ahead (the captured-graph step, which has
no host side at all, is no faster than the eager one on the same layout)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = [c for c in cols if "queue" in c]
rows = cur.execute(f"select start, end, {name_col}" + (f", {qcol[0]}" if qcol else ", 0") + " from kernels order by start").fetchall()
# steady-state window: from the end of one optimiser launch to the end of a later one (whole steps)
ad = [r[1] for r in rows if "adamw_kernel" in str(r[2])]
if len(ad) >= 6:
    a, b = len(ad) // 3, len(ad) - 2                       # skip the first third (warm-up, schedule check) and the last step
    lo, hi, nsteps = ad[a], ad[b], b - a
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    print(f"window: optimiser launch {a} -> {b} of {len(ad)} = {nsteps} steps, {(hi - lo) / 1e6 / nsteps:.2f} ms per step")
else:
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t0 + (t1 - t0) * skip
    rows = [r for r in rows if r[0] >= lo]
span = (max(r[1] for r in rows) - rows[0][0]) / 1e3
busy_sum = sum(r[1] - r[0] for r in rows) / 1e3
gaps, cur_end, cur_name = [], rows[0][1], rows[0][2]
for s, e, n, q in rows[1:]:
    if s > cur_end:
        gaps.append(((s - cur_end) / 1e3, cur_name, n))
    if e > cur_end:
        cur_end, cur_name = e, n
idle = sum(g for g, _, _ in gaps)
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", str(n))[:48]
print(f"analysed span {span / 1e3:.1f} ms, {len(rows)} kernels; sum of kernel durations {busy_sum / 1e3:.1f} ms ({busy_sum / span:.2f} x the span: overlap); "
      f"no kernel on any queue for {idle / 1e3:.2f} ms = {100 * idle / span:.1f} % of the span")
big = [g for g in gaps if g[0] >= min_gap]
print(f"gaps >= {min_gap:g} us: {len(big)}, together {sum(g for g, _, _ in big) / 1e3:.2f} ms; gaps < {min_gap:g} us: {len(gaps) - len(big)}, together "
      f"{(idle - sum(g for g, _, _ in big)) / 1e3:.2f} ms")
agg = {}
for g, a, b in big:
    k = (short(a), short(b))
    agg.setdefault(k, [0, 0.0])
    agg[k][0] += 1; agg[k][1] += g
print("largest gap classes (kernel that ended last before the gap -> kernel that started after it):")
for (a, b), (c, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {tot / 1e3:7.2f} ms in {c:4d} gaps   {a}  ->  {b}")
# context of the largest gaps: the kernels that ended / started around them (all queues)
ctx = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if ctx:
    ends = sorted(rows, key=lambda r: r[1])
    cur_end, out = rows[0][1], []
    for i, (s, e, n, q) in enumerate(rows[1:], 1):
        if s > cur_end and (s - cur_end) / 1e3 >= 200.0:
            out.append((cur_end, s))
        cur_end = max(cur_end, e)
    for a, b in out[:ctx]:
        print(f"--- gap of {(b - a) / 1e3:.0f} us")
        before = [r for r in ends if r[1] <= a][-6:]
        after = [r for r in rows if r[0] >= b][:8]
        for s, e, n, q in before:
            print(f"   ended {(e - a) / 1e3:9.1f} us  (ran {(e - s) / 1e3:7.1f} us, queue {q})  {short(n)}")
        for s, e, n, q in after:
            print(f"   began {(s - b) / 1e3:9.1f} us  (ran {(e - s) / 1e3:7.1f} us, queue {q})  {short(n)}")
