"""Summarise a rocprofv3 --pmc pass (rocpd database): average counter value per (kernel, grid) for the selected kernels."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else "gemm")
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
assert 'counters_collection' in tabs, tabs
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
namec = 'kernel_name' if 'kernel_name' in cols else [c for c in cols if 'kernel' in c and 'name' in c][0]
gridc = 'grid_size' if 'grid_size' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
sel = f"{namec}, {gridc}" if gridc else f"{namec}, 0"
rows = cur.execute(f"select {sel}, counter_name, avg(value), count(*) from counters_collection group by {sel}, counter_name").fetchall()
print(f"{'kernel':60s} {'grid':>10s} {'counter':>14s} {'avg':>14s} {'n':>4s}")
for r in rows:
    if pat.search(r[0]):
        print(f"{re.sub(r'[(]anonymous namespace[)]::', '', r[0])[:60]:60s} {r[1]:>10} {r[2]:>14s} {r[3]:14.6g} {r[4]:4d}")

if len(sys.argv) > 3:      # per-dispatch listing in dispatch order
    idc = 'dispatch_id' if 'dispatch_id' in cols else cols[0]
    for r in cur.execute(f"select {idc}, {sel}, counter_name, value from counters_collection order by {idc}"):
        if pat.search(r[1]): print(r[0], r[1][:40], r[2], r[3], r[4])
