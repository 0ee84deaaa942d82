import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = 'counters_collection' if 'counters_collection' in tabs else None
print([t for t in tabs if 'pmc' in t.lower() or 'counter' in t.lower()][:10])
if view:
    cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
    print(cols)
    namec = 'kernel_name' if 'kernel_name' in cols else [c for c in cols if 'kernel' in c and 'name' in c][0]
    rows = cur.execute(f"select {namec}, counter_name, avg(value), count(*) from {view} group by {namec}, counter_name").fetchall()
    for r in rows:
        if 'gemm' in r[0]: print(re.sub(r"\(anonymous namespace\)::", "", r[0])[:40], r[1], f"{r[2]:.4g}", r[3])
