"""Totals of one rocprofv3 --pmc pass (rocpd database) over the kernels matching a LIKE pattern:
prints launches, sum of the counter (raw unit) and the per-(kernel, grid) averages."""
import sqlite3, sys, re, json
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
counter, like = sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "%gemm_nt%")
n, s = cur.execute("select count(*), sum(value) from counters_collection where kernel_name like ? and counter_name=?", (like, counter)).fetchone()
rows = cur.execute("select kernel_name, grid_size, avg(value), count(*) from counters_collection where kernel_name like ? and counter_name=? "
                   "group by kernel_name, grid_size", (like, counter)).fetchall()
print(json.dumps({"counter": counter, "launches": n, "sum": s}))
for k, g, a, m in rows:
    print(f"{re.sub(r'[(]anonymous namespace[)]::', '', k)[:44]:44s} {g:9d} {m:4d} {a:14.1f}")
