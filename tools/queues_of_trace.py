"""Which hardware queue / stream every kernel of a rocprofv3 --kernel-trace run (rocpd database) ran on: per (queue, stream) the launch count and
the three most frequent kernel names.   python tools/queues_of_trace.py DB"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tables if "kernel_dispatch" in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
print("table", kd, "columns", cols)
qcol = [c for c in cols if "queue" in c][0]
scol = [c for c in cols if "stream" in c]
scol = scol[0] if scol else qcol
ksym = [t for t in tables if "kernel_symbol" in t]
name_expr, join = "d.kernel_id", ""
if ksym:
    kc = [r[1] for r in cur.execute(f"pragma table_info({ksym[0]})")]
    namec = [c for c in kc if "name" in c][0]
    idc = "id" if "id" in kc else kc[0]
    name_expr, join = f"k.{namec}", f" join {ksym[0]} k on k.{idc} = d.kernel_id"
rows = cur.execute(f"select d.{qcol}, d.{scol}, {name_expr}, count(*) from {kd} d{join} group by 1, 2, 3").fetchall()
agg = {}
for q, s, n, c in rows:
    agg.setdefault((q, s), []).append((c, re.sub(r"\(anonymous namespace\)::", "", str(n))[:60]))
for (q, s), v in sorted(agg.items(), key=lambda kv: -sum(c for c, _ in kv[1])):
    v.sort(reverse=True)
    print(f"queue {q} stream {s}: {sum(c for c, _ in v)} launches; top: " + " | ".join(f"{n} x{c}" for c, n in v[:3]))
