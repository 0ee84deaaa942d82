"""Kernels of a step that cannot fill the chip: fewer workgroups than 2 x 256 CUs and more than 10 us per launch.
   python3 tools/trace_small_grids.py DB"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, (grid_x*grid_y*grid_z)/(workgroup_x*workgroup_y*workgroup_z) as wgs, count(*), avg(end-start), sum(end-start) "
                   "from kernels group by name, wgs having wgs < 512 and avg(end-start) > 10000 order by 5 desc").fetchall()
print(f"{'kernel':70s} {'wgs':>6s} {'calls':>6s} {'avg_us':>8s} {'total_ms':>9s}")
for n, w, c, a, t in rows[:40]:
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)
    print(f"{n[:70]:70s} {w:6d} {c:6d} {a / 1e3:8.1f} {t / 1e6:9.3f}")
