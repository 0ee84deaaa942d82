"""Per-kernel table (calls, total, avg, %) from a rocprofv3 --kernel-trace csv:  python3 tools/trace_stats.py DIR NSTEPS"""
import csv, glob, collections, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[n]; a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print(f"total kernel time {tot / 1e3:.3f} ms over {nsteps:g} steps -> {tot / 1e3 / nsteps:.3f} ms/step")
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'%':>6s}")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{n[:90]:90s} {a[0]:7d} {a[1] / 1e3:10.3f} {a[1] / a[0]:9.2f} {a[2]:8.2f} {a[3]:9.2f} {100 * a[1] / tot:6.2f}")
