"""Summarise tools/pmc_step.sh: profiles/<prefix>_pmc_nt_gemm.{json,txt} (HBM-side traffic per NT GEMM launch, corrected as
MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts 16-B/lane streaming reads at half) and
<prefix>_pmc_sq.txt (MFMA-busy and wait fractions per kernel).   python3 tools/pmc_step_summary.py OUTDIR profiles/r02"""
import csv, glob, json, os, re, sys
from collections import defaultdict

out, prefix = sys.argv[1], sys.argv[2]


def load(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            acc[(k, int(r.get("Grid_Size", 0) or 0))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = load("fetch"), load("write")
lines = ["rocprofv3 PMC passes over one bench.py step pair (B=128, Lt=128, full depth, single-stream schedule), NT GEMM kernels; KiB per launch, raw counters",
         f"{'kernel':52s} {'grid':>8s} {'n':>4s} {'FETCH_SIZE':>12s} {'WRITE_SIZE':>12s}"]
tf = tw = n = 0
for key in sorted(fetch):
    fv = fetch[key].get("FETCH_SIZE", [])
    wv = write.get(key, {}).get("WRITE_SIZE", [])
    if not fv or not wv:
        continue
    lines.append(f"{key[0][:52]:52s} {key[1]:8d} {len(fv):4d} {sum(fv) / len(fv):12.1f} {sum(wv) / len(wv):12.1f}")
    tf += sum(fv); tw += sum(wv); n += len(fv)
if n:
    js = {"workload": {"batch": 128, "seq_len": 128, "layers": 12, "queue": 36864},
          "kernel": "gemm_nt_* (all NT GEMM launches of the step)",
          "command": "tools/pmc_step.sh: rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-include-regex gemm_nt -- python3 bench.py --steps 1 --warmup 1 "
                     "--no-cpu-baseline --no-kernel-timing (one pass per counter, SPMM_STREAMS=1)",
          "launches_counted": n, "fetch_bytes_per_launch": 2 * tf * 1024 / n, "write_bytes_per_launch": tw * 1024 / n,
          "traffic_bytes_per_launch": (2 * tf + tw) * 1024 / n,
          "corrections": "counter unit KiB; FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B/lane streaming reads, MI355X_MICROARCH.md HBM "
                         "section); WRITE_SIZE as is; Infinity-Cache hits are counted, so this is L2<->fabric traffic, an upper bound on HBM bytes"}
    json.dump(js, open(prefix + "_pmc_nt_gemm.json", "w"), indent=1)
    open(prefix + "_pmc_nt_gemm.txt", "w").write("\n".join(lines) + "\n")
    print(json.dumps(js, indent=1))

sq = load("sq")
rows = ["SQ counters per kernel (average over the launches of one bench.py step pair; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles,",
        "SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CU_CYCLES cycles; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES))",
        f"{'kernel':52s} {'grid':>8s} {'n':>4s} {'mfma_busy':>10s} {'wait_any':>9s} {'wait_inst':>10s} {'active':>8s} {'gui_active':>12s}"]
agg = defaultdict(lambda: defaultdict(float))
for key in sorted(sq):
    c = {k: sum(v) / len(v) for k, v in sq[key].items()}
    if "SQ_BUSY_CU_CYCLES" not in c or c["SQ_BUSY_CU_CYCLES"] == 0:
        continue
    nl = len(next(iter(sq[key].values())))
    wc = c.get("SQ_WAVE_CYCLES", 1) or 1
    rows.append(f"{key[0][:52]:52s} {key[1]:8d} {nl:4d} {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * c['SQ_BUSY_CU_CYCLES']):10.3f} "
                f"{c.get('SQ_WAIT_ANY', 0) / wc:9.3f} {c.get('SQ_WAIT_INST_ANY', 0) / wc:10.3f} {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:8.3f} {c.get('GRBM_GUI_ACTIVE', 0):12.0f}")
    fam = key[0].split("<")[0].replace("void ", "")
    for k, v in sq[key].items():
        agg[fam][k] += sum(v)
rows.append("")
rows.append("per kernel family, summed over all launches: MFMA busy / (4 x CU busy cycles)")
for fam, c in sorted(agg.items()):
    if c.get("SQ_BUSY_CU_CYCLES"):
        rows.append(f"  {fam:40s} {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * c['SQ_BUSY_CU_CYCLES']):.3f}")
open(prefix + "_pmc_sq.txt", "w").write("\n".join(rows) + "\n")
print("\n".join(rows[-12:]))
