"""Average rocprofv3 --pmc counters (csv output) per kernel name over the dispatches of a run directory tree."""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    for c in sorted(m):
        print(f"   {c:34s} {m[c]:16.1f}  (n={len(cs[c])})")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CU_CYCLES" in m:
        # MFMA_BUSY counts cycles per SIMD summed over the chip (4 SIMDs per CU); BUSY_CU_CYCLES counts busy cycles per CU
        print(f"   -> MFMA busy / (4 x CU busy cycles) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_BUSY_CU_CYCLES']):.3f}")
    if "SQ_WAVE_CYCLES" in m:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"):
            if c in m:
                print(f"   -> {c} / SQ_WAVE_CYCLES = {m[c] / m['SQ_WAVE_CYCLES']:.3f}")
