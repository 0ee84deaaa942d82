#!/bin/bash
# SQ counters of the fused cross-attention kernel (and the three kernels of the composite it replaces) on tools/bench_xattn.py
set -e
export TMPDIR=/tmp
out=$1; mkdir -p "$out"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-include-regex "xattn_fwd|attn_fwd|gemm_nt_p8|ln_fwd16" -d "$out/sq" -o p -f csv -- python3 tools/bench_xattn.py > "$out/sq.log" 2>&1 || tail -3 "$out/sq.log"
python3 - "$out" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        acc[(k[:60], int(r.get("Grid_Size", 0) or 0))][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("rocprofv3 --pmc SQ_* over tools/bench_xattn.py (fused cross-attention block vs the composite's kernels); per-launch means")
print(f"{'kernel':60s} {'grid':>8s} {'n':>4s} {'MFMA busy/(4 x CU busy)':>20s} {'WAIT_ANY / WAVE':>16s} {'WAIT_INST / WAVE':>17s} {'ACTIVE / WAVE':>14s}")
for key in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[key].items()}
    if not c.get("SQ_BUSY_CU_CYCLES") or not c.get("SQ_WAVE_CYCLES"):
        continue
    print(f"{key[0]:60s} {key[1]:8d} {len(acc[key]['SQ_WAVE_CYCLES']):4d} {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * c['SQ_BUSY_CU_CYCLES']):20.3f} "
          f"{c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']:16.3f} {c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:17.3f} {c.get('SQ_ACTIVE_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:14.3f}")
PY
