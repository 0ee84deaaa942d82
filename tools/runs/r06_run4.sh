#!/bin/bash
# round 6, run 4: the fused cross-attention kernel on the no-grad passes only (momentum fusion pass) vs off, inside the step; new kernel test
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06_4; mkdir -p $O
timeout 600 python -m pytest -x -q tests/test_kernels_gpu.py -k "few_device_side_rows or device_side_row_counts" 2>&1 | tail -3
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs"
for i in 1 2 3; do for v in off nograd all; do
  SPMM_FUSED_XATTN=$v timeout 600 $B > $O/x_${v}_$i.json 2> $O/x_${v}_$i.err
  python - $O/x_${v}_$i.json $v <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[2], d["ms_per_step"], d["step_ms"]["median"], d["losses"])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done; done
