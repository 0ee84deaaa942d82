#!/bin/bash
# round 6, run 7: off-path maintenance deferred behind the unimodal chains (SPMM_OFF_PATH_DEFER=1, new default) vs issued at once (=0)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06_7; mkdir -p $O
timeout 900 python -m pytest -x -q tests/test_step_gpu.py -k "graph or checkpoint or trace or token_count or full_depth_training" 2>&1 | tail -3
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs"
for i in 1 2 3 4; do for v in 0 1; do
  SPMM_OFF_PATH_DEFER=$v timeout 600 $B > $O/d_${v}_$i.json 2> $O/d_${v}_$i.err
  python - $O/d_${v}_$i.json $v <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print("defer", sys.argv[2], d["ms_per_step"], d["step_ms"]["median"])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done; done
PHASES_BOUNDARY=1 timeout 600 python tools/phase_times.py 14 2>&1 | tail -24 | head -12
