#!/bin/bash
# PV encoder backward's weight gradients on its own stream (first n layers): A/B on one box, and the phase timeline
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r20; mkdir -p $O
for i in 1 2; do for v in 3 0 6 2 4; do
  SPMM_PV_WGRAD_INLINE=$v timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-kernel-timing > $O/bench_v${v}_$i.json 2>> $O/bench.err
done; done
python - <<'PY'
import json
for v in (0, 2, 3, 4, 6):
    for i in (1, 2):
        d=json.loads([l for l in open(f"gpurun_out/r20/bench_v{v}_{i}.json") if l.startswith("{")][-1])
        print("inline layers", v, d["value"], d["ms_per_step"], d["step_ms"]["median"])
PY
python tools/phase_times.py 12 2>&1 | grep -v amdgpu.ids | tee $O/phase_times.txt
