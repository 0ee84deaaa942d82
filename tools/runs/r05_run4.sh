#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "fusion_plan or transpose_cast_gather_acc" > $O/t1.txt 2>&1; tail -3 $O/t1.txt
timeout 1200 python -m pytest tests/test_step_gpu.py -x -q -k "published_size or packed_text or full_depth or benchmark_shape_forward" > $O/t2.txt 2>&1; tail -8 $O/t2.txt
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-kernel-timing > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4/bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["step_ms"], d.get("power",{}).get("clock_mhz"))
PY
