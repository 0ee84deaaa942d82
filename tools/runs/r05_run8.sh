#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r8; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q > $O/t1.txt 2>&1; tail -4 $O/t1.txt
timeout 3000 python -m pytest tests/test_step_gpu.py -x -q > $O/t2.txt 2>&1; tail -8 $O/t2.txt
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-kernel-timing > $O/bench_dyn.json 2> $O/bench.err
SPMM_CLS_ONLY_TOP=0 timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-kernel-timing > $O/bench_full.json 2>> $O/bench.err
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-kernel-timing > $O/bench_dyn2.json 2>> $O/bench.err
tail -3 $O/bench.err
python - <<'PY'
import json
for f in ("bench_dyn","bench_full","bench_dyn2"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r8/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["step_ms"]["median"], d.get("power",{}).get("clock_mhz"), d["losses"], d["executed_step_tflop"])
    except Exception as e: print(f, "ERR", e)
PY
