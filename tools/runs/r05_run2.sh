#!/bin/bash
# round 5, GPU run 2: plan kernels + CLS-only top layer -- new kernel tests, step parity tests, bench A/B
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "pack_plan or fusion_plan or row_helpers or position0" > $O/test_new_kernels.txt 2>&1
tail -5 $O/test_new_kernels.txt
timeout 2400 python -m pytest tests/test_step_gpu.py -x -q > $O/test_step.txt 2>&1
tail -15 $O/test_step.txt
timeout 600 python bench.py --no-other-configs --no-cpu-baseline > $O/bench_cls.json 2> $O/bench_cls.err
SPMM_CLS_ONLY_TOP=0 timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-kernel-timing > $O/bench_full.json 2> $O/bench_full.err
tail -c 400 $O/bench_cls.err; python - <<'PY'
import json
for f in ("bench_cls","bench_full"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r2/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["step_ms"], d.get("power",{}).get("clock_mhz"), d["losses"])
    except Exception as e: print(f, "ERR", e)
PY
