#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "fusion_plan or transpose_cast_gather_acc or dropout_and_residual or gemm_tile_kernels" > $O/t1.txt 2>&1; tail -3 $O/t1.txt
timeout 2400 python -m pytest tests/test_step_gpu.py -x -q > $O/t2.txt 2>&1; tail -8 $O/t2.txt
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-kernel-timing > $O/bench_fused.json 2> $O/bench.err
SPMM_FUSE_DROP_RES=0 timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-kernel-timing > $O/bench_twolaunch.json 2>> $O/bench.err
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-kernel-timing > $O/bench_fused2.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ("bench_fused","bench_twolaunch","bench_fused2"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r5/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["step_ms"]["median"], d.get("power",{}).get("clock_mhz"), d["losses"])
    except Exception as e: print(f, "ERR", e)
PY
