#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r9; mkdir -p $O
for M in 6912 12300 13824 16384 24600 28700; do echo "== M=$M"; python tools/gemm_small_m.py $M; done > $O/small_m.txt 2>&1
cat $O/small_m.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > $O/t1.txt 2>&1; tail -3 $O/t1.txt
timeout 1800 python -m pytest tests/test_step_gpu.py -x -q -k "golden or gradients_match_oracle or trainer or checkpoint or hipgraph or packed_text or full_depth" > $O/t2.txt 2>&1; tail -5 $O/t2.txt
for i in 1 2; do timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-kernel-timing > $O/bench$i.json 2>> $O/bench.err; done
python - <<'PY'
import json
for f in ("bench1","bench2"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r9/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["step_ms"]["median"], d.get("power",{}).get("clock_mhz"), d["losses"])
    except Exception as e: print(f, "ERR", e)
PY
