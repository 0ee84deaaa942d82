#!/bin/bash
# momentum text encoder's last layer on position-0 rows: plan test, step parity, A/B of the bench
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r10; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "plan" > $O/t1.txt 2>&1; tail -3 $O/t1.txt
timeout 2400 python -m pytest tests/test_step_gpu.py -x -q -k "not timing and not decode" > $O/t2.txt 2>&1; tail -5 $O/t2.txt
for i in 1 2; do
  timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-kernel-timing > $O/bench$i.json 2>> $O/bench.err
done
python - <<'PY'
import json
for f in ("bench1","bench2"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r10/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["step_ms"]["median"], d.get("power",{}).get("clock_mhz"), d["losses"])
    except Exception as e: print(f, "ERR", e)
PY
python tools/bench_decode_attn.py > $O/decode_attn.txt 2>&1; tail -30 $O/decode_attn.txt
