#!/bin/bash
# the driver's round-end checks: the whole GPU suite, smoke, the default bench line
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/suite; mkdir -p $O
timeout 5400 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -8 $O/gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/suite/bench_default.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["step_ms"], d["roofline"]["frac"], d["roofline"].get("frac_at_clock"), d["roofline"].get("traffic"))
print({k:(v.get("value"),v.get("ms_per_step"),v.get("ms_per_position"),v.get("error")) for k,v in d["other_configs"].items()})
print(d["cpu_baseline"])
PY
