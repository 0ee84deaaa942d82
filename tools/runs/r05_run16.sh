#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r16; mkdir -p $O
timeout 600 python tools/stress_decode_attn.py 8 2>&1 | grep -v amdgpu.ids | tee $O/stress.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "decode_attention" > $O/t1.txt 2>&1; tail -5 $O/t1.txt
timeout 1800 python -m pytest tests/ -q -m gpu -k "decode or beam or pv2smiles or smiles2pv" > $O/t2.txt 2>&1; tail -8 $O/t2.txt
for i in 1 2; do
timeout 900 python bench.py --decode > $O/bench_decode$i.json 2> $O/bench_decode.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r16/bench_decode$i.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d.get("ms_per_position"), d.get("position_breakdown_ms"), d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
PY
done
