#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r18; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "beam_step or decode_attention" > $O/t1.txt 2>&1; tail -5 $O/t1.txt
timeout 1800 python -m pytest tests/ -q -m gpu -k "decode or beam or pv2smiles or smiles2pv" > $O/t2.txt 2>&1; tail -8 $O/t2.txt
for b in 0.5 0; do
timeout 900 python bench.py --decode --no-cpu-baseline --sep-bias $b > $O/bench_decode_b$b.json 2> $O/bench_decode.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r18/bench_decode_b$b.json") if l.startswith("{")][-1])
print("sep bias $b:", d["value"], d["ms_per_step"], d.get("ms_per_position"), d.get("finished_hypotheses"), d.get("last_chunk"), d["roofline"]["avg_launch_us"])
PY
done
