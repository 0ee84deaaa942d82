#!/bin/bash
# Upper bound of the fp8 tier (EXPERIMENTS.md 3.5): BASELINE configs[4]'s per-GPU shape (B = 512, Lt = 256) with every eligible NT GEMM
# (forward and data gradient; plain / +R / GELU + derivative epilogues) on the E4M3 kernel, operands = the bytes of the bf16 tensors
# reinterpreted (no quantisation pass: "free" quantisation; results are garbage, only the clock is read), next to the bf16 step.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/fp8_bound; mkdir -p $O
for tag in bf16 fp8all; do
  env SPMM_FP8_TIMING_EXPERIMENT=$([ $tag = fp8all ] && echo 1 || echo 0) timeout 900 python bench.py --batch 512 --seq-len 256 --steps 5 --warmup 4 --no-cpu-baseline --no-other-configs $([ $tag = fp8all ] && echo --no-kernel-timing) > $O/$tag.json 2> $O/$tag.err
done
python - <<'PY'
import json
r={}
for t in ("bf16","fp8all"):
    d=json.loads([l for l in open(f"gpurun_out/fp8_bound/{t}.json") if l.startswith("{")][-1]); r[t]=d
    ro=d.get("roofline") or {}
    print(t, "ms/step", d["ms_per_step"], "median", d["step_ms"]["median"], "| NT GEMM launches of the bf16 step:", ro.get("gemm_ms_per_step"), "ms at", ro.get("achieved"), "TF/s", "| clock", (d.get("power") or {}).get("clock_mhz"), "MHz", (d.get("power") or {}).get("power_w"), "W")
print("upper bound of the tier: %.3fx" % (r["bf16"]["ms_per_step"]/r["fp8all"]["ms_per_step"]))
PY
