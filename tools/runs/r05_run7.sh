#!/bin/bash
# knob sweep on the round-5 step (CLS-only top layer): hardware queues, weight-gradient stream, LayerNorm-backward grid
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r7; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-kernel-timing --steps 40 > $O/$tag.json 2> $O/$tag.err; python - "$tag" <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads([l for l in open(f"gpurun_out/r7/{t}.json") if l.startswith("{")][-1]); print(t, d["ms_per_step"], d["step_ms"]["median"], d.get("power",{}).get("clock_mhz"), d.get("power",{}).get("power_w"))
except Exception as e: print(t,"ERR",e)
PY
}
run base A=1
run q4 GPU_MAX_HW_QUEUES=4
run q16 GPU_MAX_HW_QUEUES=16
run nowgs SPMM_WGRAD_STREAM=0
run onestream SPMM_STREAMS=1
run lnb512 SPMM_LN_BWD_GRID=512
run lnb2048 SPMM_LN_BWD_GRID=2048
run base2 A=1
run fuse SPMM_FUSE_DROP_RES=1
