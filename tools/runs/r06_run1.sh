#!/bin/bash
# round 6, run 1: the gradient-exchange overlap test 20x in fresh processes (single attempt each); one-rank data-parallel step variants
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06_1; mkdir -p $O
pass=0
for i in $(seq 1 20); do
  SPMM_TIMING_CHILD=1 GPU_MAX_HW_QUEUES=8 timeout 300 python -m pytest -x -q -s tests/test_zz_timing_gpu.py::test_gradient_exchange_overlaps_backward > $O/ov_$i.txt 2>&1 && pass=$((pass+1))
  grep "overlap-probe\] ms" $O/ov_$i.txt | tail -1
done
echo "overlap test: $pass / 20 fresh processes passed" | tee $O/overlap_20x.txt
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing --no-other-configs"
run() { name=$1; shift; env "$@" MASTER_PORT=$((29600 + RANDOM % 300)) timeout 600 $B > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[2], d["ms_per_step"], d["step_ms"], d.get("schedule_check"), d.get("comm_exposed_ms"))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run plain A=1
run dp_default SPMM_FORCE_DIST=1
run dp_bind2 SPMM_FORCE_DIST=1 SPMM_X_DP_BIND=side0,side1
run dp_offpath SPMM_FORCE_DIST=1 SPMM_X_DP_OFFPATH=1
run dp_offpath_nonexcl SPMM_FORCE_DIST=1 SPMM_X_DP_OFFPATH=1 SPMM_X_DP_EXCLUSIVE=0
run dp_nonexcl SPMM_FORCE_DIST=1 SPMM_X_DP_EXCLUSIVE=0
run dp_persistent SPMM_FORCE_DIST=1 SPMM_NT_UNDER_COMM=persistent
run plain2 A=1
