#!/bin/bash
# round 6, run 2: overlap test diagnostics: which stream is held back in the failing processes; the step on a pool stream instead of the null stream
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06_2; mkdir -p $O
for v in null pool; do
pass=0
for i in $(seq 1 25); do
  SPMM_X_OVERLAP_MAIN=$v SPMM_TIMING_CHILD=1 GPU_MAX_HW_QUEUES=8 timeout 300 python -m pytest -x -q -s tests/test_zz_timing_gpu.py::test_gradient_exchange_overlaps_backward > $O/ov_${v}_$i.txt 2>&1 && pass=$((pass+1))
  grep "overlap-probe\] ms" $O/ov_${v}_$i.txt | tail -1 | cut -c1-140
  grep "overlap-probe\] per slice" $O/ov_${v}_$i.txt | tail -1 | cut -c1-260
done
echo "overlap test ($v main stream): $pass / 25 fresh processes passed" | tee -a $O/overlap_25x.txt
done
timeout 900 python -m pytest -x -q tests/test_kernels_gpu.py -k "device_side_row_counts" 2>&1 | tail -3
timeout 900 python -m pytest -x -q tests/test_step_gpu.py -k "eager_then_graphed or hipgraph" 2>&1 | tail -3
