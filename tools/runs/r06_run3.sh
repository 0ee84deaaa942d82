#!/bin/bash
# round 6, run 3: overlap test after the warm-up step (25 fresh processes), new tests
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06_3; mkdir -p $O
pass=0
for i in $(seq 1 25); do
  SPMM_TIMING_CHILD=1 GPU_MAX_HW_QUEUES=8 timeout 300 python -m pytest -x -q -s tests/test_zz_timing_gpu.py::test_gradient_exchange_overlaps_backward > $O/ov_$i.txt 2>&1 && pass=$((pass+1))
  grep "overlap-probe\] ms" $O/ov_$i.txt | tail -1 | cut -c1-220
done
echo "overlap test: $pass / 25 fresh processes passed" | tee -a $O/overlap_25x.txt
timeout 900 python -m pytest -x -q tests/test_kernels_gpu.py -k "device_side_row_counts" 2>&1 | tail -40
timeout 900 python -m pytest -x -q -s tests/test_step_gpu.py -k "wrong_token or published_size" 2>&1 | tail -30
