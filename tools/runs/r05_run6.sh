#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "dropout_and_residual" > $O/t1.txt 2>&1; tail -3 $O/t1.txt
timeout 3000 python -m pytest tests/test_step_gpu.py -x -q -s -k "storage_model or full_benchmark_batch or training_trace or projection_epilogue or published_size" > $O/t2.txt 2>&1; tail -5 $O/t2.txt
grep -E "relative L2|family rel|resid_fp32=|B = 128|\|hip -|storage model|identical|epilogue: losses" $O/t2.txt | head -120
