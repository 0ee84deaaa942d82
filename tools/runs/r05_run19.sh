#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r19; mkdir -p $O
python tools/phase_times.py 12 2>&1 | grep -v amdgpu.ids | tee $O/phase_times.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "decode_attention" > $O/t1.txt 2>&1; tail -3 $O/t1.txt
