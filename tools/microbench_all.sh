#!/bin/bash
# All stand-alone kernel measurements of a round into gpurun_out/<tag>_*.txt:  bash tools/microbench_all.sh <tag>
tag=${1:-rXX}
python tools/roofline_table.py > gpurun_out/${tag}_roofline.txt 2>&1
build/gemm_bench step > gpurun_out/${tag}_gemm_bench_step.txt 2>&1
build/gemm_bench tnstep >> gpurun_out/${tag}_gemm_bench_step.txt 2>&1
: > gpurun_out/${tag}_fp8_gemm.txt
for s in "84256 3072 768 6" "84256 3072 768 0" "84256 768 3072 0" "84256 768 768 0" "4096 4096 4096 0" "8192 8192 8192 0 3"; do
  build/gemm_bench f8time $s >> gpurun_out/${tag}_fp8_gemm.txt 2>&1
done
