#!/bin/bash
# round 5, GPU run 1: clock / power sources, MFMA 32x32x16 timing experiment, baseline bench with clock fields, power table per kernel family
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r1; mkdir -p $O
python tools/smi_sampler.py > $O/smi_probe.txt 2>&1
for b in gemm_bench_static gemm_bench_m32; do
  echo "=== $b"
  for shape in "84256 768 768 0" "84256 2304 768 0" "84256 768 3072 0" "84256 3072 768 6" "84256 3072 768 7" "4096 4096 4096 0"; do
    timeout 120 build/$b time $shape 5
    GEMM_BENCH_BETWEEN=1 timeout 120 build/$b sustain $shape 300
  done
done > $O/gemm_m32.txt 2>&1
timeout 600 python bench.py --no-other-configs --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err
timeout 900 python -m pytest tests/test_step_gpu.py -x -q -k "schedule_check" > $O/test_sched.txt 2>&1
ROOFLINE_POWER=1 timeout 900 python tools/roofline_table.py > $O/roofline_power.txt 2>&1
cp gpurun_out/power_table.txt gpurun_out/roofline_table.md $O/ 2>/dev/null
tail -3 $O/smi_probe.txt; tail -30 $O/gemm_m32.txt; tail -c 600 $O/bench_base.json; tail -3 $O/test_sched.txt; tail -5 $O/roofline_power.txt
