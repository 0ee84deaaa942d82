#!/bin/bash
# round 5, profile set of the final tree: PMC passes, kernel traces (multi / single stream), roofline + power tables, the default bench line
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/pmc_step.sh gpurun_out/final/pmc > gpurun_out/final/pmc.log 2>&1
python3 tools/pmc_step_summary.py gpurun_out/final/pmc gpurun_out/final/r05 > gpurun_out/final/pmc_summary.log 2>&1
bash tools/profile_step.sh r05 > gpurun_out/final/profile.log 2>&1
ROOFLINE_POWER=1 timeout 900 python tools/roofline_table.py > gpurun_out/final/roofline_power.txt 2>&1
timeout 1500 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
ls gpurun_out/final gpurun_out | head -40
head -30 gpurun_out/r05_kernel_stats_streams1.txt
tail -c 1500 gpurun_out/final/bench_default.json
# decoder: kernel trace of one short decode run (stats only)
d=/tmp/prof_r05_dec; rm -rf $d
rocprofv3 --kernel-trace --stats -d $d -o p -- python3 bench.py --decode --no-cpu-baseline --molecules 1000 > gpurun_out/final/prof_decode.log 2>&1
python3 tools/rocpd_stats.py $d/p_results.db 1 > gpurun_out/final/r05_decode_kernel_stats.txt 2>&1; head -16 gpurun_out/final/r05_decode_kernel_stats.txt
python tools/phase_times.py 12 > gpurun_out/final/phase_times.txt 2>&1; tail -12 gpurun_out/final/phase_times.txt
