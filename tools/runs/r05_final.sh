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
