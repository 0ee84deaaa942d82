#!/bin/bash
# Kernel-trace summaries of the benchmark step, multi-stream (default schedule) and single-stream: text only under gpurun_out/.
#   bash tools/profile_step.sh <tag>
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in 2 1; do
  d=/tmp/prof_${tag}_s$s
  rm -rf $d
  SPMM_STREAMS=$s rocprofv3 --kernel-trace --stats -d $d -o p -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_${tag}_s$s.log 2>&1
  python3 tools/rocpd_stats.py $d/p_results.db 9 > gpurun_out/${tag}_kernel_stats_streams$s.txt 2>&1
done
python3 tools/roofline_table.py > gpurun_out/${tag}_roofline.txt 2>&1
