#!/bin/bash
# Kernel trace of the data-parallel code path with a one-rank RCCL group (SPMM_FORCE_DIST=1): per-kernel totals and the idle gaps.
#   bash tools/profile_dist.sh <tag>
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SPMM_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
d=/tmp/prof_${tag}_dist
rm -rf $d
rocprofv3 --kernel-trace --stats -d $d -o p -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_${tag}_dist.log 2>&1
python3 tools/rocpd_stats.py $d/p_results.db 9 > gpurun_out/${tag}_kernel_stats_dist.txt 2>&1
python3 tools/trace_gaps.py $d/p_results.db > gpurun_out/${tag}_gaps_dist.txt 2>&1
python3 tools/trace_window.py $d/p_results.db oneRankReduce 8 400 900 > gpurun_out/${tag}_window_dist.txt 2>&1
