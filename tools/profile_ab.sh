#!/bin/bash
# Kernel-trace A/B of one option on the single-stream schedule:  bash tools/profile_ab.sh <tag> <ENV_VAR>   (ENV_VAR=0 vs 1)
tag=$1; var=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 1; do
  d=/tmp/prof_${tag}_$v; rm -rf $d
  export $var=$v SPMM_STREAMS=1
  rocprofv3 --kernel-trace -d $d -o p -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_${tag}_$v.log 2>&1
  python3 tools/rocpd_stats.py $d/p_results.db 9 > gpurun_out/${tag}_kernel_stats_${var}_$v.txt 2>&1
done
