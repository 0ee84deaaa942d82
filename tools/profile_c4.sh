#!/bin/bash
# Kernel-trace summary of the configs[4] per-GPU shape (B=512, Lt=256), single-stream.   bash tools/profile_c4.sh <tag> [extra bench flags]
tag=${1:-rXX}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
d=/tmp/prof_${tag}_c4
rm -rf $d
SPMM_STREAMS=1 rocprofv3 --kernel-trace --stats -d $d -o p -- python3 bench.py --batch 512 --seq-len 256 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > gpurun_out/prof_${tag}_c4.log 2>&1
python3 tools/rocpd_stats.py $d/p_results.db 3 > gpurun_out/${tag}_c4_kernel_stats_single_stream.txt 2>&1
