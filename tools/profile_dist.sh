#!/bin/bash
# Kernel trace of the data-parallel code path with a one-rank RCCL group: which hardware queue every stream landed on and what
# every kernel costs, with (slow) and without the start-up probe's stream order.   bash tools/profile_dist.sh <tag>
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SPMM_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for v in probe noprobe; do
  d=/tmp/prof_${tag}_$v; rm -rf $d
  if [ $v = noprobe ]; then export SPMM_PROBE_STREAMS=0; fi
  export MASTER_PORT=$((29571 + ${#v}))
  rocprofv3 --kernel-trace -d $d -o p -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_${tag}_$v.log 2>&1
  python3 tools/trace_queues.py $d/p_results.db 9 > gpurun_out/${tag}_dist_queues_$v.txt 2>&1
  grep -o '"median": [0-9.]*' gpurun_out/prof_${tag}_$v.log | head -1 >> gpurun_out/${tag}_dist_queues_$v.txt
done
