#!/bin/bash
# One-rank RCCL group: step time of the data-parallel code path against the number of HIP hardware queues.
D="SPMM_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
run() { echo "== $*"; env "$@" timeout 300 python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>&1 | grep metric | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
p=29550
for q in 4 12 16 24; do p=$((p+1)); run $D MASTER_PORT=$p GPU_MAX_HW_QUEUES=$q; done
p=$((p+1)); run $D MASTER_PORT=$p SPMM_STREAMS=1
p=$((p+1)); run $D MASTER_PORT=$p NCCL_MAX_NCHANNELS=4
