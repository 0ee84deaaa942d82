#!/bin/bash
# Two gloo ranks sharing one GPU (the configuration of test_two_rank_data_parallel_step_on_one_gpu) with a watchdog that dumps the
# Python stacks if the run hangs:   bash tools/two_rank_one_gpu.sh [VAR=value ...]
port=$((29600 + RANDOM % 300))
env "$@" SPMM_BENCH_WATCHDOG=${WATCHDOG:-60} SPMM_DIST_BACKEND=gloo timeout 150 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 --seq-len 32 --layers 2,1,1 --queue 64 \
  --no-cpu-baseline --check-replicas 2>&1 | grep -E "replicas identical|Timeout|File \"/root/repo|metric|\[rank|\[launch" | cut -c1-160
