#!/bin/bash
# One-rank RCCL group (SPMM_FORCE_DIST=1): what the data-parallel code path itself costs on one GPU, piece by piece.  With one rank
# RCCL's mean all-reduce is a real kernel (oneRankReduce<FuncPreMulSum>, ~0.25 ms per 30-MB slice) on RCCL's stream.
D="SPMM_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
run() { echo "== $*"; env "$@" timeout 300 python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>&1 | grep metric | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run A=1
run $D MASTER_PORT=29541
run $D MASTER_PORT=29543 SPMM_NT_UNDER_COMM=persistent
run $D MASTER_PORT=29544 SPMM_WGRAD_UNDER_COMM=1
run $D MASTER_PORT=29542 SPMM_WGRAD_UNDER_COMM=1 SPMM_CHAIN_EXCHANGE=0
run $D MASTER_PORT=29546 SPMM_WGRAD_UNDER_COMM=1 GPU_MAX_HW_QUEUES=4
run $D MASTER_PORT=29545 SPMM_GRAD_OVERLAP=0
run $D MASTER_PORT=29547 SPMM_GRAD_WIRE=bf16
