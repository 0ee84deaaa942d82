#!/bin/bash
# one-rank RCCL group: cost of the data-parallel code path with and without the start-up stream probe
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timing"
run() { tag=$1; shift; env "$@" $B 2>gpurun_out/dp_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$tag', d['step_ms']['median'], d.get('stream_placement'))"; }
run plain SPMM_FORCE_DIST=0
run probe SPMM_FORCE_DIST=1 MASTER_PORT=29601
run noprobe SPMM_FORCE_DIST=1 SPMM_PROBE_STREAMS=0 MASTER_PORT=29602
run probe_q4 SPMM_FORCE_DIST=1 GPU_MAX_HW_QUEUES=4 MASTER_PORT=29603
run noprobe_q4 SPMM_FORCE_DIST=1 SPMM_PROBE_STREAMS=0 GPU_MAX_HW_QUEUES=4 MASTER_PORT=29604
run probe_persist SPMM_FORCE_DIST=1 SPMM_NT_UNDER_COMM=persistent MASTER_PORT=29605
