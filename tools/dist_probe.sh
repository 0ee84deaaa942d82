#!/bin/bash
# One-rank RCCL group on one GPU (EngineOptions.force_dist): step time with and without the start-up stream probe, with and without the
# overlapped gradient exchange.  EXPERIMENTS.md 1.4.
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timing"
run() { tag=$1; shift; env "$@" $B 2>gpurun_out/dp_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$tag', d['step_ms']['median'], d.get('stream_placement'))"; }
p=29900
p=$((p+1)); run "noprobe" SPMM_FORCE_DIST=1 MASTER_PORT=$p
p=$((p+1)); run "probe" SPMM_FORCE_DIST=1 SPMM_PROBE_STREAMS=1 MASTER_PORT=$p
p=$((p+1)); run "noprobe_no_overlap" SPMM_FORCE_DIST=1 SPMM_GRAD_OVERLAP=0 MASTER_PORT=$p
p=$((p+1)); run "probe_no_overlap" SPMM_FORCE_DIST=1 SPMM_PROBE_STREAMS=1 SPMM_GRAD_OVERLAP=0 MASTER_PORT=$p
