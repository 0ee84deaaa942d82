#!/bin/bash
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timing"
run() { tag=$1; shift; args=$1; shift; env "$@" $B $args 2>gpurun_out/dp_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$tag', d['step_ms']['median'], d.get('stream_placement'))"; }
p=29900
for v in measure_after measure_after; do
  p=$((p+1)); run "probe_$v" "" SPMM_FORCE_DIST=1 SPMM_PROBE_VARIANT=$v MASTER_PORT=$p
done
p=$((p+1)); run "noov" "" SPMM_FORCE_DIST=1 SPMM_PROBE_STREAMS=0 SPMM_GRAD_OVERLAP=0 MASTER_PORT=$p
p=$((p+1)); run "probe_noov" "" SPMM_FORCE_DIST=1 SPMM_GRAD_OVERLAP=0 MASTER_PORT=$p
