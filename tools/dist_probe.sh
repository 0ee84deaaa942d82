#!/bin/bash
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timing"
run() { tag=$1; shift; args=$1; shift; env "$@" $B $args 2>gpurun_out/dp_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$tag', d['step_ms']['median'], d.get('stream_placement'))"; }
p=29800
run plain "" SPMM_FORCE_DIST=0
for v in "" wgrad_only early_rccl sleep_only sides_first no_wgrad no_idle; do
  p=$((p+1)); run "probe_$v" "" SPMM_FORCE_DIST=1 SPMM_PROBE_VARIANT=$v MASTER_PORT=$p
done
p=$((p+1)); run noprobe "" SPMM_FORCE_DIST=1 SPMM_PROBE_STREAMS=0 MASTER_PORT=$p
