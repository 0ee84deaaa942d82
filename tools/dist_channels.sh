#!/bin/bash
# One-rank RCCL group on one GPU (EngineOptions.force_dist): the step with the collectives' kernels forced to the channel counts an
# 8-GPU node uses (every channel is a workgroup that holds a CU while a slice is exchanged).  -> gpurun_out/<tag>_dist_channels.txt
tag=${1:-rXX}
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing"
out=gpurun_out/${tag}_dist_channels.txt
: > $out
run() { name=$1; shift; env "$@" $B 2>gpurun_out/dc_$name.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$name', 'ms/step', d['ms_per_step'], 'median', d['step_ms']['median'], 'p90', d['step_ms']['p90'], d.get('stream_placement'))" >> $out; }
p=29700
run single_rank_no_group MASTER_PORT=$p
for ch in default 8 16 32 64; do
  p=$((p+1))
  if [ $ch = default ]; then run rccl_1rank_channels_default SPMM_FORCE_DIST=1 MASTER_PORT=$p
  else run rccl_1rank_min_channels_$ch SPMM_FORCE_DIST=1 NCCL_MIN_NCHANNELS=$ch MASTER_PORT=$p; fi
done
p=$((p+1)); run rccl_1rank_no_overlap SPMM_FORCE_DIST=1 SPMM_GRAD_OVERLAP=0 MASTER_PORT=$p
p=$((p+1)); run rccl_1rank_bf16_wire SPMM_FORCE_DIST=1 SPMM_GRAD_WIRE=bf16 MASTER_PORT=$p
cat $out
