#!/bin/bash
# Clocks / power of the one-rank data-parallel step in its fast (no probe) and slow (start-up probe) stream modes.  EXPERIMENTS.md 2.7b
cd $GRAFT_REPO_ROOT
for probe in 0 1; do
  echo "=== SPMM_PROBE_STREAMS=$probe"
  SPMM_FORCE_DIST=1 SPMM_PROBE_STREAMS=$probe SPMM_SCHEDULE_CHECK=0 MASTER_PORT=$((29810+probe)) python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/smi_mode_$probe.json 2>/dev/null &
  BP=$!
  sleep 25
  for i in 1 2 3 4; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | head -6
    echo ---
    sleep 2
  done
  wait $BP
  grep "^{" gpurun_out/smi_mode_$probe.json | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'])"
done
