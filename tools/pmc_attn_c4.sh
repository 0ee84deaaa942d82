#!/bin/bash
# SQ busy / wait counters of the attention kernels at BASELINE configs[4]'s per-GPU shape (B=512, Lt=256): one rocprofv3 --pmc pass, own run,
# no trace domains, the program directly after `--`.   bash tools/pmc_attn_c4.sh OUTDIR ; python3 tools/pmc_summary.py ...
set -e
export TMPDIR=/tmp
out=$1
mkdir -p "$out"
SPMM_STREAMS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-include-regex "attn_" -d "$out/sq" -o p -f csv -- python3 bench.py --batch 512 --seq-len 256 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > "$out/sq.log" 2>&1 || tail -3 "$out/sq.log"
ls "$out"/sq/ | head
