#!/bin/bash
# rocprofv3 counter passes over one training step of bench.py (B=128, Lt=128, full depth): HBM-side traffic of the NT GEMM
# family (FETCH_SIZE / WRITE_SIZE, one pass each: they do not fit one pass) and the SQ busy / wait counters of the GEMM and
# attention kernels.  Counters in their own runs, no trace domains, the program directly after `--`.
#   tools/pmc_step.sh OUTDIR      then      python3 tools/pmc_step_summary.py OUTDIR profiles/r02
set -e
export TMPDIR=/tmp
out=$1
mkdir -p "$out"
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing"
SPMM_STREAMS=1 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "gemm_nt" -d "$out/fetch" -o p -f csv -- $B > "$out/fetch.log" 2>&1 || tail -3 "$out/fetch.log"
SPMM_STREAMS=1 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "gemm_nt" -d "$out/write" -o p -f csv -- $B > "$out/write.log" 2>&1 || tail -3 "$out/write.log"
SPMM_STREAMS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-include-regex "gemm_nt|gemm_tn|attn_" -d "$out/sq" -o p -f csv -- $B > "$out/sq.log" 2>&1 || tail -3 "$out/sq.log"
ls "$out"/*/ | head -20
