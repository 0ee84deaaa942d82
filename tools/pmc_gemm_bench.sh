#!/bin/bash
# rocprofv3 counter passes over build/gemm_bench (program directly after `--`, counters in their own run, no trace domains).
# usage: tools/pmc_gemm_bench.sh OUTDIR M N K EPI     -> OUTDIR/pass{1,2}/... csv ; summary printed by tools/pmc_csv.py
set -e
export TMPDIR=/tmp
out=$1; shift
mkdir -p "$out"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-include-regex gemm_nt -d "$out/pass1" -o p -f csv -- build/gemm_bench time "$@" 1 > "$out/pass1.log" 2>&1 || tail -5 "$out/pass1.log"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS \
  --kernel-include-regex gemm_nt -d "$out/pass2" -o p -f csv -- build/gemm_bench time "$@" 1 > "$out/pass2.log" 2>&1 || tail -5 "$out/pass2.log"
python3 tools/pmc_csv.py "$out"
