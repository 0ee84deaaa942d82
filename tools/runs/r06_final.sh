#!/bin/bash
# round 6, profile set of the final tree: PMC passes, kernel traces (multi / single stream), roofline + power tables, decoder kernel stats,
# phase timeline (incl. head / tail), the default bench line.  Everything lands under gpurun_out/final/; what is judged is copied to profiles/.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/pmc_step.sh gpurun_out/final/pmc > gpurun_out/final/pmc.log 2>&1
python3 tools/pmc_step_summary.py gpurun_out/final/pmc gpurun_out/final/r06 > gpurun_out/final/pmc_summary.log 2>&1
bash tools/profile_step.sh r06 > gpurun_out/final/profile.log 2>&1
ROOFLINE_POWER=1 timeout 900 python tools/roofline_table.py > gpurun_out/final/roofline_power.txt 2>&1
d=/tmp/prof_r06_dec; rm -rf $d
rocprofv3 --kernel-trace --stats -d $d -o p -- python3 bench.py --decode --no-cpu-baseline --molecules 1000 > gpurun_out/final/prof_decode.log 2>&1
python3 tools/rocpd_stats.py $d/p_results.db 1 > gpurun_out/final/r06_decode_kernel_stats.txt 2>&1
PHASES_BOUNDARY=1 python tools/phase_times.py 12 > gpurun_out/final/phase_times.txt 2>&1
timeout 1500 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
ls gpurun_out/final gpurun_out | head -50
head -24 gpurun_out/r06_kernel_stats_streams1.txt
tail -14 gpurun_out/final/phase_times.txt
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/final/bench_default.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_at_clock"), d["roofline"].get("traffic"), d["cross_attention"]["executed_frac_of_bf16_peak"], d["cross_attention"]["ms_per_step"], d["cross_attention"].get("other_form_ms_per_step"))
PY
