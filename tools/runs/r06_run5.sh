#!/bin/bash
# round 6, run 5: kernel timeline around the step boundary (last 3 ms, first 2.5 ms), multi-stream default schedule
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06_5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
d=/tmp/prof_r06_tl; rm -rf $d
rocprofv3 --kernel-trace --stats -d $d -o p -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-other-configs > $O/prof.log 2>&1
python3 tools/step_timeline.py $d/p_results.db 1.0 3.0 2.5 > $O/timeline.txt 2>&1
head -12 $O/timeline.txt
