#!/bin/bash
# per-queue timeline of a steady-state step (default three-stream schedule)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r17; mkdir -p $O
d=/tmp/prof_tl; rm -rf $d
rocprofv3 --kernel-trace -d $d -o p -- python3 bench.py --steps 8 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-other-configs > $O/prof.log 2>&1
tail -2 $O/prof.log | cut -c1-300
python3 tools/step_timeline.py $d/p_results.db 1.0 > $O/timeline.txt 2>&1
cat $O/timeline.txt
python3 tools/idle_gaps.py $d/p_results.db 0.5 15 > $O/idle.txt 2>&1; head -30 $O/idle.txt
