#!/bin/bash
# round 6, run 6: untraced phase timeline with the step's head and tail marked
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06_6; mkdir -p $O
PHASES_BOUNDARY=1 timeout 600 python tools/phase_times.py 14 > $O/phase_boundary.txt 2>&1
cat $O/phase_boundary.txt | tail -40
