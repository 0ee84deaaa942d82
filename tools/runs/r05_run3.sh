#!/bin/bash
# round 5, GPU run 3: kernel traces of the step (multi / single stream) after the CLS-only top layer + plan kernels
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
bash tools/profile_step.sh r05a > gpurun_out/r05a_profile.log 2>&1
head -60 gpurun_out/r05a_kernel_stats_streams1.txt
