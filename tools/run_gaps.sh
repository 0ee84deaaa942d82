cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/pg -o p -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
python3 tools/trace_gaps.py /tmp/pg/p_results.db | head -8
python3 tools/trace_small_grids.py /tmp/pg/p_results.db
