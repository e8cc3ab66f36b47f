cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x -s > gpurun_out/full_suite.log 2>&1; grep -v "^$" gpurun_out/full_suite.log | tail -25 | cut -c1-250
