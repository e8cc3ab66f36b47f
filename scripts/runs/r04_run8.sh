cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_vs_ref_window.py -x -q -m gpu -k "xorwow or strict" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_ensemble.py -q -m gpu -s > gpurun_out/ens_r8.log 2>&1; grep -A12 "windows): median" gpurun_out/ens_r8.log | cut -c1-170; tail -3 gpurun_out/ens_r8.log
timeout 1200 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_ensemble.py 2>&1 | tail -15
