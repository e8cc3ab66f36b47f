cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ensemble.py -q -m gpu -s > gpurun_out/ens_cap12.log 2>&1; tail -3 gpurun_out/ens_cap12.log
mkdir -p gpurun_out/cap12 && mv gpurun_out/ensemble_*.json gpurun_out/cap12/
VOLDOR_HIP_DEBUG=newton_cap=0 timeout 900 python -m pytest tests/test_gpu_ensemble.py -q -m gpu -s > gpurun_out/ens_cap0.log 2>&1; tail -3 gpurun_out/ens_cap0.log
mkdir -p gpurun_out/cap0 && mv gpurun_out/ensemble_*.json gpurun_out/cap0/
VOLDOR_HIP_DEBUG=newton_cap=0 timeout 300 python -m pytest tests/test_gpu_strict.py -q -m gpu -k config_variants -s 2>&1 | tail -4
