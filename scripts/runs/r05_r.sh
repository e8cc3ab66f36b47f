#!/bin/bash
# Round 5: the density reduction of an E-step riding in the next correspondence trace: window hashes, A/B, pose / window suites
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=r05r
VOLDOR_HIP_DEBUG="defer_reduce=0" timeout 300 python scripts/window_hash.py cfg2 cfg3 cfg5 > gpurun_out/${TAG}_hash_off.txt 2>&1; timeout 300 python scripts/window_hash.py cfg2 cfg3 cfg5 > gpurun_out/${TAG}_hash_on.txt 2>&1
grep -E "^cfg" gpurun_out/${TAG}_hash_off.txt gpurun_out/${TAG}_hash_on.txt
for wl in cfg2 cfg3 cfg5; do
  timeout 700 python scripts/ab_config.py $wl "@defer_reduce=0" "" "@defer_reduce=0" "" > gpurun_out/${TAG}_ab_$wl.log 2>&1; grep -E "ms/window" gpurun_out/${TAG}_ab_$wl.log
done
timeout 1200 python -m pytest tests/test_gpu_voldor.py tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_vs_ref_window.py tests/test_gpu_strict.py -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG}_pytest.log | tail -8
