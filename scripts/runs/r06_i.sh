#!/bin/bash
# Round 6: fb_smooth on the side stream (strict mode, 40-step segments) + five-point default: identity tests, hashes, A/B
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${TAG:-r06i}
timeout 1500 python -m pytest tests/test_gpu_riders.py tests/test_fivept.py tests/test_gpu_voldor.py tests/test_gpu_strict.py tests/test_gpu_ensemble.py -m gpu -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${T}_pytest.log | tail -12
VOLDOR_HIP_DEBUG="bootstrap_default=8" timeout 300 python scripts/window_hash.py cfg2 cfg3 cfg5 > gpurun_out/${T}_hash.txt 2>&1; grep -E "^cfg" gpurun_out/${T}_hash.txt
timeout 300 python scripts/window_hash.py cfg2 > gpurun_out/${T}_hash5.txt 2>&1; grep -E "^cfg" gpurun_out/${T}_hash5.txt
timeout 600 python scripts/ab_config.py cfg5 "@fb_side=0" "" "@fb_side=0" "" > gpurun_out/${T}_ab_cfg5.log 2>&1; grep -E "ms/window" gpurun_out/${T}_ab_cfg5.log
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
for wl in cfg2 cfg3; do
  timeout 900 python scripts/ab_config.py $wl "@fb_side=0 $R" "$R" > gpurun_out/${T}_ab_strict_$wl.log 2>&1; grep -E "ms/window" gpurun_out/${T}_ab_strict_$wl.log
done
