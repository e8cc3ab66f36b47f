#!/bin/bash
# Round 5: fb_smooth riding in the pose half's mode kernels: window hashes, A/B, kernel times, suites
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=r05s
VOLDOR_HIP_DEBUG="fb_ride=0" timeout 300 python scripts/window_hash.py cfg2 cfg3 cfg5 > gpurun_out/${TAG}_hash_off.txt 2>&1; timeout 300 python scripts/window_hash.py cfg2 cfg3 cfg5 > gpurun_out/${TAG}_hash_on.txt 2>&1
grep -E "^cfg" gpurun_out/${TAG}_hash_off.txt gpurun_out/${TAG}_hash_on.txt
for wl in cfg2 cfg3; do
  timeout 700 python scripts/ab_config.py $wl "@fb_ride=0" "" "@fb_ride=0" "" > gpurun_out/${TAG}_ab_$wl.log 2>&1; grep -E "ms/window" gpurun_out/${TAG}_ab_$wl.log
done
WL=cfg2 bash scripts/kstats.sh ${TAG}_cfg2 > gpurun_out/${TAG}_kstats_cfg2.txt 2>&1; grep -E "k_pose_mode|k_fb|k_cum|k_cost_rand" gpurun_out/${TAG}_kstats_cfg2.txt; rm -rf gpurun_out/ks_${TAG}_cfg2
timeout 1200 python -m pytest tests/test_gpu_voldor.py tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_vs_ref_window.py -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG}_pytest.log | tail -8
