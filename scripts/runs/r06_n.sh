#!/bin/bash
# Round 6: fb_smooth riding in every stack's own segments (40-step at 1080p; the rigidness maps alone where the prior confidences do not fit): tests, hashes, A/B
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${TAG:-r06n}
timeout 1500 python -m pytest tests/test_gpu_riders.py tests/test_fb_ride_plan.py -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${T}_pytest.log | tail -12
VOLDOR_HIP_DEBUG="bootstrap_default=8" timeout 300 python scripts/window_hash.py cfg2 cfg3 cfg5 > gpurun_out/${T}_hash.txt 2>&1; grep -E "^cfg" gpurun_out/${T}_hash.txt
for wl in cfg5 cfg3 cfg2; do
  timeout 600 python scripts/ab_config.py $wl "@fb_ride=0" "" "@fb_ride=0" "" > gpurun_out/${T}_ab_$wl.log 2>&1; grep -E "ms/window" gpurun_out/${T}_ab_$wl.log
done
WL=cfg5 bash scripts/kstats.sh ${T}_cfg5 > gpurun_out/${T}_kstats_cfg5.txt 2>&1; grep -E "k_pose_mode|k_fb|k_cum|k_cost_rand" gpurun_out/${T}_kstats_cfg5.txt; rm -rf gpurun_out/ks_${T}_cfg5
