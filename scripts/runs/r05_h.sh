#!/bin/bash
# Round 5: E-step with all gathers in flight (small images), fb_smooth with 8 / 12-step segments; the strict suite (hand-over of the cooperative mode kernel)
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=r05h
timeout 900 python -m pytest tests/test_gpu_strict.py -m gpu -q -x > gpurun_out/${TAG}_pytest_strict.log 2>&1; echo "strict rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG}_pytest_strict.log | tail -5
timeout 400 python scripts/ab_config.py cfg2 "@estep_allfly=0" "" "@fb_segment=12" "@fb_segment=8" > gpurun_out/${TAG}_ab_cfg2.log 2>&1; grep -E "ms/window" gpurun_out/${TAG}_ab_cfg2.log
timeout 400 python scripts/ab_config.py cfg3 "@estep_allfly=0" "" "@fb_segment=12" "@fb_segment=8" > gpurun_out/${TAG}_ab_cfg3.log 2>&1; grep -E "ms/window" gpurun_out/${TAG}_ab_cfg3.log
VOLDOR_HIP_DEBUG="fb_segment=12" WL=cfg2 bash scripts/kstats.sh ${TAG}_cfg2_fb12 > gpurun_out/${TAG}_kstats_cfg2_fb12.txt 2>&1; grep -E "k_fb_|k_update_rig" gpurun_out/${TAG}_kstats_cfg2_fb12.txt
VOLDOR_HIP_DEBUG="fb_segment=8" WL=cfg2 bash scripts/kstats.sh ${TAG}_cfg2_fb8 > gpurun_out/${TAG}_kstats_cfg2_fb8.txt 2>&1; grep -E "k_fb_|k_update_rig" gpurun_out/${TAG}_kstats_cfg2_fb8.txt
VOLDOR_HIP_DEBUG="estep_allfly=0" WL=cfg2 bash scripts/kstats.sh ${TAG}_cfg2_e0 > gpurun_out/${TAG}_kstats_cfg2_e0.txt 2>&1; grep -E "k_fb_|k_update_rig" gpurun_out/${TAG}_kstats_cfg2_e0.txt
rm -rf gpurun_out/ks_${TAG}_cfg2_fb12 gpurun_out/ks_${TAG}_cfg2_fb8 gpurun_out/ks_${TAG}_cfg2_e0
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_voldor.py -m gpu -q > gpurun_out/${TAG}_pytest_k.log 2>&1; echo "kernels rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG}_pytest_k.log | tail -5
