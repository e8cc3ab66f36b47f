#!/bin/bash
# Round 5: runtime knobs of the HIP / ROCr stack against the dependent-launch boundary (one window = ~300 dependent launches)
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=r05o
run() { echo "== $1"; env $1 timeout 300 python scripts/ab_config.py cfg2 "" 2>&1 | grep "ms/window"; }
run "VK_NONE=1"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "HSA_ENABLE_INTERRUPT=0"
run "GPU_MAX_HW_QUEUES=1"
run "HSA_ENABLE_SDMA=0"
run "AMD_DIRECT_DISPATCH=0"
run "HIP_LAUNCH_BLOCKING=0 HSA_NO_SCRATCH_RECLAIM=1"
run "VK_NONE=2"
