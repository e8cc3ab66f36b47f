#!/bin/bash
# Phase clocks (s_memtime) of the pose kernels: builds a second library with -DVK_PHASE_CLOCKS next to the product one and runs
# N windows of a workload through it.  Build part runs anywhere (hipcc cross-compiles); `run` needs a GPU.
#   scripts/phase_clocks.sh build ; gpurun -- 'scripts/phase_clocks.sh run cfg2 > gpurun_out/phase.txt'
set -e
cd "$(dirname "$0")/.."
LIB=voldor_amd/lib/libvoldor_hip_phase${PHASE_TAG}.so   # PHASE_TAG / PHASE_FLAGS: variants side by side
if [ "$1" = build ]; then
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -DVK_PHASE_CLOCKS -Iinclude $PHASE_FLAGS"
  objs=""
  for f in vk_abi vk_depth vk_pose vk_strict vk_bootstrap vk_voldor vk_slam vk_align vk_dist; do
    extra=""; case $f in vk_pose|vk_bootstrap|vk_strict) extra="-ffp-contract=off";; esac
    /opt/rocm/bin/hipcc $F $extra -c voldor_amd/csrc/$f.hip -o /tmp/phase${PHASE_TAG}_$f.o &
    objs="$objs /tmp/phase${PHASE_TAG}_$f.o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $LIB $objs -ldl
  echo built $LIB
else
  VOLDOR_HIP_LIB=$PWD/$LIB python scripts/phase_clocks.py "${2:-cfg2}"
fi
