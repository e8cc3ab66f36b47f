#!/bin/bash
# k_solve_mode (P3P batch + mode kernel in one launch): identity test, hashes with the switch on / off, timings, kernel stats
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_solve_and_mode" 2>&1 | tail -6
for v in 1 0; do echo "== fuse_solve_mode=$v"; VOLDOR_HIP_DEBUG=fuse_solve_mode=$v timeout 600 python scripts/window_hash.py cfg2 cfg3 cfg5 2>&1 | tail -3; done
for r in 1 2; do for wl in cfg2 cfg3 cfg5; do timeout 600 python scripts/ab_config.py $wl "@fuse_solve_mode=1" "@fuse_solve_mode=0" 2>&1 | tail -2; done; done
WL=cfg2 bash scripts/kstats.sh r04_fused_cfg2 > gpurun_out/ks_r04_fused_cfg2.txt 2>&1; head -14 gpurun_out/ks_r04_fused_cfg2.txt; rm -rf gpurun_out/ks_r04_fused_cfg2
