#!/bin/bash
# strict mode kernel: tagged block sums instead of counter + fences.  strict tests, then timings
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_strict.py tests/test_gpu_vs_ref_window.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python scripts/ab_config.py cfg2 "--strict_math 1 --reference_draw 1 --reference_svd 1" 2>&1 | tail -2
timeout 600 python scripts/ab_config.py cfg3 "--strict_math 1 --reference_draw 1 --reference_svd 1" 2>&1 | tail -2
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
bash scripts/kstats_cfg.sh r04_strict2_cfg2 cfg2 "$R" > gpurun_out/ks_r04_strict2_cfg2.txt 2>&1; head -6 gpurun_out/ks_r04_strict2_cfg2.txt; rm -rf gpurun_out/ks_r04_strict2_cfg2
