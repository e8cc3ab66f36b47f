#!/bin/bash
# strict / reference-mode test files, then reference-mode timings and kernel stats (cfg2)
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_strict.py tests/test_gpu_vs_ref_window.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -4
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
timeout 600 python scripts/ab_config.py cfg2 "$R" 2>&1 | tail -1
timeout 600 python scripts/ab_config.py cfg3 "$R" 2>&1 | tail -1
bash scripts/kstats_cfg.sh r04_strict3_cfg2 cfg2 "$R" > gpurun_out/ks_r04_strict3_cfg2.txt 2>&1; head -4 gpurun_out/ks_r04_strict3_cfg2.txt; rm -rf gpurun_out/ks_r04_strict3_cfg2
