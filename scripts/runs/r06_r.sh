#!/bin/bash
# Round 6: cooperative strict mode kernel with four waves per 512-row block: identity / bit tests, reference-mode window times, kernel time
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${TAG:-r06r}
timeout 1800 python -m pytest tests/test_gpu_strict.py tests/test_gpu_vs_ref_window.py tests/test_gpu_configs.py tests/test_gpu_vs_ref_kernels.py -m gpu -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${T}_pytest.log | tail -8
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
for wl in cfg2 cfg3 cfg5; do
  timeout 900 python scripts/ab_config.py $wl "$R" "$R" > gpurun_out/${T}_strict_$wl.log 2>&1; grep -E "ms/window" gpurun_out/${T}_strict_$wl.log | tail -2
done
bash scripts/kstats_cfg.sh ${T}_strict_cfg2 cfg2 "$R" > gpurun_out/${T}_kstats_strict_cfg2.txt 2>&1; head -6 gpurun_out/${T}_kstats_strict_cfg2.txt; rm -rf gpurun_out/ks_${T}_strict_cfg2
