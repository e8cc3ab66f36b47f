#!/bin/bash
# Round 5, fourth GPU call: prestage and fb_smooth riding in the MODE launches, k_solve_fc = finish + meet + solve: the suite, A/B, kernel stats, bench line.
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "round5 or fb_smooth" > gpurun_out/r05d_pytest_first.log 2>&1; echo "identity rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|gave up" gpurun_out/r05d_pytest_first.log | tail -12
for wl in cfg2 cfg3; do
  timeout 300 python scripts/ab_config.py $wl "@pose_fused=0 @fb_overlap=0" "@pose_fused=1 @fb_overlap=0" "" > gpurun_out/r05d_ab_$wl.log 2>&1
  grep -E "ms/window|gave up" gpurun_out/r05d_ab_$wl.log
done
WL=cfg2 bash scripts/kstats.sh r05d_cfg2 > gpurun_out/r05d_kstats_cfg2.txt 2>&1; head -16 gpurun_out/r05d_kstats_cfg2.txt
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/r05d_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r05d_pytest.log | tail -25
timeout 600 python bench.py > gpurun_out/r05d_bench.json 2> gpurun_out/r05d_bench.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r05d_bench.json; tail -3 gpurun_out/r05d_bench.err
