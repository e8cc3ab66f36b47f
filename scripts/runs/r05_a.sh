#!/bin/bash
# Round 5, first GPU call: the suite on the cleaned tree (fb_smooth on the second stream, fp32 P3P, strict hand-over), A/B of the two fast-path changes,
# kernel stats of cfg2, the default bench line.
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05a_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r05a_pytest.log
for wl in cfg2 cfg3 cfg5; do
  timeout 300 python scripts/ab_config.py $wl "@fb_overlap=0 @solve_fp32=0" "@fb_overlap=1 @solve_fp32=0" "@fb_overlap=0 @solve_fp32=1" "" > gpurun_out/r05a_ab_$wl.log 2>&1
  grep "ms/window" gpurun_out/r05a_ab_$wl.log
done
WL=cfg2 bash scripts/kstats.sh r05a_cfg2 > gpurun_out/r05a_kstats_cfg2.txt 2>&1; cat gpurun_out/r05a_kstats_cfg2.txt | head -24
timeout 600 python bench.py > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/r05a_bench.json; tail -3 gpurun_out/r05a_bench.err
