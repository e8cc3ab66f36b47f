#!/bin/bash
# round 4: D4 switch (--reference_stale_depth) + Camera::rvec() round trip in reference mode: whole GPU suite, then timings
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r04_run14.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu_r04_run14.log
timeout 600 python scripts/ab_config.py cfg2 "" "--strict_math 1 --reference_draw 1 --reference_svd 1" "--strict_math 1 --reference_draw 1 --reference_svd 1 --reference_stale_depth 1" 2>&1 | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
