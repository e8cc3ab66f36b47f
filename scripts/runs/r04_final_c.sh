#!/bin/bash
# last call of the round: whole GPU suite on the final tree, then the final profile + bench set (scripts/runs/r04_final_b.sh)
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_r04_final.log 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|passed|failed" gpurun_out/pytest_gpu_r04_final.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/runs/r04_final_b.sh
