#!/bin/bash
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_reference_cv.py -m gpu -q 2>&1 | tail -5
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_r04_run15.log 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|passed|failed" gpurun_out/pytest_gpu_r04_run15.log | cut -c1-200 | tail -30
