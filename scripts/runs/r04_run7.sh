cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused_local or local_runs" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_strict.py -x -q -m gpu -k "parallel_structures or window_bits" 2>&1 | tail -3
for wl in cfg2 cfg3 cfg5; do python scripts/ab_config.py $wl "@local_fused=0" "@local_fused=8" "@local_fused=16" 2>&1 | tail -3; done
python scripts/ab_config.py cfg2 "--strict_math 1 --reference_draw 1 --reference_svd 1 @local_fused=0" "--strict_math 1 --reference_draw 1 --reference_svd 1 @local_fused=8" "--strict_math 1 --reference_draw 1 --reference_svd 1 @local_fused=16"  2>&1 | tail -3
