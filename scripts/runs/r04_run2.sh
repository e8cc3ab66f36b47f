cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_strict.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_vs_ref_window.py tests/test_gpu_vs_ref_kernels.py -x -q -m gpu 2>&1 | tail -8
python scripts/ab_config.py cfg2 "" "@newton_cap=0" "@newton_cap=8" "@newton_cap=16" "--strict_math 1 --reference_draw 1 --reference_svd 1" 2>&1 | tail -12
bash scripts/kstats_cfg.sh strict2_cfg2 cfg2 "--strict_math 1 --reference_draw 1 --reference_svd 1"
