cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_strict.py -x -q -m gpu -k "not config_variants" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_vs_ref_window.py tests/test_gpu_vs_ref_kernels.py -x -q -m gpu -k strict 2>&1 | tail -3
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
python scripts/ab_config.py cfg2 "$R" 2>&1 | tail -2
python scripts/ab_config.py cfg3 "$R" 2>&1 | tail -1
bash scripts/kstats_cfg.sh strict4_cfg2 cfg2 "$R"
