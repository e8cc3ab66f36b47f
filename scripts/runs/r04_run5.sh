cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_strict.py::test_config_variants_fast_vs_strict -s 2>&1 | grep -v "^$" | tail -60
timeout 300 python -m pytest tests/test_gpu_strict.py -q -m gpu -k config_variants -s 2>&1 | tail -12
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
python scripts/ab_config.py cfg2 "$R" 2>&1 | tail -1
bash scripts/kstats_cfg.sh strict5_cfg2 cfg2 "$R" | head -8
