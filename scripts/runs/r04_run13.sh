cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_strict.py -x -q -m gpu -k "parallel_tree or parallel_structures or window_bits or cfg2" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_fivept.py tests/test_gpu_vs_ref_window.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -8 | cut -c1-250
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
python scripts/ab_config.py cfg2 "$R" "$R @strict_pose_coop=0" 2>&1 | tail -2
bash scripts/kstats_cfg.sh strict6_cfg2 cfg2 "$R" | head -7
