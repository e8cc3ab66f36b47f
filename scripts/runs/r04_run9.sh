cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for m in "--reference_rng 1" "--reference_tex 1" "--max_iters 1 --reference_rng 1 --reference_tex 1" "--reference_rng 1 --reference_tex 1 --fb_smooth 0" "--reference_rng 1 --reference_tex 1 --optimize_depth 0"; do
VK_T_MODE="$m" timeout 600 python -m pytest tests/test_gpu_vs_ref_window.py -x -q -m gpu -k "xorwow or strict" -p no:faulthandler -s > gpurun_out/xw_dbg.log 2>&1; echo "== $m"; grep "survived\|fault\|passed\|failed" gpurun_out/xw_dbg.log | cut -c1-200
done
