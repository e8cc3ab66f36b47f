"""The HIP kernels against the REFERENCE's own kernel outputs (tests/golden/ref_kernels.npz: the reference's .cu files run
on the CPU by oracle/ref_wrap_kernels.cpp, see tests/golden/gen_golden_kernels.py), through the C-ABI, on the seeded inputs
of tests/ref_kernel_cases.py.  No oracle in between: this is product vs reference code.

Bars (same rules as the oracle-mediated stage tests in test_gpu_kernels.py): geometry / validity decisions /
correspondence maps identical up to border rounding; Fisk-model values (rigidness, cost-driven decisions) to the accuracy
of the hardware log/exp against glibc powf -- a depth search is a chain of argmin decisions, so the share of pixels taking
the same branch is asserted, and values are compared on those pixels; solver translations on the bulk of the hypotheses
(minimal solvers are ill-conditioned on some 4-tuples); iterative estimators to float summation order.
"""
import os

import numpy as np
import pytest

import ref_kernel_cases as cases

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_kernels.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize("name,maps,s0,p", list(cases.fb_cases()), ids=[c[0] for c in cases.fb_cases()])
def test_fb_smooth_vs_reference_kernel(gold, name, maps, s0, p):
    from voldor_amd import kernels
    rc, out = kernels.fb_smooth_gpu(maps, s0, p)
    assert rc == 0
    # D7: same recurrence evaluated as composed projective maps with v_rcp_f32; contractive, a few ulp per step
    assert np.abs(out - gold[f"fb/{name}"]).max() < 2e-5


def _run_depth(c):
    from voldor_amd import kernels
    kw = c["kw"]
    N, h, w, _ = c["flows"].shape
    N_dp = 0 if c["priors"] is None else c["priors"].shape[0]
    K = np.asarray(c["K"], np.float32).reshape(3, 3)
    kernels.set_rand_epoch(c["rand_epoch"])
    return kernels.optimize_depth_gpu(
        c["flows"], c["rig"], c["priors"], c["pconfs"], c["confs"], c["depth"], K, c["Rs"], c["ts"], c["dpRs"], c["dpts"],
        kw["abs_resize_factor"], N, N_dp, w, h, kw["basefocal"], kw["n_rand_samples"], kw["global_prop_step"],
        kw["local_prop_width"], kw["lambda_"], kw["omega"], kw["disp_delta"], kw["delta"], kw["fb_smooth"], kw["s0_ems_prob"],
        kw["no_change_prob"], kw["range_factor"], kw["update_rigidness_only"])


# Gates of the FAST kernels against the reference's own kernel code (glibc): the measured value on the MI355X minus one percentage point
# (`exact > 0.5`, `> 0.93`, `0.97 / 0.99` let a regression of k_solve or of the lean arithmetic through); the
# measured values are printed by the tests (pytest -s) and quoted in DESIGN.md section 5.
DEPTH_BRANCH_AGREEMENT = {"rand": 0.99, "global": 0.99, "local": 0.99, "all": 0.989, "all_priors": 0.986, "ragged_all": 0.989}  # measured 1.0, 1.0, 1.0, 0.99967, 0.99674, 0.99907
SOLVE_WITHIN = {"lambdatwist": 0.99, "ap3p": 0.989, "lambdatwist_small": 0.99, "ap3p_small": 0.99}                                  # measured 1.0, 0.99944, 1.0, 1.0
SOLVE_EXACT = {"lambdatwist": 0.99, "lambdatwist_small": 0.99}                                                                      # measured 1.0, 1.0: every translation bit-identical


@pytest.mark.parametrize("name,c", list(cases.depth_cases()), ids=[c[0] for c in cases.depth_cases()])
def test_optimize_depth_vs_reference_kernels(gold, name, c):
    g_depth, g_rig = gold[f"od/{name}/depth"], gold[f"od/{name}/rig"]
    depth, rig, confs = _run_depth(c)
    same = np.abs(depth - g_depth) <= 1e-5 * np.abs(g_depth)
    if name in ("cost", "update_only"):  # no search: the depth map must come back untouched
        np.testing.assert_array_equal(depth, g_depth)
    else:
        need = DEPTH_BRANCH_AGREEMENT[name]
        print(f"MEASURED depth branch agreement {name}: {same.mean():.5f} (gate {need})")
        assert same.mean() >= need, f"{name}: only {same.mean():.4f} of the depth pixels take the reference's branch"
    # E-step on the pixels whose depth agrees (another depth means another residual, legitimately)
    assert np.abs(rig - g_rig)[:, same].max() < 5e-4
    if confs is not None and confs.shape[0]:
        assert np.abs(confs - gold[f"od/{name}/confs"])[:, same].max() < 5e-4


@pytest.mark.parametrize("name,c,active_idx,a", list(cases.collect_cases()), ids=[c[0] for c in cases.collect_cases()])
def test_collect_p3p_vs_reference_kernel(gold, name, c, active_idx, a):
    from voldor_amd import kernels
    N, h, w, _ = c["flows"].shape
    K = np.asarray(c["K"], np.float32).reshape(3, 3)
    p2, p3 = kernels.collect_p3p_instances(c["flows"], c["rig"], c["depth"], K, c["Rs"], c["ts"], N, w, h, active_idx, **a)
    g2, g3 = gold[f"collect/{name}/p2"], gold[f"collect/{name}/p3"]
    fg, fo = np.isfinite(g2[..., 0]), np.isfinite(p2[..., 0])
    assert np.mean(fg != fo) < 1e-3
    both = fg & fo
    if name == "cam2_sumthresh":
        assert not fo.any()  # the reference's inert / all-rejecting sum threshold (collect_p3p_instances.cu:91-93)
        return
    assert both.sum() > 100
    assert np.abs(p2[both] - g2[both]).max() < 2e-3
    assert np.abs(p3[both] - g3[both]).max() < 1e-4 * max(1.0, np.abs(g3[both]).max())


@pytest.mark.parametrize("name,X,uv,K,n_poses,use_ap3p", list(cases.solve_cases()), ids=[c[0] for c in cases.solve_cases()])
def test_solve_batch_p3p_vs_reference_kernel(gold, name, X, uv, K, n_poses, use_ap3p):
    from voldor_amd import kernels
    fn = kernels.solve_batch_p3p_ap3p_gpu if use_ap3p else kernels.solve_batch_p3p_lambdatwist_gpu
    rv, tv = fn(X, uv, K, n_poses)
    g_rv, g_tv = gold[f"solve/{name}/rvecs"], gold[f"solve/{name}/tvecs"]
    fg = np.isfinite(g_rv.sum(1) + g_tv.sum(1))
    fo = np.isfinite(rv.sum(1) + tv.sum(1))
    assert np.mean(fg != fo) < 0.02
    both = fg & fo
    err = np.maximum(np.abs(rv - g_rv).max(1), np.abs(tv - g_tv).max(1))[both]
    within = float(np.mean(err < 2e-3))
    print(f"MEASURED solve {name}: within 2e-3 {within:.5f} (gate {SOLVE_WITHIN[name]})")
    assert within >= SOLVE_WITHIN[name], np.percentile(err, [50, 90, 99])
    if not use_ap3p:  # LambdaTwist is written with the reference's literal types and no fma contraction
        exact = float(np.mean(np.all(tv[both] == g_tv[both], axis=1)))
        print(f"MEASURED solve {name}: bit-identical translations {exact:.5f} (gate {SOLVE_EXACT[name]})")
        assert exact >= SOLVE_EXACT[name], f"only {exact:.3f} of the translations are bit-identical"


@pytest.mark.parametrize("name,space,kernel_var,init_mean,ext,a", list(cases.meanshift_cases()), ids=[c[0] for c in cases.meanshift_cases()])
def test_meanshift_vs_reference(gold, name, space, kernel_var, init_mean, ext, a):
    from voldor_amd import kernels
    mean, conf, iters = kernels.meanshift_gpu(space, kernel_var, init_mean, ext, **a)
    g_mean, g_conf, g_iters = gold[f"ms/{name}/mean"], float(gold[f"ms/{name}/conf"]), int(gold[f"ms/{name}/iters"])
    assert np.abs(mean - g_mean).max() < 2e-4
    assert abs(conf - g_conf) < 1e-4 * max(1.0, g_conf)
    assert abs(iters - g_iters) <= 2


@pytest.mark.parametrize("name,space,mean0,cov0,a", list(cases.rg_cases()), ids=[c[0] for c in cases.rg_cases()])
def test_fit_robust_gaussian_vs_reference(gold, name, space, mean0, cov0, a):
    from voldor_amd import kernels
    rc, mean, covar, dens, iters = kernels.fit_robust_gaussian(space, mean0, cov0, **a)
    g_rc = int(gold[f"rg/{name}/rc"])
    assert (rc == 0) == (g_rc == 0)
    if g_rc != 0:
        np.testing.assert_array_equal(mean, mean0)  # unreliable fit leaves the inputs untouched
        np.testing.assert_array_equal(covar, cov0)
        return
    g_mean, g_cov = gold[f"rg/{name}/mean"], gold[f"rg/{name}/covar"]
    assert abs(dens - float(gold[f"rg/{name}/density"])) < 2e-3
    assert np.abs(mean - g_mean).max() < 5e-4
    assert np.abs(covar - g_cov).max() < 2e-2 * np.abs(g_cov).max()


@pytest.mark.parametrize("name,kf,photo,evals", list(cases.align_cases()), ids=[c[0] for c in cases.align_cases()])
def test_align_frame_vs_reference_kernels(gold, name, kf, photo, evals):
    from voldor_amd import kernels
    rc, shape = kernels.align_frame_init_gpu(kf["images"] if photo else None, kf["depths"], kf["weights"], kf["K"], kf["vbf"],
                                             kf["crw"] if photo else 0.0)
    assert rc == 0
    for en, rf, tf, pr, pt, want_j, apply_w in evals:
        rc, res, jac = kernels.align_frame_eval_gpu(shape, rf, tf, pr, pt, want_j, apply_w)
        assert rc == 0
        g_res = gold[f"align/{name}/{en}/residual"]
        fg, fo = np.isfinite(g_res), np.isfinite(res)
        assert np.mean(fg != fo) < 5e-3, en
        m = fg & fo
        assert np.abs(res[m] - g_res[m]).max() < 5e-4 * max(1.0, np.abs(g_res[m]).max()), en
        if want_j:
            g_jac = gold[f"align/{name}/{en}/jacobian"]
            scale = np.abs(g_jac[m]).max(axis=0) + 1e-12
            big = m & (g_res > 0.02)  # away from the r -> 0 amplification of the sqrt-Cauchy factor (see test_gpu_align.py)
            if big.sum() > 50:
                assert (np.abs(jac[big] - g_jac[big]).max(axis=0) / scale).max() < 2e-3, en
            bad = (np.abs(jac[m] - g_jac[m]) / scale).max(axis=1) > 3e-2
            assert bad.mean() < 2e-2, en


@pytest.mark.parametrize("name,src,sigma,ksize", list(cases.gblur_cases()), ids=[c[0] for c in cases.gblur_cases()])
def test_gblur_vs_reference_kernel(gold, name, src, sigma, ksize):
    from voldor_amd import kernels
    rc, dst = kernels.gblur_gpu(src, sigma, ksize)
    g_rc = int(gold[f"gblur/{name}/rc"])
    assert (rc == 0) == (g_rc == 0)
    if g_rc == 0:
        assert np.abs(dst - gold[f"gblur/{name}/dst"]).max() < 1e-5 * max(1.0, np.abs(src).max())


# ---- strict mode: the HIP kernels equal the REFERENCE's own kernel code, bit for bit ---------------------------------------------
# Goldens: ref_kernels.npz for the stages that call no libm (fb_smooth, collect_p3p_instances, fit_robust_gaussian), and
# ref_kernels_strict.npz (the reference's .cu files with their libm calls served by vk_strict_math.h, gen_golden_kernels.py --strict)
# for optimize_depth, the solvers and mean-shift.  Not bit-exact by design: rotation vectors (D8: exact polar factor instead of the
# reference's approximate SVD).
GOLD_STRICT = os.path.join(os.path.dirname(__file__), "golden", "ref_kernels_strict.npz")


@pytest.fixture()
def strict_mode():
    from voldor_amd import kernels
    kernels.set_strict_math(True)
    yield np.load(GOLD_STRICT)
    kernels.set_strict_math(False)


def _bits(a, b, what):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    eq = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    assert eq.all(), f"{what}: {(~eq).sum()} of {eq.size} values differ"


@pytest.mark.parametrize("name,maps,s0,p", list(cases.fb_cases()), ids=[c[0] for c in cases.fb_cases()])
def test_strict_fb_smooth_equals_reference_kernel(gold, strict_mode, name, maps, s0, p):
    from voldor_amd import kernels
    rc, out = kernels.fb_smooth_gpu(maps, s0, p)
    assert rc == 0
    _bits(out, gold[f"fb/{name}"], "smoothed maps")


@pytest.mark.parametrize("name,c", list(cases.depth_cases()), ids=[c[0] for c in cases.depth_cases()])
def test_strict_optimize_depth_equals_reference_kernels(strict_mode, name, c):
    """cost map, random samples, both propagations (alone and chained), priors, ragged sizes, E-step, prior confidences"""
    g = strict_mode
    depth, rig, confs = _run_depth(c)
    _bits(depth, g[f"od/{name}/depth"], "depth"); _bits(rig, g[f"od/{name}/rig"], "rigidness")
    if confs is not None and confs.shape[0]:
        _bits(confs, g[f"od/{name}/confs"], "prior confidences")


@pytest.mark.parametrize("name,c,active_idx,a", list(cases.collect_cases()), ids=[c[0] for c in cases.collect_cases()])
def test_strict_collect_equals_reference_kernel(gold, strict_mode, name, c, active_idx, a):
    from voldor_amd import kernels
    N, h, w, _ = c["flows"].shape
    K = np.asarray(c["K"], np.float32).reshape(3, 3)
    p2, p3 = kernels.collect_p3p_instances(c["flows"], c["rig"], c["depth"], K, c["Rs"], c["ts"], N, w, h, active_idx, **a)
    _bits(p2, gold[f"collect/{name}/p2"], "p2 map"); _bits(p3, gold[f"collect/{name}/p3"], "p3 map")


@pytest.mark.parametrize("name,X,uv,K,n_poses,use_ap3p", list(cases.solve_cases()), ids=[c[0] for c in cases.solve_cases()])
def test_strict_solver_translations_equal_reference_kernel(strict_mode, name, X, uv, K, n_poses, use_ap3p):
    """index draw + minimal solver + 4th-point selection: every translation bit-identical; rotation vectors to the accuracy of the
    reference's approximate SVD (D8)"""
    from voldor_amd import kernels
    fn = kernels.solve_batch_p3p_ap3p_gpu if use_ap3p else kernels.solve_batch_p3p_lambdatwist_gpu
    rv, tv = fn(X, uv, K, n_poses)
    g_rv, g_tv = strict_mode[f"solve/{name}/rvecs"], strict_mode[f"solve/{name}/tvecs"]
    _bits(tv, g_tv, "translations")
    fin = np.isfinite(g_rv.sum(1))
    assert np.percentile(np.abs(rv[fin] - g_rv[fin]).max(1), 99) < 2e-5


@pytest.mark.parametrize("name,space,kernel_var,init_mean,ext,a", list(cases.meanshift_cases()), ids=[c[0] for c in cases.meanshift_cases()])
def test_strict_meanshift_equals_reference(strict_mode, name, space, kernel_var, init_mean, ext, a):
    from voldor_amd import kernels
    mean, conf, iters = kernels.meanshift_gpu(space, kernel_var, init_mean, ext, **a)
    g = strict_mode
    _bits(mean, g[f"ms/{name}/mean"], "mode")
    assert np.float32(conf).tobytes() == np.float32(g[f"ms/{name}/conf"]).tobytes() and iters == int(g[f"ms/{name}/iters"])


@pytest.mark.parametrize("name,space,mean0,cov0,a", list(cases.rg_cases()), ids=[c[0] for c in cases.rg_cases()])
def test_strict_fit_robust_gaussian_equals_reference(gold, strict_mode, name, space, mean0, cov0, a):
    """mean, covariance, density, iteration count and verdict of fit_robust_gaussian.cu, bit for bit (covariance was 2e-2)"""
    from voldor_amd import kernels
    rc, mean, covar, dens, iters = kernels.fit_robust_gaussian(space, mean0, cov0, **a)
    g_rc = int(gold[f"rg/{name}/rc"])
    assert (rc == 0) == (g_rc == 0)
    if g_rc != 0:
        np.testing.assert_array_equal(mean, mean0); np.testing.assert_array_equal(covar, cov0)
        return
    _bits(mean, gold[f"rg/{name}/mean"], "mean"); _bits(covar, gold[f"rg/{name}/covar"], "covariance")
    assert np.float32(dens).tobytes() == np.float32(gold[f"rg/{name}/density"]).tobytes() and iters == int(gold[f"rg/{name}/iters"])
