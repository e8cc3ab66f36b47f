"""Seeded whole-window inputs shared by tests/golden/gen_golden_window.py (the reference's own py_voldor_wrapper executed on
the CPU -> tests/golden/ref_window.npz) and the tests that compare the oracle / the HIP path with those outputs."""
import numpy as np

from voldor_amd import synth

MONO = "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 8"
STEREO = "--silent --meanshift_kernel_var 0.1 --disp_delta 1 --delta 0.2 --max_iters 8"


# A window whose valid-correspondence density collapses after the first EM iteration: only pixels with rigidness > 0.9995 are
# sampled (config.h:59 rigidness_threshold), i.e. ~1 % of the image for camera 0, a few dozen pixels for camera 1, none for
# camera 2 (the window truncates there).  The reference still forms all n_poses_to_sample hypotheses from those few points
# (index draw over the compacted list, geometry.cpp:68-88 + solve_batch_lambdatwist.cu:16-19); a rejection draw with a probe
# budget would lose camera 1.
LOW_DENSITY_CFG = "--silent --meanshift_kernel_var 0.1 --disp_delta 1 --delta 0.2 --max_iters 3 --rigidness_threshold 0.9995 --trunc_sample_density 0"


def low_density_scene():
    return synth.make_scene(w=256, h=192, n_flows=4, fx=128, fy=128, cx=128, cy=96, seed=239, basefocal=64.0)


def _small():
    return synth.make_scene(w=128, h=96, n_flows=4, fx=64, fy=64, cx=64, cy=48, seed=5, basefocal=30.0)


def _case(sc, config, basefocal=0.0, disparity=False, priors=None, flows=None, b1=False, exact=True, ref_config=None, stat_only=False):
    """b1: the reference run shows SURVEY Appendix B-1 (stale device depth), which the oracle only reproduces with
    ORC_EMULATE_B1=1.  exact: small enough for the bit-exact oracle comparison on the CPU.  ref_config: what the reference
    is run with when it differs from what the oracle / product get (the big cases switch B-1 off in the reference)."""
    c = dict(flows=np.ascontiguousarray(sc["flows"] if flows is None else flows, np.float32), K=tuple(float(v) for v in sc["K"]),
             basefocal=float(basefocal), disparity=np.ascontiguousarray(sc["disparity"], np.float32) if disparity else None,
             depth_priors=None, depth_prior_poses=None, depth_prior_pconfs=None, config=config, b1=b1, exact=exact,
             ref_config=ref_config or config, poses_gt=sc["poses_gt"], stat_only=stat_only)
    if priors is not None:
        c["depth_priors"], c["depth_prior_poses"], c["depth_prior_pconfs"] = priors
    return c


def window_cases():
    sm = _small()
    yield "stereo_default", _case(sm, "--silent", basefocal=30.0, disparity=True)
    yield "stereo_ap3p", _case(sm, "--silent --lambdatwist 0 --max_iters 3", basefocal=30.0, disparity=True)
    yield "mono_nonexclusive", _case(sm, "--silent --exclusive_gpu_context 0 --max_iters 3")
    yield "mono_default_b1", _case(sm, "--silent --max_iters 3", b1=True)
    sp = synth.make_scene(w=160, h=120, n_flows=3, fx=80, fy=80, cx=80, cy=60, seed=237)
    pri = np.stack([sp["depth_gt"], sp["depth_gt"] * 1.01]).astype(np.float32)
    pposes = np.array([[0, 0, 0, 0, 0, 0], [0.001, 0, 0, 0.01, 0, 0]], np.float32)
    yield "depth_priors", _case(sp, "--silent --max_iters 4 --delta 0.5", basefocal=40.0, priors=(pri, pposes, np.full_like(pri, 0.8)))
    st = synth.make_scene(w=160, h=120, n_flows=5, fx=80, fy=80, cx=80, cy=60, seed=238)
    fl = st["flows"].copy()
    fl[3:] = np.random.default_rng(0).uniform(-40, 40, fl[3:].shape).astype(np.float32)
    yield "truncated_b1", _case(st, MONO, flows=fl, b1=True)
    yield "low_density", _case(low_density_scene(), LOW_DENSITY_CFG, basefocal=64.0, disparity=True)
    # BASELINE configs[0]: the reference's CPU geometry path (geometry.cpp:99-143, lambdatwist_p4p<double> on the host with
    # libc rand() draws, which no other implementation can replay): statistical comparison only
    yield "cfg1_cpu_p3p", _case(sp, "--silent --max_iters 8 --cpu_p3p 1 --exclusive_gpu_context 0", stat_only=True)
    # the sizes of tests/test_gpu_voldor.py: reference run with the stale-depth bug switched off (D4), statistical comparison only
    big = synth.make_scene(w=320, h=240, n_flows=5, fx=160, fy=160, cx=160, cy=120, seed=233)
    yield "mono_320x240", _case(big, MONO, exact=False, ref_config=MONO + " --exclusive_gpu_context 0")
    bs = synth.make_scene(w=312, h=96, n_flows=4, fx=180, fy=180, cx=152, cy=46, seed=236, basefocal=97.0)
    yield "stereo_312x96", _case(bs, STEREO, basefocal=97.0, disparity=True, exact=False)


def cfg2_case():
    """BASELINE configs[1] at full size, the very window bench.py times on rank 0 (scene S, seed 233): poses only."""
    sc = synth.make_scene(w=640, h=480, n_flows=5, fx=320.0, fy=320.0, cx=320.0, cy=240.0, seed=233)
    return "cfg2_640x480", _case(sc, MONO, exact=False, ref_config=MONO + " --exclusive_gpu_context 0")


ENSEMBLE_SEEDS = tuple(range(300, 308))


def ensemble_cases():
    """8 independent monocular 320x240 windows (BASELINE cfg2 at quarter size): poses only, for the accuracy-distribution
    comparison of the estimators (reference vs HIP) against analytic ground truth."""
    for seed in ENSEMBLE_SEEDS:
        sc = synth.make_scene(w=320, h=240, n_flows=5, fx=160, fy=160, cx=160, cy=120, seed=seed)
        yield f"ens{seed}", _case(sc, MONO, exact=False, ref_config=MONO + " --exclusive_gpu_context 0")
