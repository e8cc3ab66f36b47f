"""Generates tests/golden/ref_kernels.npz from the REFERENCE's own kernel files executed on the CPU.

Run in the build container only (needs /root/reference): `python tests/golden/gen_golden_kernels.py`.
oracle/_ref/libvoldor_ref.so is built by `make -C oracle ref`: oracle/ref_prep.pl rewrites the <<< >>> launches of
gpu-kernels/{optimize_depth,collect_p3p_instances,meanshift,fit_robust_gaussian,solve_batch_ap3p,solve_batch_lambdatwist,
align_frame}.cu
and fb_smooth.h into a temp directory, oracle/ref_wrap_kernels.cpp compiles them on top of oracle/ref_stubs/emul/ (sequential
launcher, host-backed GMat; substitutions: D1 counter RNG, D2 exact bilinear, tree reduction order restated, aux_funs LU).
The functions called here are the reference's host entry points themselves (optimize_depth_gpu, collect_p3p_instances,
solve_batch_p3p_*_gpu, meanshift_gpu, fit_robust_gaussian, fb_smooth_batch_inplace, align_frame_init_gpu / align_frame_eval_gpu).

Inputs are re-derived from seeds by tests/ref_kernel_cases.py (shared with the tests), only outputs are stored.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_kernel_cases as cases  # noqa: E402

F = C.POINTER(C.c_float)


def fp(a):
    return None if a is None else a.ctypes.data_as(F)


def load_ref():
    p = os.path.join(ROOT, "oracle", "_ref", "libvoldor_ref.so")
    ref = C.CDLL(p)
    assert hasattr(ref, "ref_optimize_depth"), "oracle/_ref built without the kernel emulation (make -C oracle ref)"
    return ref


def ref_fb(ref, maps, s0, p):
    m = np.ascontiguousarray(maps, np.float32).copy()
    n, h, w = m.shape
    ref.ref_fb_smooth(fp(m), n, w, h, C.c_float(s0), C.c_float(p))
    return m


def ref_depth(ref, c):
    kw = c["kw"]
    flows, rig, depth = c["flows"], c["rig"].copy(), c["depth"].copy()
    N, h, w, _ = flows.shape
    pri, pc = c["priors"], c["pconfs"]
    cf = None if c["confs"] is None else c["confs"].copy()
    N_dp = 0 if pri is None else pri.shape[0]
    cost = np.zeros_like(depth)
    K = np.ascontiguousarray(c["K"], np.float32)
    rc = ref.ref_optimize_depth(
        fp(flows), fp(rig), fp(pri), fp(pc), fp(cf), fp(depth), fp(cost), fp(K), fp(c["Rs"]), fp(c["ts"]), fp(c["dpRs"]), fp(c["dpts"]),
        C.c_float(kw["abs_resize_factor"]), N, N_dp, w, h, C.c_float(kw["basefocal"]), kw["n_rand_samples"], kw["global_prop_step"],
        kw["local_prop_width"], C.c_float(kw["lambda_"]), C.c_float(kw["omega"]), C.c_float(kw["disp_delta"]), C.c_float(kw["delta"]),
        kw["fb_smooth"], C.c_float(kw["s0_ems_prob"]), C.c_float(kw["no_change_prob"]), C.c_float(kw["range_factor"]),
        kw["update_rigidness_only"], C.c_uint(c["rand_epoch"]))
    assert rc == 0
    return depth, rig, cf, cost


def ref_collect(ref, c, active_idx, a):
    flows, rig, depth = c["flows"], c["rig"], c["depth"]
    N, h, w, _ = flows.shape
    p2 = np.zeros((h, w, 2), np.float32)
    p3 = np.zeros((h, w, 3), np.float32)
    K = np.ascontiguousarray(c["K"], np.float32)
    rc = ref.ref_collect_p3p(fp(flows), fp(rig), fp(depth), fp(K), fp(c["Rs"]), fp(c["ts"]), fp(p2), fp(p3), N, w, h, active_idx,
                             C.c_float(a["rigidness_thresh"]), C.c_float(a["rigidness_sum_thresh"]), C.c_float(a["sample_min_depth"]),
                             C.c_float(a["sample_max_depth"]), a["max_trace_on_flow"])
    assert rc == 0
    return p2, p3


def ref_solve(ref, p3s, p2s, K, n_poses, use_ap3p):
    rv = np.zeros((n_poses, 3), np.float32)
    tv = np.zeros((n_poses, 3), np.float32)
    K = np.ascontiguousarray(K, np.float32)
    rc = ref.ref_solve_batch_p3p(fp(p3s), fp(p2s), fp(rv), fp(tv), fp(K), p3s.shape[0], n_poses, int(use_ap3p))
    assert rc == 0
    return rv, tv


def ref_meanshift(ref, space, kernel_var, init_mean, ext, a):
    mean = np.ascontiguousarray(init_mean, np.float32).copy()
    conf, it = C.c_float(0), C.c_int(0)
    N, dims = space.shape
    rc = ref.ref_meanshift(fp(space), C.c_float(kernel_var), fp(mean), C.byref(conf), C.byref(it), int(ext), N, dims,
                           C.c_float(a["epsilon"]), a["max_iters"], a["max_init_trials"], C.c_float(a["good_init_confidence"]))
    assert rc == 0
    return mean, np.float32(conf.value), it.value


def ref_rg(ref, space, mean, covar, a):
    mean = np.ascontiguousarray(mean, np.float32).copy()
    covar = np.ascontiguousarray(covar, np.float32).copy()
    dens, it = C.c_float(0), C.c_int(0)
    N, dims = space.shape
    rc = ref.ref_fit_robust_gaussian(fp(space), fp(mean), fp(covar), C.c_float(a["trunc_sigma"]), C.c_float(a["covar_reg_lambda"]),
                                     C.byref(dens), C.byref(it), N, dims, C.c_float(a["epsilon"]), a["max_iters"])
    return rc, mean, covar, np.float32(dens.value), it.value


def ref_align(ref, kf, photo, evals):
    depths = np.ascontiguousarray(kf["depths"], np.float32)
    N, h, w = depths.shape
    images = np.ascontiguousarray(kf["images"], np.float32) if photo else None
    weights = np.ascontiguousarray(kf["weights"], np.float32)
    K = np.ascontiguousarray(kf["K"], np.float32).reshape(9)
    assert ref.ref_align_init(fp(images), fp(depths), fp(weights), fp(K), C.c_float(kf["vbf"]), C.c_float(kf["crw"] if photo else 0.0), N, w, h) == 0
    for name, rf, tf, pr, pt, want_j, apply_w in evals:
        res = np.zeros((h, w), np.float32)
        jac = np.zeros((h, w, 9), np.float32) if want_j else None
        pr, pt = np.ascontiguousarray(pr, np.float32), np.ascontiguousarray(pt, np.float32)
        assert ref.ref_align_eval(rf, tf, fp(pr), fp(pt), fp(res), fp(jac), int(apply_w)) == 0
        yield name, res, jac


def main_strict(ref):
    """tests/golden/ref_kernels_strict.npz: the stages that call the C library's transcendentals (optimize_depth: expf / powf / logf;
    the solvers: atan2f, AP3P's cbrtf / powf / cosf; mean-shift: expf) run again with those calls served by
    voldor_amd/csrc/vk_strict_math.h (ref_set_math_mode(1), oracle/ref_stubs/emul/cuda_emul.h).  The HIP kernels in strict mode must
    reproduce these bit for bit (tests/test_gpu_vs_ref_kernels.py); fb_smooth, collect and the robust Gaussian call no libm, their
    strict-mode outputs are compared with ref_kernels.npz itself."""
    out = {}
    ref.ref_set_math_mode(1)
    try:
        for name, c in cases.depth_cases():
            d, r, cf, cost = ref_depth(ref, c)
            out[f"od/{name}/depth"], out[f"od/{name}/rig"], out[f"od/{name}/cost"] = d, r, cost
            if cf is not None:
                out[f"od/{name}/confs"] = cf
        for name, p3s, p2s, K, n_poses, use_ap3p in cases.solve_cases():
            rv, tv = ref_solve(ref, p3s, p2s, K, n_poses, use_ap3p)
            out[f"solve/{name}/rvecs"], out[f"solve/{name}/tvecs"] = rv, tv
        for name, space, kernel_var, init_mean, ext, a in cases.meanshift_cases():
            mean, conf, it = ref_meanshift(ref, space, kernel_var, init_mean, ext, a)
            out[f"ms/{name}/mean"], out[f"ms/{name}/conf"], out[f"ms/{name}/iters"] = mean, conf, np.int32(it)
    finally:
        ref.ref_set_math_mode(0)
    path = os.path.join(HERE, "ref_kernels_strict.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    ref = load_ref()
    if "--strict" in sys.argv:
        return main_strict(ref)
    out = {}
    for name, kf, photo, evals in cases.align_cases():
        for ename, res, jac in ref_align(ref, kf, photo, evals):
            out[f"align/{name}/{ename}/residual"] = res
            if jac is not None:
                out[f"align/{name}/{ename}/jacobian"] = jac
    for name, maps, s0, p in cases.fb_cases():
        out[f"fb/{name}"] = ref_fb(ref, maps, s0, p)
    for name, c in cases.depth_cases():
        d, r, cf, cost = ref_depth(ref, c)
        out[f"od/{name}/depth"], out[f"od/{name}/rig"], out[f"od/{name}/cost"] = d, r, cost
        if cf is not None:
            out[f"od/{name}/confs"] = cf
    for name, c, active_idx, a in cases.collect_cases():
        p2, p3 = ref_collect(ref, c, active_idx, a)
        out[f"collect/{name}/p2"], out[f"collect/{name}/p3"] = p2, p3
    for name, p3s, p2s, K, n_poses, use_ap3p in cases.solve_cases():
        rv, tv = ref_solve(ref, p3s, p2s, K, n_poses, use_ap3p)
        out[f"solve/{name}/rvecs"], out[f"solve/{name}/tvecs"] = rv, tv
    for name, space, kernel_var, init_mean, ext, a in cases.meanshift_cases():
        mean, conf, it = ref_meanshift(ref, space, kernel_var, init_mean, ext, a)
        out[f"ms/{name}/mean"], out[f"ms/{name}/conf"], out[f"ms/{name}/iters"] = mean, conf, np.int32(it)
    for name, space, mean, covar, a in cases.rg_cases():
        rc, m, cv, dens, it = ref_rg(ref, space, mean, covar, a)
        out[f"rg/{name}/rc"], out[f"rg/{name}/mean"], out[f"rg/{name}/covar"] = np.int32(rc), m, cv
        out[f"rg/{name}/density"], out[f"rg/{name}/iters"] = dens, np.int32(it)
    for name, src, sigma, ksize in cases.gblur_cases():
        dst = np.zeros_like(src)
        d, h, w = src.shape
        rc = ref.ref_gblur(fp(src), fp(dst), w, h, d, C.c_float(sigma), ksize)
        out[f"gblur/{name}/rc"] = np.int32(rc)
        if rc == 0:
            out[f"gblur/{name}/dst"] = dst
    path = os.path.join(HERE, "ref_kernels.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
