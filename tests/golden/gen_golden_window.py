"""Generates tests/golden/ref_window.npz: outputs of the REFERENCE's own py_voldor_wrapper (voldor/py_export.cpp) executed
on the CPU.  Run in the build container only: `python tests/golden/gen_golden_window.py` after `make -C oracle ref`.

What runs (oracle/ref_wrap_host.cpp): voldor/{py_export,voldor,geometry,utils}.cpp compiled in place against
oracle/ref_stubs/minicv (stand-in for the OpenCV calls they make) and linked to the reference's kernel files compiled for
the CPU (oracle/ref_wrap_kernels.cpp; D1 counter RNG, D2 exact bilinear).  Monocular windows need cv::findEssentialMat /
recoverPose, which minicv does not have: the two-view pose is injected from the oracle's 8-point LMedS bootstrap
(deviation D5), everything after it is the reference's code.

Stored per case: n_registered, poses, poses_covar, depth, depth_conf (full maps for the small cases, sha256 + a 2x2
subsample for the large ones) and the injected two-view pose.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_window_cases as cases  # noqa: E402
from oracle import orc  # noqa: E402

def five_point_two_view(flow0, K9):
    """(R, unit t before cam.t = R t) of the PRODUCT's five-point LMedS bootstrap, from the host build of its own source (voldor_amd/csrc/vk_bootstrap.hip
    lmeds_essential_host5: the kernels give these bits, tests/test_fivept.py) -- what a window of the fast mode starts from since round 6.  The product returns
    t = R t_b (geometry.cpp:330): t_b = R^T t."""
    from voldor_amd import kernels
    ok, R, t = kernels.estimate_pose_epipolar5(flow0, K9)
    if not ok:
        raise RuntimeError("five-point bootstrap failed")
    tb = (R.astype(np.float64).T @ t.astype(np.float64))
    return np.ascontiguousarray(R, np.float32), (tb / np.linalg.norm(tb)).astype(np.float32)


def run_reference(c, rand_epoch=0, five_point=False):
    fx, fy, cx, cy = c["K"]
    injected, two_view = None, None
    if c["disparity"] is None and c["depth_priors"] is None:  # monocular: orc.ref_voldor injects this two-view pose (D5)
        K9 = np.array([fx, 0, cx, 0, fy, cy, 0, 0, 1], np.float32)
        R, t = five_point_two_view(c["flows"][0], K9) if five_point else orc.two_view_pose(c["flows"][0], K9)
        injected = np.concatenate([R.reshape(9), t])
        two_view = (R, t) if five_point else None
    r = orc.ref_voldor(c["flows"], fx, fy, cx, cy, basefocal=c["basefocal"], disparity=c["disparity"], depth_priors=c["depth_priors"],
                       depth_prior_poses=c["depth_prior_poses"], depth_prior_pconfs=c["depth_prior_pconfs"], config=c["ref_config"],
                       rand_epoch=rand_epoch, two_view=two_view)
    r["injected"] = injected
    return r


def main():
    out = {}
    for name, c in cases.window_cases():
        r = run_reference(c)
        out[f"{name}/n_registered"] = np.int32(r["n_registered"])
        out[f"{name}/poses"], out[f"{name}/poses_covar"] = r["poses"], r["poses_covar"]
        if r["injected"] is not None:
            out[f"{name}/two_view_pose"] = r["injected"]
        for k in ("depth", "depth_conf"):
            if c["exact"]:
                out[f"{name}/{k}"] = r[k]
            else:
                out[f"{name}/{k}_sub2"] = r[k][::2, ::2].copy()
                out[f"{name}/{k}_sha256"] = np.frombuffer(hashlib.sha256(r[k].tobytes()).digest(), np.uint8)
        print(f"{name:20s} n_registered {r['n_registered']}  |t| {np.linalg.norm(r['poses'][:, 3:], axis=1).round(4) if r['n_registered'] else ''}")
    for name, c in cases.ensemble_cases():  # poses only
        r = run_reference(c)
        out[f"{name}/n_registered"], out[f"{name}/poses"] = np.int32(r["n_registered"]), r["poses"]
        print(f"{name:20s} n_registered {r['n_registered']}")
    name, c = cases.cfg2_case()  # ~35 s on one core
    r = run_reference(c)
    out[f"{name}/n_registered"], out[f"{name}/poses"], out[f"{name}/poses_covar"] = np.int32(r["n_registered"]), r["poses"], r["poses_covar"]
    print(f"{name:20s} n_registered {r['n_registered']}")
    path = os.path.join(HERE, "ref_window.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
