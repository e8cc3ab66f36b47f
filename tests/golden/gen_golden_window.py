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
import ctypes as C
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

F = C.POINTER(C.c_float)
D = C.POINTER(C.c_double)


def fp(a):
    return None if a is None else a.ctypes.data_as(F)


def two_view_pose(flow0, K4):
    """(R float32 3x3, t float32 3): the oracle's bootstrap pose BEFORE the reference's cam.t = R * t (geometry.cpp:330)."""
    fx, fy, cx, cy = K4
    K = np.array([fx, 0, cx, 0, fy, cy, 0, 0, 1], np.float32)
    ok, R, _ = orc.estimate_pose_epipolar(flow0, K)
    assert ok
    t = np.zeros(3, np.float32)
    orc.lib().orc_last_two_view_translation(fp(t))
    return np.ascontiguousarray(R, np.float32), t


def run_reference(ref, c, rand_epoch=0):
    flows = c["flows"]
    N, h, w, _ = flows.shape
    fx, fy, cx, cy = c["K"]
    injected = None
    if c["disparity"] is None and c["depth_priors"] is None:
        R, t = two_view_pose(flows[0], c["K"])
        ref.ref_set_two_view_pose(R.astype(np.float64).ctypes.data_as(D), t.astype(np.float64).ctypes.data_as(D))
        injected = np.concatenate([R.reshape(9), t])
    pri = c["depth_priors"]
    N_dp = 0 if pri is None else pri.shape[0]
    poses = np.zeros((N, 6), np.float32)
    cov = np.zeros((N, 6, 6), np.float32)
    depth = np.zeros((h, w), np.float32)
    conf = np.zeros((h, w), np.float32)
    n = C.c_int(0)
    rc = ref.ref_py_voldor_wrapper(fp(flows), fp(c["disparity"]), None, fp(pri), fp(c["depth_prior_poses"]), fp(c["depth_prior_pconfs"]),
                                   C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), C.c_float(c["basefocal"]), N, N_dp, w, h,
                                   c["ref_config"].encode(), C.c_uint(rand_epoch), C.byref(n), fp(poses), fp(cov), fp(depth), fp(conf))
    assert rc == 0
    return dict(n_registered=n.value, poses=poses[:n.value], poses_covar=cov[:n.value], depth=depth, depth_conf=conf, injected=injected)


def main():
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libvoldor_ref.so"))
    out = {}
    for name, c in cases.window_cases():
        r = run_reference(ref, c)
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
    path = os.path.join(HERE, "ref_window.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
