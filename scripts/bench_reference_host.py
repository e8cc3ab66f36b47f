"""Throughput of the drop-in at the B-inner boundary: the REFERENCE's own host pipeline (voldor/*.cpp, unmodified, compiled
against oracle/ref_stubs/minicv) calling the HIP kernels through the gpu_kernels.h symbols of libvoldor_hip.so
(oracle/_ref/libvoldor_refhost_hip.so), next to the library's own B-outer (everything resident in HBM).  BASELINE cfg2.
Run on the GPU box:  python scripts/bench_reference_host.py
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (first: one HIP runtime)

from oracle import orc  # noqa: E402  (only for the injected two-view pose: cv::recoverPose stand-in, deviation D5)
from voldor_amd import capi, kernels, pyvoldor, synth  # noqa: E402

W, H, N = 640, 480, 5
CFG = "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 8"
F = C.POINTER(C.c_float)


def main():
    capi.lib()
    host = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libvoldor_refhost_hip.so"))
    sc = synth.make_scene(w=W, h=H, n_flows=N, fx=320.0, fy=320.0, cx=320.0, cy=240.0, seed=233)
    flows = np.ascontiguousarray(sc["flows"], np.float32)
    R, t = orc.two_view_pose(flows[0], np.array([320, 0, 320, 0, 320, 240, 0, 0, 1], np.float32))
    D = C.POINTER(C.c_double)
    R64, t64 = R.astype(np.float64), t.astype(np.float64)
    host.ref_set_two_view_pose(R64.ctypes.data_as(D), t64.ctypes.data_as(D))
    poses = np.zeros((N, 6), np.float32); covar = np.zeros((N, 6, 6), np.float32)
    depth = np.zeros((H, W), np.float32); conf = np.zeros((H, W), np.float32)
    n = C.c_int(0)

    def ref_host(cfg):
        rc = host.ref_py_voldor_wrapper(flows.ctypes.data_as(F), None, None, None, None, None, C.c_float(320), C.c_float(320), C.c_float(320), C.c_float(240),
                                        C.c_float(0), N, 0, W, H, cfg.encode(), C.c_uint(0), C.byref(n), poses.ctypes.data_as(F), covar.ctypes.data_as(F),
                                        depth.ctypes.data_as(F), conf.ctypes.data_as(F))
        assert rc == 0 and n.value == N

    def native(cfg):
        kernels.set_rand_epoch(0)
        return pyvoldor.voldor(flows, 320.0, 320.0, 320.0, 240.0, config=cfg)

    out = {}
    for name, fn, cfg in (("reference host, exclusive_gpu_context=1 (NULL protocol: device copies reused)", ref_host, CFG),
                          ("reference host, exclusive_gpu_context=0 (everything re-uploaded each call)", ref_host, CFG + " --exclusive_gpu_context 0"),
                          ("library B-outer, host buffers in and out (py_voldor_wrapper)", native, CFG)):
        for _ in range(3):
            fn(cfg)
        t0 = time.perf_counter()
        k = 10
        for _ in range(k):
            fn(cfg)
        dt = (time.perf_counter() - t0) / k
        out[name] = {"ms_per_window": round(dt * 1e3, 2), "windows_per_s": round(1 / dt, 2)}
        print(f"{name:90s} {dt * 1e3:8.2f} ms  {1 / dt:7.1f} windows/s", flush=True)
    gt = sc["poses_gt"].copy(); gt[:, 3:] /= np.mean(np.linalg.norm(gt[:, 3:], axis=1))
    rot, tr = synth.pose_errors(poses[:n.value], gt)
    out["reference_host_pose_rpe_vs_gt"] = {"rot_rad_max": float(rot.max()), "rel_trans_max": float(tr.max())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
