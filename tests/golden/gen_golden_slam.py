"""Generates tests/golden/ref_covis.npz and ref_flo.bin from the REFERENCE's own Python code
(slam_py/slam_utils.py:18-53 eval_covisibility, slam_py/flow_utils.py:23-30 save_flow).

Run in the authoring container only (needs /root/reference):   python tests/golden/gen_golden_slam.py

The reference modules import cv2 / pylab at module level but the two functions used here need numpy only, so empty stub
modules are registered first.  /root/reference does not exist on the GPU box; only the generated files travel.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
for name in ("cv2", "pylab"):
    m = types.ModuleType(name)
    if name == "pylab":
        m.box = None
    sys.modules[name] = m
sys.path.insert(0, "/root/reference/slam_py")
import slam_utils  # noqa: E402  (the reference)
import flow_utils  # noqa: E402

from voldor_amd import synth  # noqa: E402


def main():
    cases = []
    rng = np.random.default_rng(77)
    for (w, h, stride, seed) in ((320, 240, 4, 1), (322, 246, 4, 2), (640, 480, 4, 3), (200, 120, 2, 4)):
        sc = synth.make_scene(w=w, h=h, n_flows=3, fx=w / 2, fy=w / 2, cx=w / 2, cy=h / 2, seed=500 + seed)
        K = np.array([[w / 2, 0, w / 2], [0, w / 2, h / 2], [0, 0, 1]], np.float32)
        depth = sc["depth_gt"].astype(np.float32)
        conf = rng.uniform(0, 1, depth.shape).astype(np.float32)
        for step in range(1, 4):
            T = np.eye(4, dtype=np.float32)
            for i in range(step):
                Ti = np.eye(4, dtype=np.float32)
                Ti[:3, :3] = synth.rodrigues(sc["poses_gt"][i, :3] * 8).astype(np.float32)  # exaggerated motion: some points leave
                Ti[:3, 3] = sc["poses_gt"][i, 3:] * 6
                T = Ti @ T
            for use_mask in (False, True):
                mask = (conf > 0.3) if use_mask else None
                score = slam_utils.eval_covisibility(depth, T, K, mask, stride)
                cases.append(dict(w=w, h=h, stride=stride, depth=depth, T=T, K=K, mask=None if mask is None else mask, score=np.float64(score)))
    np.savez_compressed(os.path.join(HERE, "ref_covis.npz"), n=len(cases),
                        **{f"{k}_{i}": (np.zeros(0) if v is None else v) for i, c in enumerate(cases) for k, v in c.items()})
    flow = rng.normal(0, 5, (6, 9, 2)).astype(np.float32)
    path = os.path.join(HERE, "ref_flo.bin")
    flow_utils.save_flow(path, flow)
    np.save(os.path.join(HERE, "ref_flo_values.npy"), flow)
    assert np.array_equal(flow_utils.load_flow(path), flow)
    print("wrote", len(cases), "covisibility cases and ref_flo.bin (", os.path.getsize(path), "bytes )")




def gen_rot():
    """rot_with_rvec of gpu-kernels/align_frame.cu:47-137 through oracle/_ref (built in place by `make -C oracle ref`)."""
    import ctypes as C
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libvoldor_ref.so"))
    F = C.POINTER(C.c_float)
    rng = np.random.default_rng(4242)
    n = 2000
    p = rng.normal(0, 3, (n, 3)).astype(np.float32)
    w = (rng.normal(0, 1, (n, 3)) * rng.choice([1e-5, 1e-3, 0.05, 0.5, 2.0], (n, 1))).astype(np.float32)
    w[:20] = 0  # first-order branch
    out = np.zeros((n, 3), np.float32); jw = np.zeros((n, 9), np.float32); jp = np.zeros((n, 9), np.float32)
    for i in range(n):
        ref.ref_rot_with_rvec(p[i].ctypes.data_as(F), w[i].ctypes.data_as(F), out[i].ctypes.data_as(F), jw[i].ctypes.data_as(F), jp[i].ctypes.data_as(F))
    np.savez_compressed(os.path.join(HERE, "ref_rot.npz"), p=p, rvec=w, out=out, J_rvec=jw, J_p3=jp)
    print("wrote ref_rot.npz", n)


if __name__ == "__main__":
    main()
    gen_rot()
