"""Generates tests/golden/ref_*.npz from the REFERENCE's own code.

Run in the authoring container only (needs /root/reference and oracle/_ref/libvoldor_ref.so,
built by `make -C oracle ref` from the reference sources where they lie):

    python tests/golden/gen_golden.py

The vectors pin the oracle (and the product's host-compiled solver math) to what
/root/reference/lambdatwist/*.h, gpu-kernels/residual_model.h and gpu-kernels/rodrigues.h compute.
/root/reference does not exist on the GPU box; only these .npz files travel.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
import ctypes as C  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def rodrigues_np(rv):
    th = np.linalg.norm(rv)
    if th < 1e-15:
        return np.eye(3)
    k = rv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx


def main():
    ref = orc.ref()
    assert ref is not None, "oracle/_ref/libvoldor_ref.so missing: run `make -C oracle ref` where /root/reference exists"
    rng = np.random.default_rng(20240925)
    # ---- residual model (residual_model.h:15-68)
    n = 4096
    flows = rng.normal(0, 8, (n, 4)).astype(np.float32)
    flows[: n // 8, :2] = flows[: n // 8, 2:]  # zero error rows
    flows[n // 8: n // 4, 2:] = 0  # zero observed flow
    flows[n // 4: n // 4 + 64] *= 60  # beyond the fmag clamp
    lam = rng.choice([0.05, 0.15, 0.3], n).astype(np.float32)
    arf = rng.choice([0.5, 1.0, 2.0], n).astype(np.float32)
    rig = np.array([ref.ref_fun_rigidness(*map(float, flows[i]), float(lam[i]), float(arf[i])) for i in range(n)], np.float32)
    d12 = rng.uniform(0.5, 60, (n, 2)).astype(np.float32)
    bf = rng.choice([97.0, 386.1, 480.0], n).astype(np.float32)
    drig = np.array([ref.ref_fun_depth_rigidness(float(d12[i, 0]), float(d12[i, 1]), float(bf[i]), 0.15, float(arf[i])) for i in range(n)], np.float32)
    fm = rng.uniform(0, 300, n).astype(np.float32)
    fc = np.array([ref.ref_fun_fmag_c(float(v)) for v in fm], np.float32)
    fs = np.array([ref.ref_fun_fmag_scale(float(v)) for v in fm], np.float32)
    cost = np.zeros((n, 2), np.float32)
    for i in range(n):
        a, b = C.c_float(0.25), C.c_float(0.5)
        ref.ref_fun_cost(C.c_float(flows[i, 0]), C.c_float(flows[i, 1]), C.c_float(flows[i, 2]), C.c_float(flows[i, 3]),
                         C.c_float(0.7), C.c_float(lam[i]), C.c_float(arf[i]), C.byref(a), C.byref(b))
        cost[i] = (a.value, b.value)
    np.savez_compressed(os.path.join(HERE, "ref_residual.npz"), flows=flows, lam=lam, arf=arf, rigidness=rig, d12=d12, bf=bf,
                        depth_rigidness=drig, fmag=fm, fmag_c=fc, fmag_scale=fs, cost_after=cost)
    # ---- LambdaTwist P4P (lambdatwist_p4p.h:5-62) in both instantiations
    m = 2048
    fx, fy, cx, cy = 320.0, 330.0, 310.0, 245.0
    Y = np.zeros((m, 8), np.float32); X = np.zeros((m, 12), np.float32)
    for i in range(m):
        P = rng.uniform([-4, -3, 1.5], [4, 3, 25], (4, 3))
        if i % 16 == 0:
            P[2] = P[0] + 1e-3 * rng.normal(size=3)  # near-degenerate triple
        rv = rng.normal(0, 0.05, 3); t = rng.normal(0, 0.4, 3)
        Pc = P @ rodrigues_np(rv).T + t
        y = np.stack([fx * Pc[:, 0] / Pc[:, 2] + cx, fy * Pc[:, 1] / Pc[:, 2] + cy], -1)
        y += rng.normal(0, 0.4 if i % 3 else 0.0, y.shape)
        Y[i] = y.reshape(-1); X[i] = P.reshape(-1)
    out = {}
    for name, fn in (("f", ref.ref_lambdatwist_p4p_f), ("d", ref.ref_lambdatwist_p4p_d)):
        ok = np.zeros(m, np.int32); R = np.zeros((m, 9), np.float32); T = np.zeros((m, 3), np.float32)
        for i in range(m):
            ok[i] = fn(orc._fp(Y[i]), orc._fp(X[i]), fx, fy, cx, cy, orc._fp(R[i]), orc._fp(T[i]))
        out["ok_" + name] = ok; out["R_" + name] = R; out["t_" + name] = T
    np.savez_compressed(os.path.join(HERE, "ref_lambdatwist.npz"), y=Y, x=X, K=np.array([fx, fy, cx, cy], np.float32), **out)
    # ---- rodrigues (rodrigues.h:82-114: SVD-orthonormalise, then angle-axis)
    k = 1024
    Rin = np.zeros((k, 9), np.float32); rv_out = np.zeros((k, 3), np.float32); aa_out = np.zeros((k, 3), np.float32)
    ref.ref_rodrigues.argtypes = None
    for i in range(k):
        rv = rng.normal(0, [0.02, 0.3, 1.2][i % 3], 3)
        R = rodrigues_np(rv) + rng.normal(0, [0, 1e-4, 1e-2][(i // 3) % 3], (3, 3))
        Rin[i] = R.reshape(-1).astype(np.float32)
        ref.ref_rodrigues(orc._fp(Rin[i]), orc._fp(rv_out[i]), None)
        ref.ref_rotmat_to_angle_axis(orc._fp(Rin[i]), orc._fp(aa_out[i]))
    # the projected matrix itself (U V^T of svd3_cuda.h, no libm in it) and the strict-math rotation vectors: what
    # voldor_amd/csrc/vk_ref_svd.h (--reference_svd 1) must reproduce to the bit.  Plus matrices far from SO(3): scaled, reflected,
    # rank-deficient, tiny, with repeated columns -- every branch of the sort / Givens stage.
    rw = np.random.default_rng(77)
    wild = [rw.normal(0, s, (512, 9)) for s in (1.0, 1e-3, 50.0)]
    wild.append(Rin[:256] * rw.uniform(0.3, 3.0, (256, 1)))
    refl = Rin[:256].reshape(-1, 3, 3).copy(); refl[:, :, 1] *= -1; wild.append(refl.reshape(-1, 9))
    low = rw.normal(0, 1, (128, 3, 3)); low[:, :, 2] = low[:, :, 0]; wild.append(low.reshape(-1, 9))
    wild.append(np.zeros((1, 9))); wild.append(np.eye(3).reshape(1, 9)); wild.append(rw.normal(0, 1e-12, (32, 9)))
    Rw = np.concatenate(wild).astype(np.float32)
    Rall = np.concatenate([Rin, Rw])
    proj = np.zeros_like(Rall); rv_all = np.zeros((len(Rall), 3), np.float32); rv_strict = np.zeros((len(Rall), 3), np.float32)
    for i in range(len(Rall)):
        ref.ref_rodrigues(orc._fp(Rall[i]), orc._fp(rv_all[i]), orc._fp(proj[i]))
    ref.ref_set_math_mode(1)
    try:
        for i in range(len(Rall)):
            ref.ref_rodrigues(orc._fp(Rall[i]), orc._fp(rv_strict[i]), None)
    finally:
        ref.ref_set_math_mode(0)
    assert np.array_equal(rv_all[:k].view(np.uint32), rv_out.view(np.uint32))
    np.savez_compressed(os.path.join(HERE, "ref_rodrigues.npz"), R=Rin, rvec=rv_out, angle_axis_no_svd=aa_out,
                        R_all=Rall, proj_all=proj, rvec_all=rv_all, rvec_all_strict=rv_strict)
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])


if __name__ == "__main__":
    main()
