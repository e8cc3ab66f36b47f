"""SURVEY.md 8(f)-2, CPU side: the oracle of the frame-alignment maps.  rot_with_rvec is pinned to vectors produced by
the reference's own function (align_frame.cu:47-137 compiled in place, tests/golden/ref_rot.npz); the residual maps are
checked through properties."""
import os

import numpy as np

from align_scene import keyframes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_rot_with_rvec_matches_reference_vectors(orc):
    z = np.load(os.path.join(GOLD, "ref_rot.npz"))
    for i in range(len(z["p"])):
        out, jw, jp = orc.rot_with_rvec(z["p"][i], z["rvec"][i])
        assert np.abs(out - z["out"][i]).max() <= 2e-6 * max(1.0, np.abs(z["out"][i]).max())
        assert np.abs(jw.reshape(-1) - z["J_rvec"][i]).max() <= 2e-6 * max(1.0, np.abs(z["J_rvec"][i]).max())  # incl. the theta^(3/2) quirk
        assert np.abs(jp.reshape(-1) - z["J_p3"][i]).max() <= 2e-6


def test_align_residual_vanishes_at_the_true_poses_and_grows_away_from_them(orc):
    kf = keyframes()
    A = orc.Align(kf["images"], kf["depths"], kf["weights"], kf["K"], kf["vbf"], kf["crw"])
    r0, _ = A.eval(0, 1, kf["params"][0], kf["params"][1], want_jacobian=False)
    ok = np.isfinite(r0)
    assert ok.mean() > 0.7
    assert np.median(r0[ok]) < 0.05  # sqrt-Cauchy of a ~0 point-to-plane / colour error (interpolation noise only)
    off = kf["params"][0].copy()
    off[3:6] += [0.05, 0.0, 0.05]
    r1, _ = A.eval(0, 1, off, kf["params"][1], want_jacobian=False)
    both = ok & np.isfinite(r1)
    assert np.sum(r1[both] ** 2) > 3 * np.sum(r0[both] ** 2)
    # a frame against itself with identical parameters: exactly zero wherever it is valid
    rs, js = A.eval(1, 1, kf["params"][1], kf["params"][1])
    m = np.isfinite(rs)
    assert m.mean() > 0.9 and np.abs(rs[m]).max() < 2e-3 and js.shape == (120, 160, 9)


def test_align_jacobian_gives_a_descent_step(orc):
    """One damped Gauss-Newton step on the 6 pose parameters of the reference frame must lower the cost (what Ceres does
    with these maps, frame-alignment/align_frame_cost_fun.h:157-238)."""
    kf = keyframes(seed=3)
    A = orc.Align(kf["images"], kf["depths"], kf["weights"], kf["K"], kf["vbf"], kf["crw"])
    p = kf["params"][0].copy()
    p[:3] += [0.004, -0.003, 0.002]
    p[3:6] += [0.03, -0.02, 0.04]

    def cost(pp):
        r, j = A.eval(0, 1, pp, kf["params"][1])
        m = np.isfinite(r)
        return r, j, m

    r, j, m = cost(p)
    Jm, rm = j[m][:, :6].astype(np.float64), r[m].astype(np.float64)
    delta = -np.linalg.solve(Jm.T @ Jm + 1e-3 * np.eye(6) * np.trace(Jm.T @ Jm) / 6, Jm.T @ rm)
    p2 = p.copy()
    p2[:6] += delta.astype(np.float32)
    r2, _, m2 = cost(p2)
    both = m & m2
    assert np.sum(r2[both] ** 2) < 0.7 * np.sum(r[both] ** 2)
    err0 = np.linalg.norm(p[3:6] - kf["params"][0][3:6]); err1 = np.linalg.norm(p2[3:6] - kf["params"][0][3:6])
    assert err1 < err0  # and moves the translation towards the truth
