"""SURVEY.md 8(f)-3/4, CPU side: disk formats at the boundary and the oracle of the post-VO covisibility step, against
vectors produced by the reference's own Python functions (tests/golden/gen_golden_slam.py)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _covis_cases():
    z = np.load(os.path.join(GOLD, "ref_covis.npz"))
    for i in range(int(z["n"])):
        mask = z[f"mask_{i}"]
        yield dict(w=int(z[f"w_{i}"]), h=int(z[f"h_{i}"]), stride=int(z[f"stride_{i}"]), depth=z[f"depth_{i}"], T=z[f"T_{i}"],
                   K=z[f"K_{i}"], mask=None if mask.size == 0 else mask.astype(bool), score=float(z[f"score_{i}"]))


def test_oracle_covisibility_reproduces_reference():
    from oracle import orc_slam
    n = 0
    for c in _covis_cases():
        s = orc_slam.eval_covisibility(c["depth"], c["T"], c["K"], c["mask"], c["stride"])
        assert s == pytest.approx(c["score"], abs=1e-12), (n, s, c["score"])
        n += 1
    assert n == 24


def test_flo_reader_matches_reference_bytes(tmp_path):
    from voldor_amd import formats
    want = np.load(os.path.join(GOLD, "ref_flo_values.npy"))
    got = formats.load_flow(os.path.join(GOLD, "ref_flo.bin"))
    np.testing.assert_array_equal(got, want)
    p = str(tmp_path / "mine.flo")
    formats.save_flow(p, want)
    assert open(p, "rb").read() == open(os.path.join(GOLD, "ref_flo.bin"), "rb").read()  # byte-identical to flow_utils.save_flow
    # header layout: float32 202021.25, int32 w, int32 h, little endian
    assert struct.unpack("<fii", open(p, "rb").read(12)) == (202021.25, 9, 6)
    bad = str(tmp_path / "bad.flo")
    open(bad, "wb").write(struct.pack("<fii", 1.0, 9, 6) + b"\0" * 432)
    assert formats.load_flow(bad) is None  # flow_utils.py:14-21
    trunc = str(tmp_path / "trunc.flo")
    open(trunc, "wb").write(open(p, "rb").read()[:100])
    with pytest.raises(ValueError):
        formats.load_flow(trunc)


def test_flo_c_entry_points(tmp_path):
    from voldor_amd import capi, formats
    lib = capi.lib()
    want = np.load(os.path.join(GOLD, "ref_flo_values.npy"))
    w, h = C.c_int(0), C.c_int(0)
    path = os.path.join(GOLD, "ref_flo.bin").encode()
    assert lib.vk_read_flo(path, C.byref(w), C.byref(h), None, C.c_size_t(0)) == 0 and (w.value, h.value) == (9, 6)
    buf = np.zeros((6, 9, 2), np.float32)
    assert lib.vk_read_flo(path, C.byref(w), C.byref(h), buf.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(buf.size)) == 0
    np.testing.assert_array_equal(buf, want)
    assert lib.vk_read_flo(path, C.byref(w), C.byref(h), buf.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(10)) == 3
    assert lib.vk_read_flo(b"/nonexistent/x.flo", C.byref(w), C.byref(h), None, C.c_size_t(0)) == 1
    out = str(tmp_path / "c.flo")
    assert lib.vk_write_flo(out.encode(), want.ctypes.data_as(C.POINTER(C.c_float)), 9, 6) == 0
    assert open(out, "rb").read() == open(os.path.join(GOLD, "ref_flo.bin"), "rb").read()
    bad = str(tmp_path / "bad.flo")
    open(bad, "wb").write(struct.pack("<fii", 7.0, 9, 6))
    assert lib.vk_read_flo(bad.encode(), C.byref(w), C.byref(h), None, C.c_size_t(0)) == 2


def test_png_and_disparity_round_trip(tmp_path):
    from voldor_amd import formats
    rng = np.random.default_rng(5)
    img16 = rng.integers(0, 65536, (13, 17)).astype(np.uint16)
    img8 = rng.integers(0, 256, (5, 31)).astype(np.uint8)
    for ft in range(5):  # every PNG scanline filter
        p = str(tmp_path / f"f{ft}.png")
        formats.write_png_gray(p, img16, ft)
        np.testing.assert_array_equal(formats.read_png_gray(p), img16)
        formats.write_png_gray(p, img8, ft)
        np.testing.assert_array_equal(formats.read_png_gray(p), img8)
    disp = rng.uniform(0, 200, (10, 12)).astype(np.float32)
    p = str(tmp_path / "d.png")
    formats.save_disparity_png(p, disp)
    got = formats.load_disparity(p)  # voldor_slam.py:305-307: uint16 / 256
    assert got.dtype == np.float32 and np.abs(got - disp).max() <= 0.5 / 256 + 1e-6
    fl = np.stack([-disp, np.zeros_like(disp)], -1)
    formats.save_flow(str(tmp_path / "d.flo"), fl)
    np.testing.assert_array_equal(formats.load_disparity(str(tmp_path / "d.flo")), disp)  # :302-304
    with pytest.raises(ValueError):
        formats.load_disparity(str(tmp_path / "d.tiff"))


def test_pose_text_formats(tmp_path):
    from voldor_amd import formats, synth
    rng = np.random.default_rng(9)
    Ts = []
    for _ in range(5):
        T = np.eye(4)
        T[:3, :3] = synth.rodrigues(rng.normal(0, 1.0, 3))
        T[:3, 3] = rng.normal(0, 2, 3)
        Ts.append(T)
    p = str(tmp_path / "kitti.txt")
    formats.save_poses(p, Ts, "KITTI")
    np.testing.assert_allclose(formats.load_poses_kitti(p), np.stack(Ts), rtol=0, atol=0)  # str(float) round-trips exactly
    p = str(tmp_path / "ta.txt")
    formats.save_poses(p, Ts, "TartanAir")
    rows = np.loadtxt(p)
    assert rows.shape == (5, 7)
    for T, r in zip(Ts, rows):  # tz tx ty qz qx qy qw (voldor_slam.py:327)
        np.testing.assert_allclose(r[:3], [T[2, 3], T[0, 3], T[1, 3]])
        qx, qy, qz, qw = r[4], r[5], r[3], r[6]
        assert abs(qx * qx + qy * qy + qz * qz + qw * qw - 1) < 1e-12
        R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                      [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                      [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
        np.testing.assert_allclose(R, T[:3, :3], atol=1e-12)


def test_flo_files_are_read_by_the_reference_cpp_loader(tmp_path):
    """voldor/utils.cpp:23-41 load_flow -- the C++ reader the reference's main.cpp uses -- compiled in place into oracle/_ref
    (ref_wrap_host.cpp) reads the files vk_write_flo and formats.save_flow write, value for value."""
    from oracle import orc
    from voldor_amd import capi, formats
    ref = orc.ref()
    if ref is None or not hasattr(ref, "ref_load_flow"):
        pytest.skip("oracle/_ref with the host pipeline is not built on this box")
    rng = np.random.default_rng(3)
    flow = rng.normal(0, 7, (37, 53, 2)).astype(np.float32)
    for k, write in enumerate((lambda p: formats.save_flow(p, flow),
                               lambda p: capi.lib().vk_write_flo(p.encode(), flow.ctypes.data_as(C.POINTER(C.c_float)), 53, 37))):
        p = str(tmp_path / f"f{k}.flo")
        write(p)
        w, h = C.c_int(0), C.c_int(0)
        got = np.zeros_like(flow)
        assert ref.ref_load_flow(p.encode(), C.byref(w), C.byref(h), got.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(got.size)) == 0
        assert (w.value, h.value) == (53, 37)
        np.testing.assert_array_equal(got, flow)


def test_pose_conversions_either_side_of_the_vo_call():
    """slam_py/slam_utils.py:55-93 (T44_to_T6 / T6_to_T44 / polish_T44; cv2.Rodrigues restated): round trips, the small-angle
    and the pi branch, dtype and batch handling, and agreement with the oracle's restatement of the same OpenCV function."""
    from oracle import orc
    from voldor_amd import slam_utils as su
    rng = np.random.default_rng(12)
    for scale in (1e-9, 1e-3, 0.3, 2.5):
        p6 = np.concatenate([rng.normal(0, scale, 3), rng.normal(0, 1, 3)]).astype(np.float32)
        T = su.T6_to_T44(p6)
        assert T.dtype == np.float32 and T.shape == (4, 4) and T[3, 3] == 1 and np.all(T[3, :3] == 0)
        np.testing.assert_allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=2e-6)
        np.testing.assert_allclose(T[:3, :3], orc.rvec_to_rotmat(p6[:3]), atol=1e-7)  # same cvRodrigues2 restatement in C
        back = su.T44_to_T6(T)
        np.testing.assert_allclose(back, p6, atol=2e-6 * max(1.0, scale))
    # rotation by pi about a tilted axis: the acos branch with s ~ 0
    axis = np.array([0.6, -0.48, 0.64])
    Rpi = 2 * np.outer(axis, axis) - np.eye(3)
    Tpi = np.eye(4); Tpi[:3, :3] = Rpi
    r = su.T44_to_T6(Tpi)[:3]
    assert abs(np.linalg.norm(r) - np.pi) < 1e-9 and (np.allclose(r / np.pi, axis, atol=1e-9) or np.allclose(r / np.pi, -axis, atol=1e-9))
    # batches, and the exact expression of voldor_slam.py:440 / :492
    P = rng.normal(0, 0.1, (5, 6))
    TT = su.T6_to_T44(P)
    assert TT.shape == (5, 4, 4) and TT.dtype == np.float64
    np.testing.assert_allclose(su.T44_to_T6(TT), P, atol=1e-12)
    Twc_cur, Tcw = su.T6_to_T44(P[0]), np.linalg.inv(su.T6_to_T44(P[1]))
    prior_pose = su.T44_to_T6(np.linalg.inv(Twc_cur @ Tcw))
    np.testing.assert_allclose(su.T6_to_T44(prior_pose) @ (Twc_cur @ Tcw), np.eye(4), atol=1e-12)
    noisy = TT[2].copy(); noisy[:3, :3] += rng.normal(0, 1e-3, (3, 3))
    su.polish_T44(noisy)
    np.testing.assert_allclose(noisy[:3, :3] @ noisy[:3, :3].T, np.eye(3), atol=1e-12)
    with pytest.raises(ValueError):
        su.T44_to_T6(np.zeros(4))
