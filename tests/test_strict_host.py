"""CPU tier of the strict-math mode: the software transcendentals are accurate, the host build of the
product's strict residual model gives the oracle's bits, and the strict oracle is the same estimator as the glibc one."""
import ctypes as C
import math
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ulps(got, ref64):
    ref = ref64.astype(np.float32)
    spacing = np.abs(np.nextafter(ref, np.float32(np.inf)) - ref).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / np.maximum(spacing, 1e-300)


def test_strict_math_is_correctly_rounded_on_the_model_domain():
    """probe ops 6-12 of the test library = vsm_expf / logf / powf / atan2f / sinf / cosf / cbrtf, host build"""
    import hooks
    rng = np.random.default_rng(0)
    n = 200000
    x = rng.uniform(-30, 30, n).astype(np.float32)
    y = np.exp(rng.uniform(-40, 40, n)).astype(np.float32)
    z = rng.uniform(-3, 3, n).astype(np.float32)
    a = rng.uniform(-1, 1, n).astype(np.float32); b = rng.uniform(-1, 1, n).astype(np.float32)
    x64, y64, z64 = x.astype(np.float64), y.astype(np.float64), z.astype(np.float64)
    cases = [(6, x, x, np.exp(x64)), (7, y, y, np.log(y64)), (8, y, z, np.power(y64, z64)), (9, a, b, np.arctan2(a.astype(np.float64), b.astype(np.float64))),
             (10, x, x, np.sin(x64)), (11, x, x, np.cos(x64)), (12, x, x, np.cbrt(x64))]
    for op, p, q, ref in cases:
        got = hooks.probe(op, p, q, False).astype(np.float32)
        fin = np.isfinite(ref) & (np.abs(ref) < 3e38) & (np.abs(ref) > 1e-37)
        assert _ulps(got[fin], ref[fin]).max() <= 0.5 + 1e-6, op


def test_strict_math_special_values():
    import hooks
    f = lambda op, a, b=0.0: float(hooks.probe(op, np.array([a], np.float32), np.array([b], np.float32), False)[0])  # noqa: E731
    assert f(6, -200.0) == 0.0 and f(6, 200.0) == math.inf and f(6, 0.0) == 1.0
    assert f(7, 0.0) == -math.inf and f(7, 1.0) == 0.0
    assert f(8, 0.0, -1.0) == math.inf and f(8, 0.0, 2.0) == 0.0 and f(8, 5.0, 0.0) == 1.0 and f(8, 1.0, 1e30) == 1.0
    assert f(9, 0.0, -1.0) == float(np.float32(math.pi)) and f(9, 1.0, 0.0) == float(np.float32(math.pi / 2)) and f(9, 0.0, 0.0) == 0.0
    assert math.isnan(f(6, math.nan)) and math.isnan(f(9, math.nan, 1.0))


def test_host_build_of_the_strict_model_gives_the_oracle_bits(orc):
    """vk_strict_model.hpp (what the strict kernels call) compiled for the host vs the oracle with orc_set_strict_math(1)"""
    import hooks
    L = orc.lib(); H = hooks.lib()
    L.orc_set_strict_math(1)
    try:
        rng = np.random.default_rng(0)
        for i in range(20000):
            a = rng.normal(0, 5, 4).astype(np.float32)
            if i % 3 == 0:
                a[2:] = a[:2] + rng.normal(0, 0.05, 2).astype(np.float32)  # well-fitted pixels: rigidness near 1
            args = [float(v) for v in a] + [0.15, 1.0 if i % 2 else 0.5]
            o = np.float32(L.orc_fun_rigidness(*args)); h = np.float32(H.vkt_strict_rigidness(*args))
            assert o.tobytes() == h.tobytes(), args
            d = np.abs(rng.normal(5, 3, 2)).astype(np.float32) + 0.1
            args = [float(d[0]), float(d[1]), 160.0, 0.15, 1.0]
            o = np.float32(L.orc_fun_depth_rigidness(*args)); h = np.float32(H.vkt_strict_depth_rigidness(*args))
            assert o.tobytes() == h.tobytes(), args
        for i in range(2000):
            rv = rng.normal(0, 0.3 if i % 2 else 0.003, 3).astype(np.float32)
            Ro = np.zeros(9, np.float32); Rh = np.zeros(9, np.float32)
            L.orc_rvec_to_rotmat(rv.ctypes.data_as(C.POINTER(C.c_float)), Ro.ctypes.data_as(C.POINTER(C.c_float)))
            H.vkt_strict_rvec_to_rotmat(rv.ctypes.data_as(C.POINTER(C.c_float)), Rh.ctypes.data_as(C.POINTER(C.c_float)))
            assert Ro.tobytes() == Rh.tobytes()
    finally:
        L.orc_set_strict_math(0)


def test_pow_m2_shortcut_gives_the_plain_call_bits():
    """vk_strict_model.hpp pow_m2: 1 / (x x) in double where that provably rounds like vsm_powf(x, -2), the plain call elsewhere.  EVERY float of four
    binades, 2^24 random ones over the whole domain, the edges -- equal bits; the shortcut answers > 99.9 % of them; and the two doubles never lie more
    than a few hundred units of the last place apart, against the 2^17 the shortcut keeps from a float rounding boundary."""
    import hooks
    H = hooks.lib()
    H.vkt_pow_m2_host.restype = C.c_long
    H.vkt_pow_m2_host.argtypes = [C.POINTER(C.c_float), C.c_long, C.POINTER(C.c_long)]
    H.vkt_pow_m2_gap_host.restype = C.c_double
    H.vkt_pow_m2_gap_host.argtypes = [C.POINTER(C.c_float), C.c_long]
    rng = np.random.default_rng(5)
    sets = []
    for lo in (1.0, 2.0, 1024.0, 2.0 ** 40):  # every float of [lo, 2 lo)
        sets.append((np.arange(1 << 23, dtype=np.uint32) + np.float32(lo).view(np.uint32)).view(np.float32))
    sets.append(np.exp(rng.uniform(0, np.log(1e18), 1 << 24)).astype(np.float32))
    sets.append((1.0 + np.exp(rng.uniform(np.log(1e-7), 0, 1 << 22))).astype(np.float32))  # 1 + small: where fisk_pdf's argument sits for a poor fit
    sets.append(np.array([1.0, np.nextafter(np.float32(1), np.float32(2)), 1e18, np.nextafter(np.float32(1e18), np.float32(np.inf)), 3e38, np.inf, np.nan, 0.5, 0.0, -1.0, -np.inf], np.float32))
    for x in sets:
        x = np.ascontiguousarray(x, np.float32)
        nf = C.c_long(0)
        bad = H.vkt_pow_m2_host(x.ctypes.data_as(C.POINTER(C.c_float)), x.size, C.byref(nf))
        assert bad == 0, (bad, x[:4])
        inside = int(((x >= 1) & (x <= 1e18)).sum())
        if inside > 1000:
            assert nf.value > 0.999 * inside, (nf.value, inside)
        gap = H.vkt_pow_m2_gap_host(x.ctypes.data_as(C.POINTER(C.c_float)), min(x.size, 1 << 22))
        assert gap < 1024, gap  # (measured: 2 units of the last place near 1, 65 at x ~ 1e18)


def test_straight_line_model_gives_the_call_by_call_bits():
    """vk_strict_model.hpp rig_core (entry tests first, then the logarithms / exponentials / reciprocal squares of a residual side by side) against rig_core_plain
    (residual_model.h's calls one after the other): equal bits over the model's domain and far outside it -- zero, denormal, huge, infinite and NaN magnitudes."""
    import hooks
    H = hooks.lib()
    H.vkt_rig_core_host.restype = C.c_long
    H.vkt_rig_core_host.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_long]
    rng = np.random.default_rng(21)
    n = 1 << 21
    p = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    mag = np.exp(rng.uniform(np.log(1e-6), np.log(5000.0), n)).astype(np.float32)
    diff = np.exp(rng.uniform(np.log(1e-9), np.log(1e5), n)).astype(np.float32)
    k = rng.choice([0.15, 0.05, 0.3, 1.0], n).astype(np.float32)
    assert H.vkt_rig_core_host(p(mag), p(diff), p(k), n) == 0
    wild = np.array([0.0, -0.0, 1e-45, 1e-38, 1.0, 2.0, 4.0, 200.0, 1e10, 1e30, 3.4e38, np.inf, -np.inf, np.nan, -1.0, 1.1920929e-07, 2.3841858e-07], np.float32)
    M, D, K = np.meshgrid(wild, wild, np.array([0.15, 0.0, np.nan, np.inf, 1e30, -0.15], np.float32), indexing="ij")
    assert H.vkt_rig_core_host(p(M.ravel()), p(D.ravel()), p(K.ravel()), M.size) == 0
    # differences that put r = x^2 / s at exactly 1 and around it (the base the plain pow answers without a logarithm)
    g = np.clip(mag * 0.5, 2, 100).astype(np.float32)
    s_ = (np.float32(0.01) * np.exp(np.float32(0.09) * g)).astype(np.float32)
    d1 = (2 * np.sqrt(s_.astype(np.float64))).astype(np.float32)
    for nudge in (0, 1, -1, 2, -2):
        dd = (d1.view(np.int32) + nudge).view(np.float32)
        assert H.vkt_rig_core_host(p(mag), p(dd), p(k), n) == 0


def test_strict_oracle_is_the_same_estimator(orc):
    """switching the libm does not change what is estimated: against ground truth the strict oracle is as accurate as the
    glibc one, and the residual model agrees to float rounding"""
    from voldor_amd import synth
    L = orc.lib()
    rng = np.random.default_rng(1)
    a = rng.normal(0, 4, (3000, 4)).astype(np.float32)
    r0 = np.array([L.orc_fun_rigidness(*[float(v) for v in q], 0.15, 1.0) for q in a])
    L.orc_set_strict_math(1)
    try:
        r1 = np.array([L.orc_fun_rigidness(*[float(v) for v in q], 0.15, 1.0) for q in a])
        sc = synth.make_scene(w=160, h=120, n_flows=3, fx=80, fy=80, cx=80, cy=60, seed=5)
        fx, fy, cx, cy = sc["K"]
        cfg = "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 3"
        s = orc.voldor(sc["flows"], fx, fy, cx, cy, config=cfg)
    finally:
        L.orc_set_strict_math(0)
    g = orc.voldor(sc["flows"], fx, fy, cx, cy, config=cfg)
    assert np.abs(r0 - r1).max() < 2e-6
    gt = sc["poses_gt"].copy(); gt[:, 3:] /= np.mean(np.linalg.norm(gt[:, 3:], axis=1))
    assert s["n_registered"] == g["n_registered"] == 3
    rs, ts = synth.pose_errors(s["poses"], gt); rg, tg = synth.pose_errors(g["poses"], gt)
    assert rs.max() < 2 * rg.max() + 1e-3 and ts.max() < 2 * tg.max() + 1e-2
