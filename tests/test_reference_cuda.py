"""Reference mode, round 4: D1 (cuRAND XORWOW) and D2 (CUDA linear texture filtering) restated in voldor_amd/csrc/vk_ref_cuda.h.
CPU tests of the restatement through the oracle library (which compiles that header):
  * the XORWOW stream against known answers of an INDEPENDENT big-integer implementation (tests/golden/gen_golden_xorwow.py):
    state after curand_init(233, sub, 0) and the first outputs, for subsequences 0, 1, .. 307199 (the last pixel of 640x480), 2073599
    (of 1920x1080), 2^31 - 1;
  * the computed skip-ahead matrix T^(2^67) against rocRAND's PUBLISHED table of the same recurrence
    (/opt/rocm/include/rocrand/rocrand_xorwow_precomputed.h, h_xorwow_sequence_jump_matrices[0]) -- an implementation by other people;
  * curand_uniform's range and end points;
  * the texture rule: texel centres exact, 8-bit fractions, bleed into the next layer of the stack, clamp only at the ends of the stack,
    and the distance to the exact bilinear value bounded by the quantisation step.
What no test here can pin: the four seed-scramble constants of curand_init (from the public curand_kernel.h as remembered) and the two
choices the CUDA guide leaves open (rounding of the fraction to 8 bits, order of the four products): stated in the header."""
import ctypes as C
import os
import re

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_xorwow.npz")


@pytest.fixture(scope="module")
def L(orc):
    lib = orc.lib()
    lib.orc_xorwow_stream.argtypes = [C.c_ulonglong, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_xorwow_jumps.restype = C.POINTER(C.c_uint32)
    lib.orc_fetch1.restype = C.c_float
    lib.orc_fetch1.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
    lib.orc_fetch2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    return lib


def _stream(L, sub, n=8):
    raw = np.zeros(n, np.uint32); uni = np.zeros(n, np.float32); st = np.zeros(6, np.uint32)
    L.orc_xorwow_stream(233, sub, n, raw.ctypes.data, uni.ctypes.data, st.ctypes.data)
    return st, raw, uni


def test_xorwow_stream_matches_the_independent_implementation(L):
    g = np.load(GOLD)
    subs = sorted(int(k.split("/")[1]) for k in g.files if k.startswith("state/"))
    assert 307199 in subs and 2073599 in subs
    for sub in subs:
        st, raw, uni = _stream(L, sub)
        np.testing.assert_array_equal(st, g[f"state/{sub}"], err_msg=f"state after curand_init(233, {sub}, 0)")
        np.testing.assert_array_equal(raw, g[f"out/{sub}"], err_msg=f"outputs of subsequence {sub}")
        assert np.all(uni > 0) and np.all(uni <= 1)
        np.testing.assert_array_equal(uni, (raw.astype(np.float32) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33)).astype(np.float32))
    # the Weyl word is untouched by the subsequence skip (362437 * 2^67 = 0 mod 2^32)
    assert len({int(g[f"state/{s}"][5]) for s in subs}) == 1


def test_sequence_jump_matrix_equals_rocrands_published_table(L):
    J = np.ctypeslib.as_array(L.orc_xorwow_jumps(), shape=(32, 160, 5)).copy()
    g = np.load(GOLD)
    np.testing.assert_array_equal(J[0], g["jump0"])  # the independent implementation's T^(2^67)
    hdr = "/opt/rocm/include/rocrand/rocrand_xorwow_precomputed.h"
    if not os.path.exists(hdr):
        pytest.skip("no rocRAND headers on this box")
    txt = open(hdr).read()
    m = re.search(r"h_xorwow_sequence_jump_matrices\[XORWOW_JUMP_MATRICES\]\[XORWOW_SIZE\]\s*=\s*\{(.*?)\n\};", txt, re.S)
    assert m, "table not found"
    nums = np.array([int(x) for x in re.findall(r"\b\d+\b", re.sub(r"//[^\n]*", "", m.group(1)))], np.uint64)
    assert nums.size == 32 * 800, nums.size
    tab = nums.reshape(32, 160, 5).astype(np.uint32)
    # rocRAND: matrices A^(2^67), A^(4 * 2^67), A^(16 * 2^67) ... (XORWOW_JUMP_LOG2 = 2): its k-th table = our J[2 k] for 2 k < 32
    for k in range(16):
        np.testing.assert_array_equal(tab[k], J[2 * k], err_msg=f"A^(2^67 * 4^{k})")


def test_curand_uniform_end_points(L):
    # x = 2^32 - 1 rounds to 2^32 in float: the draw is exactly 1.0, the reference's index (int)(u * N) is then one past the end (SURVEY B-4)
    f = lambda x: np.float32(np.float32(x) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33))
    assert f(np.uint32(0xFFFFFFFF)) == np.float32(1.0) and f(np.uint32(0)) == np.float32(2.0 ** -33) and f(np.uint32(1 << 31)) == np.float32(0.5)


def _tex(L, on):
    L.orc_set_reference_tex(int(on))


def test_texture_rule_properties(L):
    rng = np.random.default_rng(5)
    w, h, n = 9, 7, 3
    st = rng.uniform(-5, 5, (n, h, w)).astype(np.float32)
    f1 = lambda x, y, f: float(L.orc_fetch1(st.ctypes.data, f, n, w, h, C.c_float(x), C.c_float(y)))
    try:
        _tex(L, True)
        # texel centres are exact
        for f in range(n):
            for y in (0, 3, h - 1):
                for x in (0, 4, w - 1):
                    assert f1(x, y, f) == st[f, y, x]
        # fractions are multiples of 1/256: positions inside one cell of the 1/256 grid give one value; the value is the 8-bit lerp
        a = f1(2.0 + 37 / 256.0, 3.0, 1)
        assert a == f1(2.0 + 37 / 256.0 + 0.0015, 3.0, 1) and a == f1(2.0 + 37 / 256.0 - 0.0015, 3.0, 1)
        al = np.float32(37 / 256.0)
        assert a == np.float32(np.float32((1 - al) * np.float32(1.0)) * st[1, 3, 2] + np.float32(al * np.float32(1.0)) * st[1, 3, 3])
        # the row below the last row of layer 0 is row 0 of layer 1 (one texture over the stack); layer n-1 clamps
        b = np.float32(0.5)
        assert f1(4.0, h - 1 + 0.5, 0) == np.float32((1 - b) * st[0, h - 1, 4] + b * st[1, 0, 4])
        assert f1(4.0, h - 1 + 0.5, n - 1) == st[n - 1, h - 1, 4]
        # x clamps at both ends, y at the top of the stack
        assert f1(-0.4, 2.0, 0) == st[0, 2, 0] and f1(w - 1 + 0.4, 2.0, 0) == st[0, 2, w - 1] and f1(3.0, -0.3, 0) == st[0, 0, 3]
        # against the exact bilinear value: off by at most the quantisation of the two fractions
        _tex(L, False)
        pts = [(float(rng.uniform(0, w - 1)), float(rng.uniform(0, h - 1.001)), int(rng.integers(0, n))) for _ in range(400)]
        exact = [f1(x, y, f) for x, y, f in pts]
        _tex(L, True)
        quant = [f1(x, y, f) for x, y, f in pts]
        span = float(st.max() - st.min())
        assert max(abs(a - b) for a, b in zip(exact, quant)) <= 2 * span / 512 + 1e-5
        assert exact != quant
    finally:
        _tex(L, False)
