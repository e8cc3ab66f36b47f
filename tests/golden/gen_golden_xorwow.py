"""Known answers for the XORWOW restatement of voldor_amd/csrc/vk_ref_cuda.h (reference mode, --reference_rng 1), produced by an
INDEPENDENT implementation: the 160-bit state as one Python integer, the subsequence skip of 2^67 outputs as a GF(2) matrix power by
repeated squaring on integers -- no code shared with the C header.  Writes tests/golden/ref_xorwow.npz:
  state/<sub>   v[0..4], d after curand_init(233, sub, 0)        out/<sub>  the first 8 outputs of curand()
  jump0         the 160 x 5 words of T^(2^67) in cuRAND's / rocRAND's layout (row 32 i + j = image of bit j of word i)
What this cannot pin (no CUDA here): the four seed-scramble constants of curand_init, taken from the public curand_kernel.h as
remembered; everything else is arithmetic.  python tests/golden/gen_golden_xorwow.py"""
import os

import numpy as np

M32 = 0xFFFFFFFF


def step_words(v):
    t = (v[0] ^ (v[0] >> 2)) & M32
    return [v[1], v[2], v[3], v[4], ((v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1))) & M32]


def pack(v):
    return sum(w << (32 * i) for i, w in enumerate(v))


def unpack(x):
    return [(x >> (32 * i)) & M32 for i in range(5)]


def build_T():
    return [pack(step_words(unpack(1 << b))) for b in range(160)]  # column b = image of basis bit b


def matvec(cols, x):
    r = 0
    b = 0
    while x:
        if x & 1:
            r ^= cols[b]
        x >>= 1
        b += 1
    return r


def matmul(A, B):  # A after B
    return [matvec(A, c) for c in B]


def matpow2(A, k):  # A^(2^k)
    for _ in range(k):
        A = matmul(A, A)
    return A


def curand_init(seed, sub):
    s0 = (seed & M32) ^ 0xaad26b49
    s1 = ((seed >> 32) & M32) ^ 0xf7dcefdd
    t0 = (1099087573 * s0) & M32
    t1 = (2591861531 * s1) & M32
    d = (6615241 + t1 + t0) & M32
    v = [(123456789 + t0) & M32, 362436069 ^ t0, (521288629 + t1) & M32, 88675123 ^ t1, (5783321 + t0) & M32]
    x = pack(v)
    J = matpow2(build_T(), 67)
    k = 0
    while sub >> k:
        if (sub >> k) & 1:
            x = matvec(J, x)
        J = matmul(J, J)
        k += 1
    return unpack(x), d


def outputs(v, d, n):
    out = []
    for _ in range(n):
        v = step_words(v)
        d = (d + 362437) & M32
        out.append((v[4] + d) & M32)
    return out


def main():
    res = {}
    for sub in (0, 1, 2, 3, 5, 255, 4096, 307199, 2073599, 0x7FFFFFFF):
        v, d = curand_init(233, sub)
        res[f"state/{sub}"] = np.array(v + [d], np.uint32)
        res[f"out/{sub}"] = np.array(outputs(v, d, 8), np.uint32)
    # brute force for a short skip: 5 outputs skipped one by one = the matrix T^5 (the algebra of the jump, checked on a small power)
    T = build_T()
    x = pack([1, 2, 3, 4, 5])
    y = x
    for _ in range(5):
        y = pack(step_words(unpack(y)))
    T5 = matmul(T, matmul(T, matmul(T, matmul(T, T))))
    assert matvec(T5, x) == y
    J0 = matpow2(T, 67)
    res["jump0"] = np.array([unpack(c) for c in J0], np.uint32)  # [160][5]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_xorwow.npz")
    np.savez_compressed(path, **res)
    print("wrote", path, {k: v.tolist() for k, v in res.items() if k.startswith("out/") and k in ("out/0", "out/1", "out/307199")})


if __name__ == "__main__":
    main()
