"""ORACLE (test infrastructure only, see orc.h): the five-point relative-pose problem solved by a DIFFERENT algorithm than the product's.

The reference calls cv::findEssentialMat(..., LMEDS, ...) (voldor/geometry.cpp:316-326): OpenCV's minimal solver is the five-point algorithm.  The
product restates Nister's form of it (voldor_amd/csrc/vk_fivept.hpp: the 10 x 20 constraint matrix reduced to a degree-10 polynomial in z, roots by
Laguerre iteration).  This file solves the same polynomial system the way Stewenius, Engels and Nister (2006) do -- Gauss-Jordan elimination of the ten
cubic monomials, the 10 x 10 action matrix of multiplication by x on the quotient ring, its eigenvectors are the solutions -- with numpy's SVD and
eigen-solver: no shared code, no shared formulation past the ten constraints themselves, so agreement of the two solution sets checks both.
OpenCV's own numerics stay unpinned (not in the tree, not in the image)."""
import numpy as np

# monomials up to degree 3 in (x, y, z): the ten cubics first (graded order as in Stewenius et al.), then the basis of the quotient ring
MONO = [(3, 0, 0), (2, 1, 0), (2, 0, 1), (1, 2, 0), (1, 1, 1), (1, 0, 2), (0, 3, 0), (0, 2, 1), (0, 1, 2), (0, 0, 3),
        (2, 0, 0), (1, 1, 0), (1, 0, 1), (0, 2, 0), (0, 1, 1), (0, 0, 2), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]


def _pmul(a, b):
    """product of two polynomials stored as [4, 4, 4] coefficient arrays (x^i y^j z^k), total degree <= 3"""
    out = np.zeros((4, 4, 4))
    for (i, j, k), v in np.ndenumerate(a):
        if v != 0.0:
            for (l, m, n), w in np.ndenumerate(b):
                if w != 0.0 and i + l < 4 and j + m < 4 and k + n < 4:
                    out[i + l, j + m, k + n] += v * w
    return out


def solve(q1, q2):
    """q1, q2: [5, 2] normalised image points of the two views (q2^T E q1 = 0).  Returns the real solutions [n, 3, 3], Frobenius norm sqrt(2)."""
    q1 = np.asarray(q1, np.float64); q2 = np.asarray(q2, np.float64)
    h1 = np.concatenate([q1, np.ones((5, 1))], 1); h2 = np.concatenate([q2, np.ones((5, 1))], 1)
    A = np.stack([np.outer(h2[i], h1[i]).ravel() for i in range(5)])  # q2^T E q1 = <q2 q1^T, E>
    _, _, vt = np.linalg.svd(A)
    B = vt[5:9].reshape(4, 3, 3)  # E = x B0 + y B1 + z B2 + B3
    E = np.empty((3, 3), object)
    for r in range(3):
        for c in range(3):
            p = np.zeros((4, 4, 4)); p[1, 0, 0] = B[0, r, c]; p[0, 1, 0] = B[1, r, c]; p[0, 0, 1] = B[2, r, c]; p[0, 0, 0] = B[3, r, c]
            E[r, c] = p
    def mat(a, b):
        return np.array([[sum(_pmul(a[r, k], b[k, c]) for k in range(3)) for c in range(3)] for r in range(3)], object)
    Et = E.T
    EEt = mat(E, Et)
    tr = EEt[0, 0] + EEt[1, 1] + EEt[2, 2]
    EEtE = mat(EEt, E)
    eqs = [2.0 * EEtE[r, c] - _pmul(tr, E[r, c]) for r in range(3) for c in range(3)]
    det = (_pmul(E[0, 0], _pmul(E[1, 1], E[2, 2]) - _pmul(E[1, 2], E[2, 1])) - _pmul(E[0, 1], _pmul(E[1, 0], E[2, 2]) - _pmul(E[1, 2], E[2, 0]))
           + _pmul(E[0, 2], _pmul(E[1, 0], E[2, 1]) - _pmul(E[1, 1], E[2, 0])))
    eqs.append(det)
    M = np.array([[p[m] for m in MONO] for p in eqs])  # 10 x 20
    try:
        Bq = np.linalg.solve(M[:, :10], M[:, 10:])  # [I | Bq]: every cubic monomial in terms of the quotient-ring basis (with a minus sign)
    except np.linalg.LinAlgError:
        return np.zeros((0, 3, 3))
    # multiplication by x on the basis [x^2, xy, xz, y^2, yz, z^2, x, y, z, 1]
    Ax = np.zeros((10, 10))
    Ax[0:6] = -Bq[0:6]        # x * (x^2, xy, xz, y^2, yz, z^2) = x^3, x^2 y, x^2 z, x y^2, x y z, x z^2
    Ax[6, 0] = 1.0            # x * x = x^2
    Ax[7, 1] = 1.0            # x * y = xy
    Ax[8, 2] = 1.0            # x * z = xz
    Ax[9, 6] = 1.0            # x * 1 = x
    w, V = np.linalg.eig(Ax)
    out = []
    for i in range(10):
        if abs(w[i].imag) > 1e-9 * max(1.0, abs(w[i])):
            continue
        v = V[:, i].real
        if abs(v[9]) < 1e-14:
            continue
        x, y, z = v[6] / v[9], v[7] / v[9], v[8] / v[9]
        Ei = x * B[0] + y * B[1] + z * B[2] + B[3]
        out.append(Ei / np.linalg.norm(Ei) * np.sqrt(2.0))
    return np.array(out).reshape(-1, 3, 3)
