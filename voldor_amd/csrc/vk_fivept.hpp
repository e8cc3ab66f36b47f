// voldor_amd/csrc/vk_fivept.hpp -- the five-point relative-pose minimal solver (Nister, "An efficient solution to the five-point relative
// pose problem", PAMI 26(6), 2004), the solver behind cv::findEssentialMat, which the reference's monocular bootstrap calls
// (voldor/geometry.cpp:316-326: findEssentialMat(pts1, pts2, K, LMEDS, 0.999, 1.0) + recoverPose; OpenCV is not part of the reference tree,
// so this is a restatement of the PUBLISHED algorithm, not of OpenCV's code: parity with OpenCV's numerics stays unpinned, row f1).
//
//   1. five correspondences q'^T E q = 0  ->  5 x 9 system; E = x X + y Y + z Z + W over a basis X, Y, Z, W of its null space
//   2. det E = 0 and 2 E E^T E - trace(E E^T) E = 0: ten cubic equations in (x, y, z), a 10 x 20 coefficient matrix over the monomials
//      x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1
//   3. Gauss-Jordan on the first ten columns; rows e..j (leading x^2z, x^2, y^2z, y^2, xyz, xy) combine to k = e - z f, l = g - z h,
//      m = i - z j: three equations B(z) (x, y, 1)^T = 0 with polynomial entries (degrees 3, 3, 4)
//   4. det B(z) = a degree-10 polynomial: its real roots z (Laguerre's method with deflation, polished on the full polynomial);
//      (x, y, 1) = the null vector of B(z) (x = p1 / p3, y = p2 / p3 of the paper for rows 0 and 1; here the best-conditioned row pair)
//   5. every (x, y, z) is polished by Gauss-Newton on the ten cubic constraints of step 2 (the polynomial is ill conditioned where its
//      roots cluster -- small baselines, forward motion --, the constraints are not)
// Up to ten essential matrices per sample.  Plain double arithmetic, host and device from one source (the host build is what the CPU tests
// call: vk_fivept_solve).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include "vk_p3p.hpp"  // VK_HD, vk_abs, vk_sqrt

namespace vk {
namespace fivept {

// index of x^a y^b z^c among the 20 monomials of degree <= 3 in the order above
VK_HD inline int mono3(int a, int b, int c) {
    switch (a * 16 + b * 4 + c) {
        case 3 * 16: return 0;          case 3 * 4: return 1;            case 2 * 16 + 4: return 2;      case 16 + 2 * 4: return 3;
        case 2 * 16 + 1: return 4;      case 2 * 16: return 5;           case 2 * 4 + 1: return 6;       case 2 * 4: return 7;
        case 16 + 4 + 1: return 8;      case 16 + 4: return 9;           case 16 + 2: return 10;         case 16 + 1: return 11;
        case 16: return 12;             case 4 + 2: return 13;           case 4 + 1: return 14;          case 4: return 15;
        case 3: return 16;              case 2: return 17;               case 1: return 18;              default: return 19;
    }
}
// degree <= 2: x^2 y^2 z^2 xy xz yz x y z 1
VK_HD inline int mono2(int a, int b, int c) {
    switch (a * 16 + b * 4 + c) {
        case 2 * 16: return 0; case 2 * 4: return 1; case 2: return 2; case 16 + 4: return 3; case 16 + 1: return 4;
        case 4 + 1: return 5;  case 16: return 6;    case 4: return 7; case 1: return 8;      default: return 9;
    }
}
VK_HD inline void exps1(int i, int& a, int& b, int& c) { a = i == 0; b = i == 1; c = i == 2; }  // x y z 1
VK_HD inline void exps2(int i, int& a, int& b, int& c) {
    const int ea[10] = { 2, 0, 0, 1, 1, 0, 1, 0, 0, 0 }, eb[10] = { 0, 2, 0, 1, 0, 1, 0, 1, 0, 0 }, ec[10] = { 0, 0, 2, 0, 1, 1, 0, 0, 1, 0 };
    a = ea[i]; b = eb[i]; c = ec[i];
}
// p (degree 1) * q (degree 1) -> degree 2, accumulated with factor s
VK_HD inline void mul11(const double* p, const double* q, double s, double* out10) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            int a, b, c, d, e, f;
            exps1(i, a, b, c); exps1(j, d, e, f);
            out10[mono2(a + d, b + e, c + f)] += s * p[i] * q[j];
        }
}
// p (degree 2) * q (degree 1) -> degree 3, accumulated with factor s
VK_HD inline void mul21(const double* p10, const double* q, double s, double* out20) {
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 4; j++) {
            int a, b, c, d, e, f;
            exps2(i, a, b, c); exps1(j, d, e, f);
            out20[mono3(a + d, b + e, c + f)] += s * p10[i] * q[j];
        }
}

// null space of the 5 x 9 epipolar system: N[4][9], rows X, Y, Z, W (Gaussian elimination with complete pivoting: the free columns
// span the null space).  Returns false for a rank-deficient sample.
VK_HD inline bool null_space_5x9(const double (*Q)[9], double (*N)[9]) {
    double A[5][9];
    int perm[9];
    for (int r = 0; r < 5; r++) for (int c = 0; c < 9; c++) A[r][c] = Q[r][c];
    for (int c = 0; c < 9; c++) perm[c] = c;
    for (int k = 0; k < 5; k++) {
        int pr = k, pc = k; double best = -1.0;
        for (int r = k; r < 5; r++) for (int c = k; c < 9; c++) { const double v = vk_abs(A[r][c]); if (v > best) { best = v; pr = r; pc = c; } }
        if (!(best > 1e-12)) return false;
        if (pr != k) for (int c = 0; c < 9; c++) { const double t = A[k][c]; A[k][c] = A[pr][c]; A[pr][c] = t; }
        if (pc != k) { for (int r = 0; r < 5; r++) { const double t = A[r][k]; A[r][k] = A[r][pc]; A[r][pc] = t; } const int t = perm[k]; perm[k] = perm[pc]; perm[pc] = t; }
        const double piv = A[k][k];
        for (int c = k; c < 9; c++) A[k][c] /= piv;
        for (int r = 0; r < 5; r++) if (r != k) { const double m = A[r][k]; if (m != 0.0) for (int c = k; c < 9; c++) A[r][c] -= m * A[k][c]; }
    }
    // reduced row echelon [I5 | F]: null vector j (free column 5 + j) = (-F[:, j], e_j), un-permuted; then orthonormalised (Gram-Schmidt)
    for (int j = 0; j < 4; j++) {
        double v[9];
        for (int r = 0; r < 5; r++) v[r] = -A[r][5 + j];
        for (int c = 0; c < 4; c++) v[5 + c] = c == j ? 1.0 : 0.0;
        for (int c = 0; c < 9; c++) N[j][perm[c]] = v[c];
    }
    for (int j = 0; j < 4; j++) {
        for (int i = 0; i < j; i++) { double d = 0; for (int c = 0; c < 9; c++) d += N[j][c] * N[i][c]; for (int c = 0; c < 9; c++) N[j][c] -= d * N[i][c]; }
        double n = 0; for (int c = 0; c < 9; c++) n += N[j][c] * N[j][c];
        n = vk_sqrt(n);
        if (!(n > 1e-12)) return false;
        for (int c = 0; c < 9; c++) N[j][c] /= n;
    }
    return true;
}

// the ten cubic constraints as a 10 x 20 matrix
VK_HD inline void constraint_matrix(const double (*N)[9], double (*A)[20]) {
    double e[9][4];  // E_ij as a degree-1 polynomial (x, y, z, 1)
    for (int k = 0; k < 9; k++) { e[k][0] = N[0][k]; e[k][1] = N[1][k]; e[k][2] = N[2][k]; e[k][3] = N[3][k]; }
    for (int r = 0; r < 10; r++) for (int c = 0; c < 20; c++) A[r][c] = 0.0;
    // det E
    {
        double m[10];
        const int cof[3][4] = { { 4, 8, 5, 7 }, { 3, 8, 5, 6 }, { 3, 7, 4, 6 } };
        for (int k = 0; k < 3; k++) {
            for (int c = 0; c < 10; c++) m[c] = 0.0;
            mul11(e[cof[k][0]], e[cof[k][1]], 1.0, m); mul11(e[cof[k][2]], e[cof[k][3]], -1.0, m);
            mul21(m, e[k], k == 1 ? -1.0 : 1.0, A[0]);
        }
    }
    // (E E^T - 1/2 trace(E E^T) I) E = 0
    double eet[3][3][10], tr[10];
    for (int c = 0; c < 10; c++) tr[c] = 0.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            for (int c = 0; c < 10; c++) eet[i][j][c] = 0.0;
            for (int k = 0; k < 3; k++) mul11(e[i * 3 + k], e[j * 3 + k], 1.0, eet[i][j]);
        }
    for (int i = 0; i < 3; i++) for (int c = 0; c < 10; c++) tr[c] += eet[i][i][c];
    for (int i = 0; i < 3; i++) for (int c = 0; c < 10; c++) eet[i][i][c] -= 0.5 * tr[c];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) mul21(eet[i][k], e[k * 3 + j], 1.0, A[1 + i * 3 + j]);
}

// reduced row echelon form over the first ten columns (partial pivoting); false if singular
VK_HD inline bool gauss_jordan_10x20(double (*A)[20]) {
    for (int k = 0; k < 10; k++) {
        int pr = k; double best = vk_abs(A[k][k]);
        for (int r = k + 1; r < 10; r++) { const double v = vk_abs(A[r][k]); if (v > best) { best = v; pr = r; } }
        if (!(best > 1e-14)) return false;
        if (pr != k) for (int c = 0; c < 20; c++) { const double t = A[k][c]; A[k][c] = A[pr][c]; A[pr][c] = t; }
        const double piv = A[k][k];
        for (int c = k; c < 20; c++) A[k][c] /= piv;
        for (int r = 0; r < 10; r++) if (r != k) { const double m = A[r][k]; if (m != 0.0) for (int c = k; c < 20; c++) A[r][c] -= m * A[k][c]; }
    }
    return true;
}

// polynomials as ascending coefficient arrays
VK_HD inline void pmul_acc(const double* a, int da, const double* b, int db, double s, double* out) {
    for (int i = 0; i <= da; i++) for (int j = 0; j <= db; j++) out[i + j] += s * a[i] * b[j];
}
VK_HD inline double peval(const double* p, int d, double z) { double v = p[d]; for (int i = d - 1; i >= 0; i--) v = v * z + p[i]; return v; }

struct Cx { double re, im; };
VK_HD inline Cx cadd(Cx a, Cx b) { return { a.re + b.re, a.im + b.im }; }
VK_HD inline Cx csub(Cx a, Cx b) { return { a.re - b.re, a.im - b.im }; }
VK_HD inline Cx cmul(Cx a, Cx b) { return { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; }
VK_HD inline Cx cscale(Cx a, double s) { return { a.re * s, a.im * s }; }
VK_HD inline double cabs_(Cx a) { return vk_sqrt(a.re * a.re + a.im * a.im); }
VK_HD inline Cx cdiv(Cx a, Cx b) { const double d = b.re * b.re + b.im * b.im; return { (a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d }; }
VK_HD inline Cx csqrt_(Cx a) {
    const double r = cabs_(a);
    if (r == 0.0) return { 0.0, 0.0 };
    const double w = vk_sqrt(0.5 * (r + vk_abs(a.re)));
    if (a.re >= 0.0) return { w, a.im / (2.0 * w) };
    return { vk_abs(a.im) / (2.0 * w), a.im >= 0.0 ? w : -w };
}
// one root of the complex polynomial c[0..m] (ascending) by Laguerre's method from x
VK_HD inline Cx laguerre(const Cx* c, int m, Cx x) {
    for (int it = 0; it < 80; it++) {
        Cx b = c[m], d = { 0, 0 }, f = { 0, 0 };
        double err = cabs_(b);
        const double ax = cabs_(x);
        for (int j = m - 1; j >= 0; j--) {
            f = cadd(cmul(x, f), d);
            d = cadd(cmul(x, d), b);
            b = cadd(cmul(x, b), c[j]);
            err = cabs_(b) + ax * err;
        }
        if (cabs_(b) <= err * 1e-15) return x;  // on the root to rounding
        const Cx g = cdiv(d, b), g2 = cmul(g, g), h = csub(g2, cscale(cdiv(f, b), 2.0));
        const Cx sq = csqrt_(cscale(csub(cscale(h, (double)m), g2), (double)(m - 1)));
        Cx gp = cadd(g, sq), gm = csub(g, sq);
        if (cabs_(gp) < cabs_(gm)) gp = gm;
        const Cx dx = cabs_(gp) > 0.0 ? cdiv({ (double)m, 0.0 }, gp) : Cx{ (1.0 + ax) * 0.7071, (1.0 + ax) * 0.7071 };
        const Cx x1 = csub(x, dx);
        if (x1.re == x.re && x1.im == x.im) return x;
        x = (it % 10 == 9) ? csub(x, cscale(dx, 0.37 + 0.06 * (it / 10))) : x1;  // break rare limit cycles
    }
    return x;
}
// real roots of the real polynomial p[0..deg] (ascending); returns their number
VK_HD inline int real_roots(const double* p, int deg, double* roots) {
    double scale = 0.0;
    for (int i = 0; i <= deg; i++) scale = vk_abs(p[i]) > scale ? vk_abs(p[i]) : scale;
    if (!(scale > 0.0)) return 0;
    while (deg > 0 && vk_abs(p[deg]) <= 1e-14 * scale) deg--;  // a vanishing leading coefficient: the root at infinity is dropped
    if (deg < 1) return 0;
    Cx c[11], work[11];
    for (int i = 0; i <= deg; i++) { c[i] = { p[i] / scale, 0.0 }; work[i] = c[i]; }
    int n = 0;
    for (int m = deg; m >= 1; m--) {
        Cx x = laguerre(work, m, { 0.0, 0.0 });
        x = laguerre(c, deg, x);  // polish on the undeflated polynomial
        // A real root, or one member of a near-double real pair: where two real roots nearly coincide (forward motion, small baselines) the
        // polynomial's rounding turns them into a complex pair re +- i im with a small imaginary part.  Such a pair is tried as the two real
        // starts re -+ |im| (the conjugate produces the same two: the duplicate test drops them); a start that belongs to no real root ends up
        // as a model whose residual the LMedS discards.  (Round 4, found with the independent solver oracle/orc_fivept.py: with the strict
        // 1e-7 test alone the planted motion was missing in a quarter of the forward-motion samples.)
        const double tol_im = vk_abs(x.im) / (1.0 + vk_abs(x.re));
        if (tol_im <= 1e-3) {
            const int n_start = tol_im <= 1e-7 ? 1 : 2;
            for (int st = 0; st < n_start; st++) {
                double z = n_start == 1 ? x.re : x.re + (st == 0 ? -vk_abs(x.im) : vk_abs(x.im));
                for (int it = 0; it < 3; it++) {  // real Newton steps on the real polynomial
                    double v = p[deg], dv = 0.0;
                    for (int i = deg - 1; i >= 0; i--) { dv = dv * z + v; v = v * z + p[i]; }
                    if (dv != 0.0) z -= v / dv;
                }
                bool dup = false;
                for (int k = 0; k < n; k++) dup = dup || vk_abs(roots[k] - z) <= 1e-9 * (1.0 + vk_abs(z));
                if (!dup && n < 10) roots[n++] = z;
            }
        }
        // deflate by (t - x): synthetic division from the top
        Cx rem = work[m];
        for (int j = m - 1; j >= 0; j--) { const Cx t = work[j]; work[j] = rem; rem = cadd(cmul(x, rem), t); }
    }
    return n;
}

// the solutions E = x N0 + y N1 + z N2 + N3 for ONE basis N of the null space (steps 2 .. 5)
VK_HD inline int solve_basis(const double (*N)[9], double (*Es)[9], double* dbg_poly = nullptr, double* dbg_roots = nullptr, int* dbg_nroots = nullptr) {
    double A[10][20];
    constraint_matrix(N, A);
    double A0[10][20];  // the constraints before the elimination: every solution is polished on them (below)
    for (int r = 0; r < 10; r++) for (int c = 0; c < 20; c++) A0[r][c] = A[r][c];
    if (!gauss_jordan_10x20(A)) return 0;
    // B(z): rows k = e - z f, l = g - z h, m = i - z j; columns: coefficient of x (degree 3), of y (degree 3), constant (degree 4)
    double B[3][3][5];
    for (int r = 0; r < 3; r++) {
        const double* e = A[4 + 2 * r]; const double* f = A[5 + 2 * r];
        double* bx = B[r][0]; double* by = B[r][1]; double* b1 = B[r][2];
        bx[0] = e[12]; bx[1] = e[11] - f[12]; bx[2] = e[10] - f[11]; bx[3] = -f[10]; bx[4] = 0.0;
        by[0] = e[15]; by[1] = e[14] - f[15]; by[2] = e[13] - f[14]; by[3] = -f[13]; by[4] = 0.0;
        b1[0] = e[19]; b1[1] = e[18] - f[19]; b1[2] = e[17] - f[18]; b1[3] = e[16] - f[17]; b1[4] = -f[16];
    }
    double p1[8], p2[8], p3[8], n[11];
    for (int i = 0; i < 8; i++) { p1[i] = 0.0; p2[i] = 0.0; p3[i] = 0.0; }
    for (int i = 0; i < 11; i++) n[i] = 0.0;
    pmul_acc(B[0][1], 3, B[1][2], 4, 1.0, p1); pmul_acc(B[0][2], 4, B[1][1], 3, -1.0, p1);  // b12 b23 - b13 b22
    pmul_acc(B[0][2], 4, B[1][0], 3, 1.0, p2); pmul_acc(B[0][0], 3, B[1][2], 4, -1.0, p2);  // b13 b21 - b11 b23
    pmul_acc(B[0][0], 3, B[1][1], 3, 1.0, p3); pmul_acc(B[0][1], 3, B[1][0], 3, -1.0, p3);  // b11 b22 - b12 b21
    pmul_acc(p1, 7, B[2][0], 3, 1.0, n); pmul_acc(p2, 7, B[2][1], 3, 1.0, n); pmul_acc(p3, 6, B[2][2], 4, 1.0, n);
    double zs[10];
    const int nz = real_roots(n, 10, zs);
    if (dbg_poly) for (int i = 0; i < 11; i++) dbg_poly[i] = n[i];
    if (dbg_roots) for (int i = 0; i < nz; i++) dbg_roots[i] = zs[i];
    if (dbg_nroots) *dbg_nroots = nz;
    int ne = 0;
    for (int k = 0; k < nz; k++) {
        const double z = zs[k];
        // (x, y, 1) spans the null space of B(z): the cross product of the best-conditioned pair of its rows (with rows 0 and 1 this is
        // x = p1 / p3, y = p2 / p3 of the paper; a fixed pair loses digits where its 2 x 2 minor is small)
        double Bz[3][3];
        for (int r = 0; r < 3; r++) { Bz[r][0] = peval(B[r][0], 3, z); Bz[r][1] = peval(B[r][1], 3, z); Bz[r][2] = peval(B[r][2], 4, z); }
        double best[3] = { 0, 0, 0 }, bestn = -1.0;
        for (int a = 0; a < 3; a++)
            for (int b = a + 1; b < 3; b++) {
                const double c[3] = { Bz[a][1] * Bz[b][2] - Bz[a][2] * Bz[b][1], Bz[a][2] * Bz[b][0] - Bz[a][0] * Bz[b][2], Bz[a][0] * Bz[b][1] - Bz[a][1] * Bz[b][0] };
                const double na = Bz[a][0] * Bz[a][0] + Bz[a][1] * Bz[a][1] + Bz[a][2] * Bz[a][2], nb = Bz[b][0] * Bz[b][0] + Bz[b][1] * Bz[b][1] + Bz[b][2] * Bz[b][2];
                const double nn = (c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) / (na * nb > 0.0 ? na * nb : 1.0);  // sin^2 of the angle between the rows
                if (nn > bestn) { bestn = nn; best[0] = c[0]; best[1] = c[1]; best[2] = c[2]; }
            }
        if (!(vk_abs(best[2]) > 1e-300)) continue;
        double x = best[0] / best[2], y = best[1] / best[2], zz = z;
        // Gauss-Newton on the ten cubic constraints themselves: the degree-10 polynomial is ill conditioned where its roots cluster (small
        // baselines, forward motion: a root is then good to 1e-6 only), the constraints in (x, y, z) are not
        for (int it = 0; it < 4; it++) {
            const double X = x, Y = y, Z = zz;
            const double mo[20] = { X * X * X, Y * Y * Y, X * X * Y, X * Y * Y, X * X * Z, X * X, Y * Y * Z, Y * Y, X * Y * Z, X * Y, X * Z * Z, X * Z, X, Y * Z * Z, Y * Z, Y, Z * Z * Z, Z * Z, Z, 1.0 };
            const double dx[20] = { 3 * X * X, 0, 2 * X * Y, Y * Y, 2 * X * Z, 2 * X, 0, 0, Y * Z, Y, Z * Z, Z, 1.0, 0, 0, 0, 0, 0, 0, 0 };
            const double dy[20] = { 0, 3 * Y * Y, X * X, 2 * X * Y, 0, 0, 2 * Y * Z, 2 * Y, X * Z, X, 0, 0, 0, Z * Z, Z, 1.0, 0, 0, 0, 0 };
            const double dz[20] = { 0, 0, 0, 0, X * X, 0, Y * Y, 0, X * Y, 0, 2 * X * Z, X, 0, 2 * Y * Z, Y, 0, 3 * Z * Z, 2 * Z, 1.0, 0 };
            double JtJ[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } }, Jtr[3] = { 0, 0, 0 };
            for (int r = 0; r < 10; r++) {
                double res = 0, j0 = 0, j1 = 0, j2 = 0;
                for (int c = 0; c < 20; c++) { res += A0[r][c] * mo[c]; j0 += A0[r][c] * dx[c]; j1 += A0[r][c] * dy[c]; j2 += A0[r][c] * dz[c]; }
                const double J[3] = { j0, j1, j2 };
                for (int a = 0; a < 3; a++) { Jtr[a] += J[a] * res; for (int b = 0; b < 3; b++) JtJ[a][b] += J[a] * J[b]; }
            }
            const double det = JtJ[0][0] * (JtJ[1][1] * JtJ[2][2] - JtJ[1][2] * JtJ[2][1]) - JtJ[0][1] * (JtJ[1][0] * JtJ[2][2] - JtJ[1][2] * JtJ[2][0]) +
                               JtJ[0][2] * (JtJ[1][0] * JtJ[2][1] - JtJ[1][1] * JtJ[2][0]);
            if (!(vk_abs(det) > 1e-300)) break;
            const double d0 = (Jtr[0] * (JtJ[1][1] * JtJ[2][2] - JtJ[1][2] * JtJ[2][1]) - JtJ[0][1] * (Jtr[1] * JtJ[2][2] - JtJ[1][2] * Jtr[2]) + JtJ[0][2] * (Jtr[1] * JtJ[2][1] - JtJ[1][1] * Jtr[2])) / det;
            const double d1 = (JtJ[0][0] * (Jtr[1] * JtJ[2][2] - JtJ[1][2] * Jtr[2]) - Jtr[0] * (JtJ[1][0] * JtJ[2][2] - JtJ[1][2] * JtJ[2][0]) + JtJ[0][2] * (JtJ[1][0] * Jtr[2] - Jtr[1] * JtJ[2][0])) / det;
            const double d2 = (JtJ[0][0] * (JtJ[1][1] * Jtr[2] - Jtr[1] * JtJ[2][1]) - JtJ[0][1] * (JtJ[1][0] * Jtr[2] - Jtr[1] * JtJ[2][0]) + Jtr[0] * (JtJ[1][0] * JtJ[2][1] - JtJ[1][1] * JtJ[2][0])) / det;
            if (!(d0 == d0 && d1 == d1 && d2 == d2)) break;
            const double step = vk_abs(d0) + vk_abs(d1) + vk_abs(d2), size = 1.0 + vk_abs(X) + vk_abs(Y) + vk_abs(Z);
            if (step > 0.5 * size) break;  // not in the basin of a solution: keep what the polynomial gave
            x = X - d0; y = Y - d1; zz = Z - d2;
            if (step <= 1e-15 * size) break;
        }
        {  // what is not a solution of the ten cubics after the polish is dropped (starts of the near-double-root handling that belong to no real root)
            const double X = x, Y = y, Z = zz;
            const double mo[20] = { X * X * X, Y * Y * Y, X * X * Y, X * Y * Y, X * X * Z, X * X, Y * Y * Z, Y * Y, X * Y * Z, X * Y, X * Z * Z, X * Z, X, Y * Z * Z, Y * Z, Y, Z * Z * Z, Z * Z, Z, 1.0 };
            double worst = 0.0;
            for (int r = 0; r < 10; r++) {
                double res = 0, mag = 0;
                for (int c = 0; c < 20; c++) { res += A0[r][c] * mo[c]; mag += vk_abs(A0[r][c] * mo[c]); }
                const double rel = mag > 0.0 ? vk_abs(res) / mag : 0.0;
                worst = rel > worst ? rel : worst;
            }
            if (!(worst <= 1e-9)) continue;
        }
        double E[9], nn = 0.0;
        for (int c = 0; c < 9; c++) { E[c] = x * N[0][c] + y * N[1][c] + zz * N[2][c] + N[3][c]; nn += E[c] * E[c]; }
        if (!(nn > 0.0) || !(nn < 1e300)) continue;
        const double s = 1.4142135623730951 / vk_sqrt(nn);
        for (int c = 0; c < 9; c++) Es[ne][c] = E[c] * s;
        ne++;
    }
    return ne;
}

// q1, q2: five normalised correspondences (image 1, image 2), [5][2].  Es: up to ten essential matrices (row-major, Frobenius norm sqrt 2).
// The basis of the four-dimensional null space is arbitrary, the solutions are not: where the roots of the degree-10 polynomial in z cluster (pure
// forward motion, small baselines: near-multiple roots get lost) they are spread out in another basis.  The system is therefore solved in TWO bases --
// the one the elimination produced and a fixed rotation of it -- and the union is returned (round 4; measured against the independent solver
// oracle/orc_fivept.py, tests/test_fivept.py).
VK_HD inline int solve(const double (*q1)[2], const double (*q2)[2], double (*Es)[9], double* dbg_poly = nullptr, double* dbg_roots = nullptr, int* dbg_nroots = nullptr) {
    double Q[5][9], N[4][9];
    for (int k = 0; k < 5; k++) {
        const double a[9] = { q2[k][0] * q1[k][0], q2[k][0] * q1[k][1], q2[k][0], q2[k][1] * q1[k][0], q2[k][1] * q1[k][1], q2[k][1], q1[k][0], q1[k][1], 1.0 };
        for (int c = 0; c < 9; c++) Q[k][c] = a[c];
    }
    if (!null_space_5x9(Q, N)) return 0;
    int ne = solve_basis(N, Es, dbg_poly, dbg_roots, dbg_nroots);
    // second basis: an orthogonal mix of the four vectors (a product of two plane rotations by 45 and ~59 degrees with a row swap: every new
    // vector has a share of every old one)
    const double c1 = 0.7071067811865476, s1 = 0.7071067811865476, c2 = 0.5144957554275265, s2 = 0.8574929257125441;
    double M[4][9], T[4][9], E2[10][9];
    for (int c = 0; c < 9; c++) {
        T[0][c] = c1 * N[0][c] + s1 * N[3][c]; T[3][c] = -s1 * N[0][c] + c1 * N[3][c];
        T[1][c] = c1 * N[1][c] + s1 * N[2][c]; T[2][c] = -s1 * N[1][c] + c1 * N[2][c];
    }
    for (int c = 0; c < 9; c++) {
        M[0][c] = c2 * T[0][c] + s2 * T[1][c]; M[1][c] = -s2 * T[0][c] + c2 * T[1][c];
        M[2][c] = c2 * T[2][c] + s2 * T[3][c]; M[3][c] = -s2 * T[2][c] + c2 * T[3][c];
    }
    const int n2 = solve_basis(M, E2);
    for (int k = 0; k < n2 && ne < 10; k++) {
        bool dup = false;
        for (int j = 0; j < ne && !dup; j++) {
            double dp = 0.0, dm = 0.0;
            for (int c = 0; c < 9; c++) { const double a = vk_abs(E2[k][c] - Es[j][c]), b2 = vk_abs(E2[k][c] + Es[j][c]); dp = a > dp ? a : dp; dm = b2 > dm ? b2 : dm; }
            dup = (dp < dm ? dp : dm) <= 1e-6;
        }
        if (!dup) { for (int c = 0; c < 9; c++) Es[ne][c] = E2[k][c]; ne++; }
    }
    return ne;
}

}  // namespace fivept
}  // namespace vk
