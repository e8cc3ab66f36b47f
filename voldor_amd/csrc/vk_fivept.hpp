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
//   4. det B(z) = a degree-10 polynomial: all its roots at once by the Aberth-Ehrlich iteration (round 6; rounds 4-5: Laguerre with deflation), the real ones kept;
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
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int a, b, c, d, e, f;
            exps1(i, a, b, c); exps1(j, d, e, f);
            out10[mono2(a + d, b + e, c + f)] += s * p[i] * q[j];
        }
}
// p (degree 2) * q (degree 1) -> degree 3, accumulated with factor s
VK_HD inline void mul21(const double* p10, const double* q, double s, double* out20) {
#pragma unroll
    for (int i = 0; i < 10; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int a, b, c, d, e, f;
            exps2(i, a, b, c); exps1(j, d, e, f);
            out20[mono3(a + d, b + e, c + f)] += s * p10[i] * q[j];
        }
}

// null space of the 5 x 9 epipolar system: N[4][9], rows X, Y, Z, W (Gaussian elimination with complete pivoting: the free columns
// span the null space).  Returns false for a rank-deficient sample.
// (a) the elimination: A [5][9] (a copy of the system) to reduced row echelon form [I5 | F] under row and column exchanges, perm = where the columns went
VK_HD inline bool ns_eliminate(double (*A)[9], int* perm) {
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
    return true;
}
// (b) null vector j (free column 5 + j) = (-F[:, j], e_j), un-permuted; then orthonormalised (Gram-Schmidt) -- on a local copy with constant trip
// counts: registers on the device
VK_HD inline bool ns_finish(const double (*A)[9], const int* perm, double (*Nout)[9]) {
    double N[4][9];
    int pm[9];
#pragma unroll
    for (int c = 0; c < 9; c++) pm[c] = perm[c];
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int c = 0; c < 9; c++) {
            const double v = c < 5 ? -A[c][5 + j] : ((c - 5) == j ? 1.0 : 0.0);
            // N[j][pm[c]] = v with a constant-index store: the one slot whose index matches takes the value
#pragma unroll
            for (int t = 0; t < 9; t++) if (pm[c] == t) N[j][t] = v;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int i = 0; i < j; i++) {
            double d = 0;
#pragma unroll
            for (int c = 0; c < 9; c++) d += N[j][c] * N[i][c];
#pragma unroll
            for (int c = 0; c < 9; c++) N[j][c] -= d * N[i][c];
        }
        double n = 0;
#pragma unroll
        for (int c = 0; c < 9; c++) n += N[j][c] * N[j][c];
        n = vk_sqrt(n);
        if (!(n > 1e-12)) return false;
#pragma unroll
        for (int c = 0; c < 9; c++) N[j][c] /= n;
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int c = 0; c < 9; c++) Nout[j][c] = N[j][c];
    return true;
}
VK_HD inline bool null_space_5x9_ws(const double (*Q)[9], double (*N)[9], double (*A)[9] /* [5][9] */, int* perm /* [9] */) {
    for (int r = 0; r < 5; r++) for (int c = 0; c < 9; c++) A[r][c] = Q[r][c];
    if (!ns_eliminate(A, perm)) return false;
    return ns_finish(A, perm, N);
}
VK_HD inline bool null_space_5x9(const double (*Q)[9], double (*N)[9]) {
    double A[5][9];
    int perm[9];
    return null_space_5x9_ws(Q, N, A, perm);
}

// the entries of E = x N0 + y N1 + z N2 + N3 as degree-1 polynomials (x, y, z, 1)
VK_HD inline void entry_polys(const double (*N)[9], double (*e)[4]) {
    for (int k = 0; k < 9; k++) { e[k][0] = N[0][k]; e[k][1] = N[1][k]; e[k][2] = N[2][k]; e[k][3] = N[3][k]; }
}
// ONE of the ten cubic constraints (row 0: det E; row 1 + 3 i + j: entry (i, j) of (E E^T - 1/2 trace(E E^T) I) E) into row_out[20].  Rows are independent: the device computes row r on lane r, the host one after the other -- every entry by the same operations in the same
// order as constraint_matrix below did in rounds 4-5 (each product polynomial accumulated term by term, the trace from the three diagonal products in
// order, the diagonal entry minus half of it).
VK_HD inline void constraint_row(const double (*e)[4], int r, double* row_out) {
    // (local arrays, constant trip counts: on the device every accumulator is a register -- the monomial index of a product term folds to a constant)
    double row[20];
#pragma unroll
    for (int c = 0; c < 20; c++) row[c] = 0.0;
    if (r == 0) {
        const int cof[3][4] = { { 4, 8, 5, 7 }, { 3, 8, 5, 6 }, { 3, 7, 4, 6 } };
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double m[10];
#pragma unroll
            for (int c = 0; c < 10; c++) m[c] = 0.0;
            mul11(e[cof[k][0]], e[cof[k][1]], 1.0, m); mul11(e[cof[k][2]], e[cof[k][3]], -1.0, m);
            mul21(m, e[k], k == 1 ? -1.0 : 1.0, row);
        }
    } else {
        const int i = (r - 1) / 3, j = (r - 1) % 3;
        double tr[10];
#pragma unroll
        for (int c = 0; c < 10; c++) tr[c] = 0.0;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            double t10[10];
#pragma unroll
            for (int c = 0; c < 10; c++) t10[c] = 0.0;
#pragma unroll
            for (int k = 0; k < 3; k++) mul11(e[d * 3 + k], e[d * 3 + k], 1.0, t10);
#pragma unroll
            for (int c = 0; c < 10; c++) tr[c] += t10[c];
        }
#pragma unroll
        for (int k2 = 0; k2 < 3; k2++) {
            double o[10];  // (E E^T)_{i k2}
#pragma unroll
            for (int c = 0; c < 10; c++) o[c] = 0.0;
#pragma unroll
            for (int k = 0; k < 3; k++) mul11(e[i * 3 + k], e[k2 * 3 + k], 1.0, o);
            if (k2 == i) {
#pragma unroll
                for (int c = 0; c < 10; c++) o[c] -= 0.5 * tr[c];
            }
            mul21(o, e[k2 * 3 + j], 1.0, row);
        }
    }
#pragma unroll
    for (int c = 0; c < 20; c++) row_out[c] = row[c];
}

// the ten cubic constraints as a 10 x 20 matrix
VK_HD inline void constraint_matrix_rows(const double (*N)[9], double (*A)[20]) {
    double e[9][4];
    entry_polys(N, e);
    for (int r = 0; r < 10; r++) constraint_row(e, r, A[r]);
}
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
#pragma unroll
    for (int i = 0; i <= da; i++) {
#pragma unroll
        for (int j = 0; j <= db; j++) out[i + j] += s * a[i] * b[j];
    }
}
VK_HD inline double peval(const double* p, int d, double z) { double v = p[d]; for (int i = d - 1; i >= 0; i--) v = v * z + p[i]; return v; }

struct Cx { double re, im; };
VK_HD inline Cx cadd(Cx a, Cx b) { return { a.re + b.re, a.im + b.im }; }
VK_HD inline Cx csub(Cx a, Cx b) { return { a.re - b.re, a.im - b.im }; }
VK_HD inline Cx cmul(Cx a, Cx b) { return { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; }
VK_HD inline double cabs_(Cx a) { return vk_sqrt(a.re * a.re + a.im * a.im); }
VK_HD inline Cx cdiv(Cx a, Cx b) { const double inv = 1.0 / (b.re * b.re + b.im * b.im); return { (a.re * b.re + a.im * b.im) * inv, (a.im * b.re - a.re * b.im) * inv }; }  // one division
VK_HD inline Cx cinv(Cx b) { const double inv = 1.0 / (b.re * b.re + b.im * b.im); return { b.re * inv, -b.im * inv }; }
VK_HD inline double cabs1(Cx a) { return vk_abs(a.re) + vk_abs(a.im); }  // |a|_1: |a| <= |a|_1 <= sqrt 2 |a|, no square root

// ---- all roots of the degree-10 polynomial at once: the Aberth-Ehrlich iteration (round 6) ------------------------------------------------
// Rounds 4-5 found the roots one after the other (Laguerre's method with deflation, each root polished on the full polynomial): ~20 dependent
// complex iterations of a degree-10 Horner scheme per root, ten roots, two bases -- the serial chain that made a five-point sample cost 1.9 ms on
// one lane.  The Aberth iteration moves ALL roots at once,
//     z_k <- z_k - w_k / (1 - w_k sum_{j != k} 1 / (z_k - z_j)),   w_k = p(z_k) / p'(z_k),
// every root from the values of the previous sweep (Jacobi order): one root per lane on the device, a loop over the roots on the host, the same
// operations on the same operands either way -- the host build stays the bit-for-bit reference of the kernels (tests/test_fivept.py).  No
// transcendental function anywhere (the starting points are constants): libm and the device library cannot disagree.
constexpr int ABERTH_SWEEPS = 48;
// a root's state between sweeps
struct RootState { Cx z; bool done; };
// starting points: ten points of the unit circle, turned by 0.7 rad (no symmetry about the real axis, which the roots of a real polynomial have)
VK_HD inline Cx aberth_start(int k) {
    const double c[10] = { 0.7648421872844885, 0.24010867170377775, -0.37633819547418595, -0.8490366632458126, -0.9974319833523372, -0.7648421872844884, -0.24010867170377764, 0.376338195474186, 0.8490366632458125, 0.9974319833523373 };   // cos(2 pi k / 10 + 0.7)
    const double s_[10] = { 0.644217687237691, 0.9707460150691567, 0.9264823595877222, 0.5283339327209796, -0.0716200990352768, -0.6442176872376911, -0.9707460150691568, -0.9264823595877222, -0.5283339327209797, 0.07162009903527669 };  // sin(2 pi k / 10 + 0.7)
    return { c[k], s_[k] };
}
// p and p' at z (Horner, complex argument, real coefficients ascending p[0..deg]) and a bound of the evaluation's rounding (the running sum of Laguerre's
// stopping rule, in the 1-norm: the dependent chain of a sweep carries no square root)
VK_HD inline void aberth_eval(const double* p, int deg, Cx z, Cx& pv, Cx& dv, double& err) {
    pv = { p[deg], 0.0 }; dv = { 0.0, 0.0 };
    err = vk_abs(p[deg]);
    const double az = cabs1(z);
    for (int i = deg - 1; i >= 0; i--) {
        dv = cadd(cmul(z, dv), pv);
        pv = cadd(cmul(z, pv), Cx{ p[i], 0.0 });
        err = cabs1(pv) + az * err;
    }
}
// the new value of root k given every root of the previous sweep (zs[0 .. deg - 1]; the loop over them has a constant bound so that a caller whose
// zs live in registers gets constant indices); returns the state after the move
VK_HD inline RootState aberth_move(const double* p, int deg, const Cx* zs, RootState r, int k) {
    if (r.done) return r;
    Cx pv, dv; double err;
    aberth_eval(p, deg, r.z, pv, dv, err);
    if (cabs1(pv) <= err * 4e-16) { r.done = true; return r; }  // on a root to the rounding of the evaluation
    Cx w;
    if (dv.re == 0.0 && dv.im == 0.0) w = { 1e-3 * (1.0 + cabs1(r.z)), 1e-3 };  // a critical point that is no root: step aside
    else w = cdiv(pv, dv);
    Cx sum = { 0.0, 0.0 };
#pragma unroll
    for (int j = 0; j < 10; j++) {
        if (j >= deg || j == k) continue;
        const Cx d = csub(r.z, zs[j]);
        if (d.re == 0.0 && d.im == 0.0) continue;  // two iterates on one point: no repulsion between them this sweep
        sum = cadd(sum, cinv(d));
    }
    const Cx den = csub(Cx{ 1.0, 0.0 }, cmul(w, sum));
    const Cx step = (den.re == 0.0 && den.im == 0.0) ? w : cdiv(w, den);
    const Cx z1 = csub(r.z, step);
    if (!(z1.re == z1.re) || !(z1.im == z1.im)) { r.done = true; return r; }  // (overflow: keep the last finite value; the polish below rejects what is no root)
    if (z1.re == r.z.re && z1.im == r.z.im) r.done = true;
    r.z = z1;
    return r;
}
// the polynomial, normalised and with vanishing leading coefficients trimmed (a root at infinity is dropped); returns the degree (< 1: nothing to solve)
VK_HD inline int poly_trim(const double* p, int deg, double* c) {
    double scale = 0.0;
    for (int i = 0; i <= deg; i++) scale = vk_abs(p[i]) > scale ? vk_abs(p[i]) : scale;
    if (!(scale > 0.0)) return 0;
    while (deg > 0 && vk_abs(p[deg]) <= 1e-14 * scale) deg--;
    for (int i = 0; i <= deg; i++) c[i] = p[i] / scale;
    return deg;
}
// From the complex roots to the real starting values z of the solutions, in root order.  A real root, or one member of a near-double real pair:
// where two real roots nearly coincide (forward motion, small baselines) the polynomial's rounding turns them into a complex pair re +- i im with a
// small imaginary part.  Such a pair is tried as the two real starts re -+ |im| (the conjugate produces the same two: the duplicate test drops
// them); a start that belongs to no real root ends up as a model the polish on the ten cubics rejects.  (Round 4, found with the independent solver
// oracle/orc_fivept.py: with the strict 1e-7 test alone the planted motion was missing in a quarter of the forward-motion samples.)
VK_HD inline int real_candidates(const double* c, int deg, const Cx* roots, double* zs) {
    int n = 0;
    for (int m = 0; m < deg; m++) {
        const Cx x = roots[m];
        const double tol_im = vk_abs(x.im) / (1.0 + vk_abs(x.re));
        if (!(tol_im <= 1e-3)) continue;
        const int n_start = tol_im <= 1e-7 ? 1 : 2;
        for (int st = 0; st < n_start; st++) {
            double z = n_start == 1 ? x.re : x.re + (st == 0 ? -vk_abs(x.im) : vk_abs(x.im));
            for (int it = 0; it < 3; it++) {  // real Newton steps on the real polynomial
                double v = c[deg], dv = 0.0;
                for (int i = deg - 1; i >= 0; i--) { dv = dv * z + v; v = v * z + c[i]; }
                if (dv != 0.0) z -= v / dv;
            }
            bool dup = false;
            for (int k = 0; k < n; k++) dup = dup || vk_abs(zs[k] - z) <= 1e-9 * (1.0 + vk_abs(z));
            if (!dup && n < 20 && z == z) zs[n++] = z;
        }
    }
    return n;
}
// real starting values of the real polynomial p[0..deg] by the host loop over the Aberth sweeps (the device runs the same sweeps one root per lane)
VK_HD inline int real_roots(const double* p, int deg_in, double* zs) {
    double c[11];
    const int deg = poly_trim(p, deg_in, c);
    if (deg < 1) return 0;
    Cx z[10], zn[10]; bool done[10], dn[10];
    for (int k = 0; k < deg; k++) { z[k] = aberth_start(k); done[k] = false; }
    for (int sw = 0; sw < ABERTH_SWEEPS; sw++) {
        bool all = true;
        for (int k = 0; k < deg; k++) { const RootState r = aberth_move(c, deg, z, RootState{ z[k], done[k] }, k); zn[k] = r.z; dn[k] = r.done; all = all && r.done; }
        for (int k = 0; k < deg; k++) { z[k] = zn[k]; done[k] = dn[k]; }
        if (all) break;
    }
    return real_candidates(c, deg, z, zs);
}

// ---- the phases of one basis of the null space ---------------------------------------------------------------------------------------------
// (1) everything up to the degree-10 polynomial: the ten cubics (A0), their elimination, B(z), det B(z) = n.  false: singular sample.
struct BasisWork { double A0[10][20]; double B[3][3][5]; double n[11]; };
// B(z) and its determinant from the eliminated constraints A
VK_HD inline void basis_poly(const double (*A)[20], BasisWork& W) {
    // B(z): rows k = e - z f, l = g - z h, m = i - z j; columns: coefficient of x (degree 3), of y (degree 3), constant (degree 4)
    for (int r = 0; r < 3; r++) {
        const double* e = A[4 + 2 * r]; const double* f = A[5 + 2 * r];
        double* bx = W.B[r][0]; double* by = W.B[r][1]; double* b1 = W.B[r][2];
        bx[0] = e[12]; bx[1] = e[11] - f[12]; bx[2] = e[10] - f[11]; bx[3] = -f[10]; bx[4] = 0.0;
        by[0] = e[15]; by[1] = e[14] - f[15]; by[2] = e[13] - f[14]; by[3] = -f[13]; by[4] = 0.0;
        b1[0] = e[19]; b1[1] = e[18] - f[19]; b1[2] = e[17] - f[18]; b1[3] = e[16] - f[17]; b1[4] = -f[16];
    }
    double B[3][3][5], p1[8], p2[8], p3[8], n[11];  // (local copies, constant trip counts: registers on the device)
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int i = 0; i < 5; i++) B[r][c][i] = W.B[r][c][i];
#pragma unroll
    for (int i = 0; i < 8; i++) { p1[i] = 0.0; p2[i] = 0.0; p3[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < 11; i++) n[i] = 0.0;
    pmul_acc(B[0][1], 3, B[1][2], 4, 1.0, p1); pmul_acc(B[0][2], 4, B[1][1], 3, -1.0, p1);  // b12 b23 - b13 b22
    pmul_acc(B[0][2], 4, B[1][0], 3, 1.0, p2); pmul_acc(B[0][0], 3, B[1][2], 4, -1.0, p2);  // b13 b21 - b11 b23
    pmul_acc(B[0][0], 3, B[1][1], 3, 1.0, p3); pmul_acc(B[0][1], 3, B[1][0], 3, -1.0, p3);  // b11 b22 - b12 b21
    pmul_acc(p1, 7, B[2][0], 3, 1.0, n); pmul_acc(p2, 7, B[2][1], 3, 1.0, n); pmul_acc(p3, 6, B[2][2], 4, 1.0, n);
#pragma unroll
    for (int i = 0; i < 11; i++) W.n[i] = n[i];
}
VK_HD inline bool prepare_basis(const double (*N)[9], BasisWork& W, double (*A)[20] /* scratch [10][20] */) {
    constraint_matrix_rows(N, W.A0);
    for (int r = 0; r < 10; r++) for (int c = 0; c < 20; c++) A[r][c] = W.A0[r][c];
    if (!gauss_jordan_10x20(A)) return false;
    basis_poly(A, W);
    return true;
}
// (3) one starting value z -> the essential matrix of the solution it belongs to, or false
VK_HD inline bool polish_candidate(const double (*N)[9], const BasisWork& W, double z, double* Eout) {
    // (x, y, 1) spans the null space of B(z): the cross product of the best-conditioned pair of its rows (with rows 0 and 1 this is
    // x = p1 / p3, y = p2 / p3 of the paper; a fixed pair loses digits where its 2 x 2 minor is small)
    double Bz[3][3];
    for (int r = 0; r < 3; r++) { Bz[r][0] = peval(W.B[r][0], 3, z); Bz[r][1] = peval(W.B[r][1], 3, z); Bz[r][2] = peval(W.B[r][2], 4, z); }
    double best[3] = { 0, 0, 0 }, bestn = -1.0;
    for (int a = 0; a < 3; a++)
        for (int b = a + 1; b < 3; b++) {
            const double c[3] = { Bz[a][1] * Bz[b][2] - Bz[a][2] * Bz[b][1], Bz[a][2] * Bz[b][0] - Bz[a][0] * Bz[b][2], Bz[a][0] * Bz[b][1] - Bz[a][1] * Bz[b][0] };
            const double na = Bz[a][0] * Bz[a][0] + Bz[a][1] * Bz[a][1] + Bz[a][2] * Bz[a][2], nb = Bz[b][0] * Bz[b][0] + Bz[b][1] * Bz[b][1] + Bz[b][2] * Bz[b][2];
            const double nn = (c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) / (na * nb > 0.0 ? na * nb : 1.0);  // sin^2 of the angle between the rows
            if (nn > bestn) { bestn = nn; best[0] = c[0]; best[1] = c[1]; best[2] = c[2]; }
        }
    if (!(vk_abs(best[2]) > 1e-300)) return false;
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
            for (int c = 0; c < 20; c++) { res += W.A0[r][c] * mo[c]; j0 += W.A0[r][c] * dx[c]; j1 += W.A0[r][c] * dy[c]; j2 += W.A0[r][c] * dz[c]; }
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
            for (int c = 0; c < 20; c++) { res += W.A0[r][c] * mo[c]; mag += vk_abs(W.A0[r][c] * mo[c]); }
            const double rel = mag > 0.0 ? vk_abs(res) / mag : 0.0;
            worst = rel > worst ? rel : worst;
        }
        if (!(worst <= 1e-9)) return false;
    }
    double E[9], nn = 0.0;
    for (int c = 0; c < 9; c++) { E[c] = x * N[0][c] + y * N[1][c] + zz * N[2][c] + N[3][c]; nn += E[c] * E[c]; }
    if (!(nn > 0.0) || !(nn < 1e300)) return false;
    const double s = 1.4142135623730951 / vk_sqrt(nn);
    for (int c = 0; c < 9; c++) Eout[c] = E[c] * s;
    return true;
}
// two models are one (up to sign, to 1e-6 per entry)
VK_HD inline bool same_model(const double* a, const double* b) {
    double dp = 0.0, dm = 0.0;
    for (int c = 0; c < 9; c++) { const double u = vk_abs(a[c] - b[c]), v = vk_abs(a[c] + b[c]); dp = u > dp ? u : dp; dm = v > dm ? v : dm; }
    return (dp < dm ? dp : dm) <= 1e-6;
}
// the second basis: an orthogonal mix of the four vectors (a product of two plane rotations by 45 and ~59 degrees with a row swap: every new
// vector has a share of every old one)
VK_HD inline void second_basis(const double (*N)[9], double (*M)[9]) {
    const double c1 = 0.7071067811865476, s1 = 0.7071067811865476, c2 = 0.5144957554275265, s2 = 0.8574929257125441;
    double T[4][9];
    for (int c = 0; c < 9; c++) {
        T[0][c] = c1 * N[0][c] + s1 * N[3][c]; T[3][c] = -s1 * N[0][c] + c1 * N[3][c];
        T[1][c] = c1 * N[1][c] + s1 * N[2][c]; T[2][c] = -s1 * N[1][c] + c1 * N[2][c];
    }
    for (int c = 0; c < 9; c++) {
        M[0][c] = c2 * T[0][c] + s2 * T[1][c]; M[1][c] = -s2 * T[0][c] + c2 * T[1][c];
        M[2][c] = c2 * T[2][c] + s2 * T[3][c]; M[3][c] = -s2 * T[2][c] + c2 * T[3][c];
    }
}
// the 5 x 9 epipolar system of five normalised correspondences
VK_HD inline void epipolar_rows(const double (*q1)[2], const double (*q2)[2], double (*Q)[9]) {
    for (int k = 0; k < 5; k++) {
        const double a[9] = { q2[k][0] * q1[k][0], q2[k][0] * q1[k][1], q2[k][0], q2[k][1] * q1[k][0], q2[k][1] * q1[k][1], q2[k][1], q1[k][0], q1[k][1], 1.0 };
        for (int c = 0; c < 9; c++) Q[k][c] = a[c];
    }
}

// the solutions E = x N0 + y N1 + z N2 + N3 for ONE basis N of the null space, in the order of their starting values
VK_HD inline int solve_basis(const double (*N)[9], double (*Es)[9], double* dbg_poly = nullptr, double* dbg_roots = nullptr, int* dbg_nroots = nullptr) {
    BasisWork W;
    double A[10][20];
    if (!prepare_basis(N, W, A)) return 0;
    double zs[20];
    const int nz = real_roots(W.n, 10, zs);
    if (dbg_poly) for (int i = 0; i < 11; i++) dbg_poly[i] = W.n[i];
    if (dbg_roots) for (int i = 0; i < nz && i < 10; i++) dbg_roots[i] = zs[i];
    if (dbg_nroots) *dbg_nroots = nz < 10 ? nz : 10;
    int ne = 0;
    for (int k = 0; k < nz && ne < 10; k++)
        if (polish_candidate(N, W, zs[k], Es[ne])) ne++;
    return ne;
}

// q1, q2: five normalised correspondences (image 1, image 2), [5][2].  Es: up to ten essential matrices (row-major, Frobenius norm sqrt 2).
// The basis of the four-dimensional null space is arbitrary, the solutions are not: where the roots of the degree-10 polynomial in z cluster (pure
// forward motion, small baselines: near-multiple roots get lost) they are spread out in another basis.  The system is therefore solved in TWO bases --
// the one the elimination produced and a fixed rotation of it -- and the union is returned (round 4; measured against the independent solver
// oracle/orc_fivept.py, tests/test_fivept.py).
VK_HD inline int merge_models(double (*Es)[9], int ne, const double (*E2)[9], int n2) {
    for (int k = 0; k < n2 && ne < 10; k++) {
        bool dup = false;
        for (int j = 0; j < ne && !dup; j++) dup = same_model(E2[k], Es[j]);
        if (!dup) { for (int c = 0; c < 9; c++) Es[ne][c] = E2[k][c]; ne++; }
    }
    return ne;
}
VK_HD inline int solve(const double (*q1)[2], const double (*q2)[2], double (*Es)[9], double* dbg_poly = nullptr, double* dbg_roots = nullptr, int* dbg_nroots = nullptr) {
    double Q[5][9], N[4][9], M[4][9], E2[10][9];
    epipolar_rows(q1, q2, Q);
    if (!null_space_5x9(Q, N)) return 0;
    int ne = solve_basis(N, Es, dbg_poly, dbg_roots, dbg_nroots);
    second_basis(N, M);
    const int n2 = solve_basis(M, E2);
    return merge_models(Es, ne, E2, n2);
}

}  // namespace fivept
}  // namespace vk
