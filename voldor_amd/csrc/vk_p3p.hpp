// voldor_amd/csrc/vk_p3p.hpp -- per-lane minimal pose solvers, everything in registers.
//   * LambdaTwist P3P (Persson & Nordberg, ECCV'18) + 4th-point disambiguation: what the
//     reference runs through lambdatwist/lambdatwist_p4p.h:5-62 / lambdatwist_p3p.h:19-294.
//   * AP3P (Ke & Roumeliotis, CVPR'17) as in gpu-kernels/solve_batch_ap3p.cu:28-378.
//   * rotation matrix -> nearest rotation -> angle-axis (gpu-kernels/rodrigues.h:5-114).
// Written for one pose hypothesis per lane of a wave64; no local arrays are indexed
// dynamically, so nothing spills to scratch.  `S` is the solver scalar (float = the
// reference's GPU instantiation).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include "vk_strict_math.h"
// the solvers are plain arithmetic: compiled for host and device so the host build can be
// checked against the oracle without a GPU (vk_host_* entry points, tests/test_host_math.py)
#define VK_HD __host__ __device__

namespace vk {

// Overload-by-argument-type math, like the std:: overloads the reference relies on: a double
// argument (e.g. `b * b - 3.0 * c`) takes the double routine, a float one the float routine.
VK_HD __forceinline__ float vk_sqrt(float x) { return sqrtf(x); }
VK_HD __forceinline__ double vk_sqrt(double x) { return ::sqrt(x); }
VK_HD __forceinline__ float vk_abs(float x) { return fabsf(x); }
VK_HD __forceinline__ double vk_abs(double x) { return ::fabs(x); }

template <typename S> struct V3 { S x, y, z; };
template <typename S> VK_HD __forceinline__ S dot(V3<S> a, V3<S> b) {
#pragma clang fp contract(off)
    S s = S(0);  // matrix.h:846-851 accumulates from zero
    s += a.x * b.x; s += a.y * b.y; s += a.z * b.z;
    return s;
}
template <typename S> VK_HD __forceinline__ V3<S> cross(V3<S> a, V3<S> b) {
#pragma clang fp contract(off)
    return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
template <typename S> VK_HD __forceinline__ V3<S> sub(V3<S> a, V3<S> b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
template <typename S> VK_HD __forceinline__ V3<S> scale(V3<S> a, S s) { return { a.x * s, a.y * s, a.z * s }; }
template <typename S> VK_HD __forceinline__ V3<S> normalized(V3<S> a) { return scale(a, S(1.0) / vk_sqrt(dot(a, a))); }
template <typename S> struct M3 { S m[9]; };  // row-major

// NUMERICS NOTE.  The reference instantiates its solver with _T=float but keeps double
// literals (2.0, 0.5, 4.0 ...) in many expressions, so parts of every hypothesis are evaluated in
// double and rounded back (lambdatwist_p3p.h:38-137, solve_cubic.h:17-34,160-207, solve_eig0.h:
// 40-67, refine_lambda.h:31-57).  P3P on noisy flow is ill-conditioned for a few percent of the
// 4-tuples, and those samples move the mean-shift mode by ~1e-3 if they are evaluated with a
// different rounding sequence.  To stay inside the 1e-3 pose tolerance against the reference's
// float path, the literals below are written with the reference's types (a bare `2.0` is a
// double on purpose, `S(2.0)` is the solver scalar) and fma contraction is disabled.

// x^2 + b x + c = 0, numerically stable pair (solve_cubic.h:13-35)
template <typename S> VK_HD __forceinline__ bool root2real(S b, S c, S& r1, S& r2) {
#pragma clang fp contract(off)
    S v = b * b - 4.0 * c;
    if (v < 0) { r1 = r2 = 0.5 * b; return false; }
    S y = vk_sqrt(v);
    if (b < 0) { r1 = 0.5 * (-b + y); r2 = 0.5 * (-b - y); }
    else { r1 = 2.0 * c / (-b + y); r2 = 2.0 * c / (-b - y); }
    return true;
}

// One real root of x^3 + b x^2 + c x + d with the steepest derivative (solve_cubic.h:154-210)
// The reference runs up to 50 Newton steps and leaves early only on |f| <= 1e-7 (float), a threshold most cubics never meet in float
// arithmetic although the iterate has long stopped moving (measured on 640x480 pools: 89 % of the hypotheses run all 50 steps; 70 % sit on
// a fixed point, 17 % on a two-cycle of neighbouring floats).  Every mode runs the reference's 50: a cap of 12 steps was measured in round 4
// (k_solve 17.6 -> 14.4 us) and fails the ensemble test -- the ~1.4 % of cubics still wandering near a double root re-draw the pool.
template <typename S> VK_HD __forceinline__ S cubic_root(S b, S c, S d) {
#pragma clang fp contract(off)
    S r0;
    if (b * b >= 3.0 * c) {
        S v = vk_sqrt(b * b - 3.0 * c);
        S t1 = (-b - v) / (3.0);
        S k = ((t1 + b) * t1 + c) * t1 + d;
        if (k > 0.0) r0 = t1 - vk_sqrt(-k / (3.0 * t1 + b));
        else {
            S t2 = (-b + v) / (3.0);
            k = ((t2 + b) * t2 + c) * t2 + d;
            r0 = t2 + vk_sqrt(-k / (3.0 * t2 + b));
        }
    } else {
        r0 = -b / 3.0;
        if (vk_abs(((S(3.0) * r0 + S(2.0) * b) * r0 + c)) < 1e-4) r0 += 1;
    }
    const S lim = sizeof(S) == 4 ? S(1e-7) : S(1e-13);  // get_numeric_limit<T>() (solve_cubic.h:87-108)
#pragma unroll 1
    for (int cnt = 0; cnt < 50; ++cnt) {
        S fx = (((r0 + b) * r0 + c) * r0 + d);
        if (cnt < 7 || vk_abs(fx) > lim) {
            S fpx = ((S(3.0) * r0 + S(2.0) * b) * r0 + c);
            r0 -= fx / fpx;
        } else
            break;
    }
    return r0;
}

// Eigen-decomposition of a symmetric 3x3 with one known zero eigenvalue (solve_eig0.h:11-80).
// Only the first two columns of the eigenvector matrix are needed by the caller.
template <typename S>
VK_HD __forceinline__ void eig_known0(const S* x, S& e1, S& e2, V3<S>& v1, V3<S>& v2) {
#pragma clang fp contract(off)
    S x01s = x[1] * x[1];
    S b = -x[0] - x[4] - x[8];
    S c = -x01s - x[2] * x[2] - x[5] * x[5] + x[0] * (x[4] + x[8]) + x[4] * x[8];
    root2real(b, c, e1, e2);
    if (vk_abs(e1) < vk_abs(e2)) { S t = e1; e1 = e2; e2 = t; }
    S mx0011 = -x[0] * x[4];
    S prec0 = x[1] * x[5] - x[2] * x[4];
    S prec1 = x[1] * x[2] - x[0] * x[5];
    {
        S tmp = 1.0 / (e1 * (x[0] + x[4]) + mx0011 - e1 * e1 + x01s);
        S a1 = -(e1 * x[2] + prec0) * tmp, a2 = -(e1 * x[5] + prec1) * tmp;
        S rn = ((S)1.0) / vk_sqrt(a1 * a1 + a2 * a2 + 1.0);
        a1 *= rn; a2 *= rn;
        v1 = { a1, a2, rn };
    }
    {
        S tmp = 1.0 / (e2 * (x[0] + x[4]) + mx0011 - e2 * e2 + x01s);
        S a1 = -(e2 * x[2] + prec0) * tmp, a2 = -(e2 * x[5] + prec1) * tmp;
        S rn = 1.0 / vk_sqrt(a1 * a1 + a2 * a2 + 1.0);
        a1 *= rn; a2 *= rn;
        v2 = { a1, a2, rn };
    }
}

// 5 Gauss-Newton steps on the three law-of-cosines residuals (refine_lambda.h:5-102)
template <typename S>
VK_HD __forceinline__ void refine_lambda(V3<S>& L, S a12, S a13, S a23, S b12, S b13, S b23) {
#pragma clang fp contract(off)
#pragma unroll 1
    for (int i = 0; i < 5; ++i) {
        S l1 = L.x, l2 = L.y, l3 = L.z;
        S r1 = l1 * l1 + l2 * l2 + b12 * l1 * l2 - a12;
        S r2 = l1 * l1 + l3 * l3 + b13 * l1 * l3 - a13;
        S r3 = l2 * l2 + l3 * l3 + b23 * l2 * l3 - a23;
        if (vk_abs(r1) + vk_abs(r2) + vk_abs(r3) < 1e-10) break;
        S v0 = (2.0) * l1 + b12 * l2, v1 = (2.0) * l2 + b12 * l1;
        S v3 = (2.0) * l1 + b13 * l3, v5 = (2.0) * l3 + b13 * l1;
        S v7 = (2.0) * l2 + b23 * l3, v8 = (2.0) * l3 + b23 * l2;
        S det = (1.0) / (-v0 * v5 * v7 - v1 * v3 * v8);
        // L1 = L - det * (Ji * r), the product accumulated from zero as matrix.h:795-810 does
        S s0 = S(0), s1 = S(0), s2 = S(0);
        s0 += (-v5 * v7) * r1; s0 += (-v1 * v8) * r2; s0 += (v1 * v5) * r3;
        s1 += (-v3 * v8) * r1; s1 += (v0 * v8) * r2; s1 += (-v0 * v5) * r3;
        s2 += (v3 * v7) * r1; s2 += (-v0 * v7) * r2; s2 += (-v1 * v3) * r3;
        V3<S> n = { l1 - s0 * det, l2 - s1 * det, l3 - s2 * det };
        S q1 = n.x * n.x + n.y * n.y + b12 * n.x * n.y - a12;
        S q2 = n.x * n.x + n.z * n.z + b13 * n.x * n.z - a13;
        S q3 = n.y * n.y + n.z * n.z + b23 * n.y * n.z - a23;
        if (vk_abs(q1) + vk_abs(q2) + vk_abs(q3) > vk_abs(r1) + vk_abs(r2) + vk_abs(r3)) break;
        L = n;
    }
}

// Candidate bookkeeping without dynamically indexed arrays: the best (min 4th-point
// reprojection) pose is tracked while candidates are generated.
template <typename S> struct BestPose {
    S R[9]; S t[3]; S err; int n;
};

template <typename S>
VK_HD __forceinline__ void consider(BestPose<S>& B, V3<S> L, S a12, S a13, S a23, S b12, S b13, S b23,
                                         V3<S> y1, V3<S> y2, V3<S> y3, V3<S> x1, const S* Xi, const float* x4, float y4u, float y4v,
                                         float fx, float fy, float cx, float cy) {
#pragma clang fp contract(off)
    refine_lambda(L, a12, a13, a23, b12, b13, b23);
    V3<S> ry1 = scale(y1, L.x), ry2 = scale(y2, L.y), ry3 = scale(y3, L.z);
    V3<S> yd1 = sub(ry1, ry2), yd2 = sub(ry1, ry3), yc = cross(yd1, yd2);
    // R = Y * X^-1, Y = [yd1 yd2 yc] (columns); sums start from zero like matrix.h:795-810
    const S Y[9] = { yd1.x, yd2.x, yc.x, yd1.y, yd2.y, yc.y, yd1.z, yd2.z, yc.z };
    S R[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            S sum = S(0);
            sum += Y[r * 3] * Xi[c]; sum += Y[r * 3 + 1] * Xi[3 + c]; sum += Y[r * 3 + 2] * Xi[6 + c];
            R[r * 3 + c] = sum;
        }
    S t[3];
    const S ry[3] = { ry1.x, ry1.y, ry1.z };
#pragma unroll
    for (int r = 0; r < 3; r++) {
        S sum = S(0);
        sum += R[r * 3] * x1.x; sum += R[r * 3 + 1] * x1.y; sum += R[r * 3 + 2] * x1.z;
        t[r] = ry[r] - sum;
    }
    // 4th-point reprojection (lambdatwist_p4p.h:31-41): _T * float products, float intrinsics
    S X = R[0] * x4[0] + R[1] * x4[1] + R[2] * x4[2] + t[0];
    S Yp = R[3] * x4[0] + R[4] * x4[1] + R[5] * x4[2] + t[1];
    S Z = R[6] * x4[0] + R[7] * x4[1] + R[8] * x4[2] + t[2];
    S mu = cx + fx * X / Z, mv = cy + fy * Yp / Z;
    S err = (mu - y4u) * (mu - y4u) + (mv - y4v) * (mv - y4v);
    // first candidate always taken, later ones only if strictly better
    if (B.n == 0 || B.err > err) {
#pragma unroll
        for (int k = 0; k < 9; k++) B.R[k] = R[k];
        B.t[0] = t[0]; B.t[1] = t[1]; B.t[2] = t[2]; B.err = err;
    }
    B.n++;
}

// y: 4 pixel observations (u,v), x: 4 3-D points. Returns false if P3P has no solution.
// `only` >= 0 evaluates just candidate root (block only>>1, root only&1) and reports its 4th-point error through
// `err_out` (k_solve spreads the up-to-four candidates of one hypothesis over four lanes and folds them afterwards
// with the same "first one, then strictly better" rule); only < 0 walks all candidates like the reference.
template <typename S>
VK_HD static bool lambdatwist_p4p(const float* yu, const float* yv, const float (*xp)[3], float fxf, float fyf,
                                       float cxf, float cyf, float* Rout, float* tout, int only = -1, S* err_out = nullptr,
                                       S* dbg = nullptr /* tests only: intermediate values of the sequential path (tests/cxx/vk_testhooks.hip) */) {
#pragma clang fp contract(off)
    // bearings are formed in float and then widened (lambdatwist_p4p.h:13-15)
    V3<S> y1 = normalized<S>({ (S)((yu[0] - cxf) / fxf), (S)((yv[0] - cyf) / fyf), S(1.0) });
    V3<S> y2 = normalized<S>({ (S)((yu[1] - cxf) / fxf), (S)((yv[1] - cyf) / fyf), S(1.0) });
    V3<S> y3 = normalized<S>({ (S)((yu[2] - cxf) / fxf), (S)((yv[2] - cyf) / fyf), S(1.0) });
    V3<S> x1 = { (S)xp[0][0], (S)xp[0][1], (S)xp[0][2] }, x2 = { (S)xp[1][0], (S)xp[1][1], (S)xp[1][2] };
    V3<S> x3 = { (S)xp[2][0], (S)xp[2][1], (S)xp[2][2] };
    S b12 = -2.0 * (dot(y1, y2)), b13 = -2.0 * (dot(y1, y3)), b23 = -2.0 * (dot(y2, y3));
    V3<S> d12 = sub(x1, x2), d13 = sub(x1, x3), d23 = sub(x2, x3), dc = cross(d12, d13);
    S a12 = dot(d12, d12), a13 = dot(d13, d13), a23 = dot(d23, d23);
    S c31 = -0.5 * b13, c23 = -0.5 * b23, c12 = -0.5 * b12;
    S blob = (c12 * c23 * c31 - 1.0);
    S s31 = 1.0 - c31 * c31, s23 = 1.0 - c23 * c23, s12 = 1.0 - c12 * c12;
    S p3 = (a13 * (a23 * s31 - a13 * s23));
    S p2 = 2.0 * blob * a23 * a13 + a13 * (2.0 * a12 + a13) * s23 + a23 * (a23 - a12) * s31;
    S p1 = a23 * (a13 - a23) * s12 - a12 * a12 * s23 - 2.0 * a12 * (blob * a23 + a13 * s23);
    S p0 = a12 * (a12 * s23 - a23 * s12);
    p3 = 1.0 / p3;
    p2 *= p3; p1 *= p3; p0 *= p3;
    S g = cubic_root<S>(p2, p1, p0);

    S A[9];
    A[0] = a23 * (1.0 - g); A[1] = (a23 * b12) * 0.5; A[2] = (a23 * b13 * g) * (-0.5);
    A[4] = a23 - a12 + a13 * g; A[5] = b23 * (a13 * g - a12) * 0.5; A[8] = g * (a13 - a23) - a12;
    A[3] = A[1]; A[6] = A[2]; A[7] = A[5];
    S e1, e2; V3<S> v1, v2;
    eig_known0<S>(A, e1, e2, v1, v2);
    S v = vk_sqrt(-e2 / e1 > 0 ? -e2 / e1 : S(0));
    if (dbg) {
        dbg[0] = p2; dbg[1] = p1; dbg[2] = p0; dbg[3] = g; dbg[4] = e1; dbg[5] = e2; dbg[6] = v1.x; dbg[7] = v1.y; dbg[8] = v1.z;
        dbg[9] = v2.x; dbg[10] = v2.y; dbg[11] = v2.z; dbg[12] = v; dbg[13] = a12; dbg[14] = a13; dbg[15] = a23; dbg[16] = b12; dbg[17] = b13;
        dbg[18] = b23; dbg[19] = y1.x; dbg[20] = y1.y; dbg[21] = y1.z;
    }

    // X^-1 with X = [d12 d13 d12xd13] columns (matrix.h:636-656 adjugate form)
    S Xm[9] = { d12.x, d13.x, dc.x, d12.y, d13.y, dc.y, d12.z, d13.z, dc.z };
    S Xi[9];
    {
        S M0 = Xm[4] * Xm[8] - Xm[5] * Xm[7], M1 = Xm[2] * Xm[7] - Xm[1] * Xm[8], M2 = Xm[1] * Xm[5] - Xm[2] * Xm[4];
        S M3_ = Xm[5] * Xm[6] - Xm[3] * Xm[8], M4 = Xm[0] * Xm[8] - Xm[2] * Xm[6], M5 = Xm[2] * Xm[3] - Xm[0] * Xm[5];
        S M6 = Xm[3] * Xm[7] - Xm[4] * Xm[6], M7 = Xm[1] * Xm[6] - Xm[0] * Xm[7], M8 = Xm[0] * Xm[4] - Xm[1] * Xm[3];
        S idet = S(1.0) / (Xm[0] * M0 + Xm[1] * M3_ + Xm[2] * M6);
        Xi[0] = M0 * idet; Xi[1] = M1 * idet; Xi[2] = M2 * idet; Xi[3] = M3_ * idet; Xi[4] = M4 * idet;
        Xi[5] = M5 * idet; Xi[6] = M6 * idet; Xi[7] = M7 * idet; Xi[8] = M8 * idet;
    }
    BestPose<S> B;
    B.n = 0; B.err = S(0);
    if (only >= 0) {  // one candidate, straight-line: the lanes of a hypothesis differ in data only
        const S s = (only >> 1) == 0 ? v : -v;
        S w2 = S(1.0) / (s * v2.x - v1.x);
        S w0 = (v1.y - s * v2.y) * w2;
        S w1 = (v1.z - s * v2.z) * w2;
        S a = S(1.0) / ((a13 - a12) * w1 * w1 - a12 * b13 * w1 - a12);
        S b = (a13 * b12 * w1 - a12 * b13 * w0 - S(2.0) * w0 * w1 * (a12 - a13)) * a;
        S c = ((a13 - a12) * w0 * w0 + a13 * b12 * w0 + a13) * a;
        if (b * b - 4.0 * c >= 0) {
            S tau1, tau2;
            root2real(b, c, tau1, tau2);
            const S tau = (only & 1) == 0 ? tau1 : tau2;
            if (tau > 0) {
                S d = a23 / (tau * (b23 + tau) + S(1.0));
                if (d > 0) {
                    S l2 = vk_sqrt(d), l3 = tau * l2, l1 = w0 * l2 + w1 * l3;
                    if (l1 >= 0)
                        consider<S>(B, { l1, l2, l3 }, a12, a13, a23, b12, b13, b23, y1, y2, y3, x1, Xi, xp[3], yu[3], yv[3],
                                    fxf, fyf, cxf, cyf);
                }
            }
        }
        if (err_out) *err_out = B.err;
    } else
#pragma unroll 1
    for (int blk = 0; blk < 2; blk++) {  // s = +v, then s = -v (lambdatwist_p3p.h:140-240)
        S s = blk == 0 ? v : -v;
        S w2 = S(1.0) / (s * v2.x - v1.x);
        S w0 = (v1.y - s * v2.y) * w2;
        S w1 = (v1.z - s * v2.z) * w2;
        S a = S(1.0) / ((a13 - a12) * w1 * w1 - a12 * b13 * w1 - a12);
        S b = (a13 * b12 * w1 - a12 * b13 * w0 - S(2.0) * w0 * w1 * (a12 - a13)) * a;
        S c = ((a13 - a12) * w0 * w0 + a13 * b12 * w0 + a13) * a;
        if (b * b - 4.0 * c >= 0) {
            S tau1, tau2;
            root2real(b, c, tau1, tau2);
#pragma unroll 1
            for (int k = 0; k < 2; k++) {
                S tau = k == 0 ? tau1 : tau2;
                if (tau > 0) {
                    S d = a23 / (tau * (b23 + tau) + S(1.0));
                    if (d > 0) {  // the +v block relies on vk_sqrt(NaN) failing l1>=0: same outcome
                        S l2 = vk_sqrt(d), l3 = tau * l2, l1 = w0 * l2 + w1 * l3;
                        if (dbg) { S* q = dbg + 32 + (blk * 2 + k) * 16; q[0] = w0; q[1] = w1; q[2] = a; q[3] = b; q[4] = c; q[5] = tau; q[6] = d; q[7] = l1; q[8] = l2; q[9] = l3; }
                        if (l1 >= 0) {
                            consider<S>(B, { l1, l2, l3 }, a12, a13, a23, b12, b13, b23, y1, y2, y3, x1, Xi, xp[3], yu[3], yv[3],
                                        fxf, fyf, cxf, cyf);
                            if (dbg) { S* q = dbg + 32 + (blk * 2 + k) * 16; q[10] = B.err; q[11] = B.t[0]; q[12] = B.t[1]; q[13] = B.t[2]; q[14] = (S)B.n; }
                        }
                    }
                }
            }
        }
    }
    if (dbg) { for (int i = 0; i < 9; i++) dbg[22 + i] = Xi[i]; }
    if (B.n == 0) return false;
#pragma unroll
    for (int k = 0; k < 9; k++) Rout[k] = (float)B.R[k];
    tout[0] = (float)B.t[0]; tout[1] = (float)B.t[1]; tout[2] = (float)B.t[2];
    return true;
}

// ---- rotation matrix -> angle-axis -----------------------------------------------------------
// Nearest rotation by Newton iteration on the polar factor, X <- (X + X^-T)/2 (the reference
// gets the same factor as U*V^T from an approximate SVD, rodrigues.h:82-108), then Ceres'
// atan2 formula (rodrigues.h:5-79).
VK_HD __forceinline__ void nearest_rotation(float* Rf) {
#pragma clang fp contract(off)
    // fp64: P3P on degenerate 4-tuples returns near-singular "rotations" whose polar factor is
    // ill-conditioned; evaluating it in double keeps those hypotheses reproducible (8192 lanes
    // x ~10 iterations of 3x3 algebra is negligible next to the per-pixel passes).
    double X[9];
#pragma unroll
    for (int i = 0; i < 9; i++) X[i] = (double)Rf[i];
#pragma unroll 1
    for (int it = 0; it < 30; it++) {
        double c[9];
        c[0] = X[4] * X[8] - X[5] * X[7]; c[1] = X[5] * X[6] - X[3] * X[8]; c[2] = X[3] * X[7] - X[4] * X[6];
        c[3] = X[2] * X[7] - X[1] * X[8]; c[4] = X[0] * X[8] - X[2] * X[6]; c[5] = X[1] * X[6] - X[0] * X[7];
        c[6] = X[1] * X[5] - X[2] * X[4]; c[7] = X[2] * X[3] - X[0] * X[5]; c[8] = X[0] * X[4] - X[1] * X[3];
        double det = X[0] * c[0] + X[1] * c[1] + X[2] * c[2];
        if (!(vk_abs(det) > 1e-300)) break;
        const double idet = 1.0 / det;
        double delta = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) { double y = 0.5 * (X[i] + c[i] * idet); delta += vk_abs(y - X[i]); X[i] = y; }
        if (delta < 1e-10) break;  // quadratic convergence: the next step would move X by ~delta^2
    }
#pragma unroll
    for (int i = 0; i < 9; i++) Rf[i] = (float)X[i];
}
// `strict`: software atan2 / sin / cos (vk_strict_math.h) -- the same bits on the host and on the device (strict-math mode)
VK_HD __forceinline__ void rotmat_to_angle_axis(const float* R, float* aa, bool strict = false) {
    float a0 = R[7] - R[5], a1 = R[2] - R[6], a2 = R[3] - R[1];
    float costheta = fminf(fmaxf((R[0] + R[4] + R[8] - 1.f) * 0.5f, -1.f), 1.f);
    float sintheta = fminf(sqrtf(a0 * a0 + a1 * a1 + a2 * a2) * 0.5f, 1.f);
    const float theta = strict ? vsm_atan2f(sintheta, costheta) : atan2f(sintheta, costheta);
    if (sintheta > 1.1920929e-07f) {
        const float r = theta / (2.f * sintheta);
        aa[0] = a0 * r; aa[1] = a1 * r; aa[2] = a2 * r;
        return;
    }
    if (costheta > 0.f) { aa[0] = a0 * 0.5f; aa[1] = a1 * 0.5f; aa[2] = a2 * 0.5f; return; }
    // theta ~ pi: axis magnitudes from the diagonal; sintheta >= 0 here, so the reference's
    // sign fix-up (rodrigues.h:72-77) only flips negative components when sintheta > 0.
    const float inv = 1.f / (1.f - costheta);
    float b0 = theta * sqrtf((R[0] - costheta) * inv), b1 = theta * sqrtf((R[4] - costheta) * inv),
          b2 = theta * sqrtf((R[8] - costheta) * inv);
    if (sintheta > 0.f) { if (b0 < 0.f) b0 = -b0; if (b1 < 0.f) b1 = -b1; if (b2 < 0.f) b2 = -b2; }
    aa[0] = b0; aa[1] = b1; aa[2] = b2;
}
// angle-axis -> rotation matrix (cv::Rodrigues vec->mat as used at voldor/geometry.cpp:258)
VK_HD __forceinline__ void angle_axis_to_rotmat(const float* rv, float* R, bool strict = false) {
    double rx = rv[0], ry = rv[1], rz = rv[2];
    double th = sqrt(rx * rx + ry * ry + rz * rz);
    if (th < 2.220446049250313e-16) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
        return;
    }
    double c, s;
    if (strict) vsm_sincos(th, &s, &c);
    else { c = cos(th); s = sin(th); }
    const double c1 = 1. - c, it = 1. / th;
    rx *= it; ry *= it; rz *= it;
    R[0] = (float)(c + c1 * rx * rx);      R[1] = (float)(c1 * rx * ry - s * rz); R[2] = (float)(c1 * rx * rz + s * ry);
    R[3] = (float)(c1 * rx * ry + s * rz); R[4] = (float)(c + c1 * ry * ry);      R[5] = (float)(c1 * ry * rz - s * rx);
    R[6] = (float)(c1 * rx * rz - s * ry); R[7] = (float)(c1 * ry * rz + s * rx); R[8] = (float)(c + c1 * rz * rz);
}

// ---- AP3P ------------------------------------------------------------------------------------
// Float-complex helpers with the semantics of CUDA's cuComplex.h as the reference uses them.
struct Cx { float x, y; };
VK_HD __forceinline__ float cx_abs(Cx z) {
    float a = fabsf(z.x), b = fabsf(z.y), v = fmaxf(a, b), w = fminf(a, b);
    float t = w / v;
    t = v * sqrtf(1.0f + t * t);
    if (v == 0.0f || v > 3.402823466e38f || w > 3.402823466e38f) t = v + w;
    return t;
}
VK_HD __forceinline__ Cx cx_div(Cx x, Cx y) {
    float s = fabsf(y.x) + fabsf(y.y), oos = 1.0f / s;
    float ars = x.x * oos, ais = x.y * oos, brs = y.x * oos, bis = y.y * oos;
    s = brs * brs + bis * bis; oos = 1.0f / s;
    return { (ars * brs + ais * bis) * oos, (ais * brs - ars * bis) * oos };
}
VK_HD __forceinline__ Cx cx_sqrt(Cx x) {  // solve_batch_ap3p.cu:9-15 (principal root, Im <= 0)
    float m = cx_abs(x), u = x.x / m;
    return { sqrtf(m * (u + 1.0f) / 2.0f), -fabsf(sqrtf(m * (1.0f - u) / 2.0f)) };
}
// Ferrari quartic as written at solve_batch_ap3p.cu:28-82, INCLUDING the double square root
// in the q3<0 branch (:57, differs from OpenCV's ap3p.cpp): result parity with the reference
// wins over fixing it; the two Newton polish steps (:85-98) follow.
VK_HD static void ap3p_quartic(float a4, float a3, float a2, float a1, float a0, float& r0, float& r1, float& r2, float& r3, bool strict = false) {
    float a4_2 = a4 * a4, a3_2 = a3 * a3, a4_3 = a4_2 * a4, a2a4 = a2 * a4;
    float p4 = (8 * a2a4 - 3 * a3_2) / (8 * a4_2);
    float q4 = (a3_2 * a3 - 4 * a2a4 * a3 + 8 * a1 * a4_2) / (8 * a4_3);
    float r4 = (256 * a0 * a4_3 - 3 * (a3_2 * a3_2) - 64 * a1 * a3 * a4_2 + 16 * a2a4 * a3_2) / (256 * (a4_3 * a4));
    float p3 = ((p4 * p4) / 12 + r4) / 3;
    float q3 = (72 * r4 * p4 - 2 * p4 * p4 * p4 - 27 * q4 * q4) / 432;
    float t;
    Cx w = cx_sqrt({ q3 * q3 - p3 * p3 * p3, 0.f });
    if (q3 >= 0) { w.x = -w.x - q3; w.y = -w.y; }
    else { w = cx_sqrt(w); w.x = w.x - q3; }
    if (w.y == 0.0f) { w.x = strict ? vsm_cbrtf(w.x) : cbrtf(w.x); t = 2.0f * (w.x + p3 / w.x); }
    else {
        float theta = strict ? vsm_atan2f(w.y, w.x) : atan2f(w.y, w.x), mag = strict ? vsm_powf(cx_abs(w), 1.0f / 3.0f) : powf(cx_abs(w), 1.0f / 3.0f);
        t = 4.0f * (mag * (strict ? vsm_cosf((1.0f / 3.0f) * theta) : cosf(theta * (1.0f / 3.0f))));
    }
    Cx sq2m = cx_sqrt({ -2 * p4 / 3 + t, 0.f });
    float B_4A = -a3 / (4 * a4);
    Cx c1 = { 4 * p4 / 3 + t, 0.f };
    Cx c2 = cx_div({ 2 * q4, 0.f }, sq2m);
    float h = sq2m.x * 0.5f;
    float s1 = cx_sqrt({ -(c1.x + c2.x), -(c1.y + c2.y) }).x * 0.5f;
    float s2 = cx_sqrt({ -(c1.x - c2.x), -(c1.y - c2.y) }).x * 0.5f;
    r0 = B_4A + h + s1; r1 = B_4A + h - s1; r2 = B_4A - h + s2; r3 = B_4A - h - s2;
}
VK_HD __forceinline__ float ap3p_polish(float r, float c0, float c1, float c2, float c3, float c4) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
        float err = (((c0 * r + c1) * r + c2) * r + c3) * r + c4;
        float der = ((4 * c0 * r + 3 * c1) * r + 2 * c2) * r + c3;
        r -= err / der;
    }
    return r;
}
// cross product with the reference's vect_cross operand order (solve_batch_ap3p.cu:100-104)
VK_HD __forceinline__ V3<float> vcross(V3<float> a, V3<float> b) {
    return { a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x };
}
VK_HD static bool ap3p_p4p(const float* yu, const float* yv, const float (*xp)[3], float fx, float fy, float cx,
                                float cy, float* Rout, float* tout, bool strict = false) {
    typedef V3<float> F3;
    F3 bv[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {  // solve_all :294-318
        float u = (yu[i] - cx) / fx, v = (yv[i] - cy) / fy;
        float k = 1.f / sqrtf(u * u + v * v + 1);
        bv[i] = { u * k, v * k, k };
    }
    F3 w1 = { xp[0][0], xp[0][1], xp[0][2] }, w2 = { xp[1][0], xp[1][1], xp[1][2] }, w3 = { xp[2][0], xp[2][1], xp[2][2] };
    F3 b1 = bv[0], b2 = bv[1], b3 = bv[2];
    // computePoses :152-292
    F3 u0 = sub(w1, w2);
    float nu0 = sqrtf(dot(u0, u0));
    F3 k1 = { u0.x / nu0, u0.y / nu0, u0.z / nu0 };
    F3 k3 = vcross(b1, b2);
    float nk3 = sqrtf(dot(k3, k3));
    k3 = { k3.x / nk3, k3.y / nk3, k3.z / nk3 };
    F3 tz = vcross(b1, k3), v1 = vcross(b1, b3), v2 = vcross(b2, b3);
    F3 u1 = sub(w1, w3);
    float u1k1 = dot(u1, k1), k3b3 = dot(k3, b3);
    float f11 = k3b3, f13 = dot(k3, v1), f15 = -u1k1 * f11;
    F3 nl = vcross(u1, k1);
    float delta = sqrtf(dot(nl, nl));
    nl = { nl.x / delta, nl.y / delta, nl.z / delta };
    f11 *= delta; f13 *= delta;
    float u2k1 = u1k1 - nu0;
    float f21 = dot(tz, v2), f22 = nk3 * k3b3, f23 = dot(k3, v2);
    float f24 = u2k1 * f22, f25 = -u2k1 * f21;
    f21 *= delta; f22 *= delta; f23 *= delta;
    float g1 = f13 * f22, g2 = f13 * f25 - f15 * f23, g3 = f11 * f23 - f13 * f21, g4 = -f13 * f24;
    float g5 = f11 * f22, g6 = f11 * f25 - f15 * f21, g7 = -f15 * f24;
    float q0 = g5 * g5 + g1 * g1 + g3 * g3;
    float q1 = 2 * (g5 * g6 + g1 * g2 + g3 * g4);
    float q2 = g6 * g6 + 2 * g5 * g7 + g2 * g2 + g4 * g4 - g1 * g1 - g3 * g3;
    float q3 = 2 * (g6 * g7 - g1 * g2 - g3 * g4);
    float q4 = g7 * g7 - g2 * g2 - g4 * g4;
    float s0, s1, s2, s3;
    ap3p_quartic(q0, q1, q2, q3, q4, s0, s1, s2, s3, strict);
    // polishQuarticRoots interleaves the 4 roots per iteration but roots are independent
    s0 = ap3p_polish(s0, q0, q1, q2, q3, q4); s1 = ap3p_polish(s1, q0, q1, q2, q3, q4);
    s2 = ap3p_polish(s2, q0, q1, q2, q3, q4); s3 = ap3p_polish(s3, q0, q1, q2, q3, q4);
    F3 tmp = vcross(k1, nl);
    // Ck1nl = [k1 nl tmp] (columns), Cb1k3tzT = rows b1,k3,tz
    F3 b3p = scale(b3, delta / k3b3);
    const F3 x4 = { xp[3][0], xp[3][1], xp[3][2] };
    int n = 0;
    float best = 0.f;
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
        float ct1 = i == 0 ? s0 : (i == 1 ? s1 : (i == 2 ? s2 : s3));
        if (fabsf(ct1) > 1) continue;
        float st1 = sqrtf(1 - ct1 * ct1);
        st1 = (k3b3 > 0) ? st1 : -st1;
        float ct3 = g1 * ct1 + g2, st3 = g3 * ct1 + g4;
        float nt3 = st1 / ((g5 * ct1 + g6) * ct1 + g7);
        ct3 *= nt3; st3 *= nt3;
        // C13 rows
        float C[9] = { ct3, 0, -st3, st1 * st3, ct1, st1 * ct3, ct1 * st3, -st1, ct1 * ct3 };
        // T = Ck1nl * C13
        float T[9];
        const float A[9] = { k1.x, nl.x, tmp.x, k1.y, nl.y, tmp.y, k1.z, nl.z, tmp.z };
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) T[r * 3 + c] = A[r * 3] * C[c] + A[r * 3 + 1] * C[3 + c] + A[r * 3 + 2] * C[6 + c];
        const float Bm[9] = { b1.x, b1.y, b1.z, k3.x, k3.y, k3.z, tz.x, tz.y, tz.z };
        float R[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) R[r * 3 + c] = T[r * 3] * Bm[c] + T[r * 3 + 1] * Bm[3 + c] + T[r * 3 + 2] * Bm[6 + c];
        // rp3 = R^T w3 ; t = st1*b3p - rp3 ; solution rotation = R^T (:263-287)
        float rp0 = w3.x * R[0] + w3.y * R[3] + w3.z * R[6];
        float rp1 = w3.x * R[1] + w3.y * R[4] + w3.z * R[7];
        float rp2 = w3.x * R[2] + w3.y * R[5] + w3.z * R[8];
        float t0 = b3p.x * st1 - rp0, t1 = b3p.y * st1 - rp1, t2 = b3p.z * st1 - rp2;
        float Rt[9] = { R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8] };
        float X = Rt[0] * x4.x + Rt[1] * x4.y + Rt[2] * x4.z + t0;
        float Y = Rt[3] * x4.x + Rt[4] * x4.y + Rt[5] * x4.z + t1;
        float Z = Rt[6] * x4.x + Rt[7] * x4.y + Rt[8] * x4.z + t2;
        float du = cx + fx * X / Z - yu[3], dv = cy + fy * Y / Z - yv[3];
        float err = du * du + dv * dv;
        if (n == 0 || best > err) {
#pragma unroll
            for (int k = 0; k < 9; k++) Rout[k] = Rt[k];
            tout[0] = t0; tout[1] = t1; tout[2] = t2; best = err;
        }
        n++;
    }
    return n > 0;
}

}  // namespace vk
