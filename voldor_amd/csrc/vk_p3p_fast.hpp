// voldor_amd/csrc/vk_p3p_fast.hpp -- the fast window pipeline's LambdaTwist P3P + 4th-point test: the reference's float instantiation with its double
// literals (vk_p3p.hpp: lambdatwist/lambdatwist_p3p.h:38-137, solve_cubic.h:17-34,160-207, solve_eig0.h:40-67, refine_lambda.h:31-57) evaluated on
// CHEAPER INSTRUCTIONS THAT ROUND THE SAME WAY (round 5).
//
// A third of the 3 712 VALU instructions a wave of k_solve issues are fp64, and most of those are IEEE fp64 divisions and square roots (~35 / ~25
// dependent instructions each) the reference's `1.0 / x`, `2.0 * c / x`, `1.0 / sqrt(.. + 1.0)` promote to.  P3P on noisy flow is ill conditioned: a
// last-bit change of an intermediate moves a hypothesis by ~1e-5 (measured: median |dt| 7e-6, 1 % above 8e-4 between two float evaluations of the same
// algorithm), and a plain-fp32 solver (fused multiply-adds, v_rcp_f32) -- 17.7 -> 13.3 us per launch -- fails the ensemble test (tests/test_gpu_ensemble.py:
// depth distance to the reference 1.12x its self-noise).  So the VALUES are kept and only the instructions change:
//   * a double division whose operands are floats (or exact doubles: 2.0 * c) and whose result is rounded to float IS the IEEE float division up to
//     double rounding (a 2^-29 event): `1.0 / x`, `2.0 * c / x`, `(-b - v) / 3.0` -> one fp32 division each (10 instructions instead of ~35);
//   * round32(1 / sqrt(D)) for a double D: v_rsq_f32 + ONE Newton step in double (relative error 2^-45, then the rounding to float: same float
//     except within 2^-45 of a rounding boundary) instead of a double square root + a double division;
//   * sums the reference takes in double (the cubic's coefficients, discriminants) stay in double -- additions are cheap;
//   * the start value of the cubic's Newton iteration and its 50 steps are the reference's own operations (the loop is float already);
//   * nearest_rotation (rodrigues.h:82-108; D8: the exact polar factor) on fp32: its output goes straight into the hypothesis (no amplification), and a
//     P3P solution is orthonormal up to the rounding of its construction: two or three Newton steps, 1e-7 in the rotation vector.
// What a hypothesis is therefore differs from vk_p3p.hpp's by bits only where double rounding or the 2^-45 boundary strikes (tests/test_gpu_kernels.py
// counts them: > 99.9 % of the translations are bit-identical).  k_solve takes this path in fast mode with the float LambdaTwist solver
// (vk_debug_switch "solve_fp32" = 0: vk_p3p.hpp as it is; strict mode and the host-pointer API always run that one).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include "vk_p3p.hpp"
#define VKF_HD __host__ __device__

namespace vk {
namespace p3pf {

// round32(1 / sqrt(D)), D > 0 double (see above)
VKF_HD __forceinline__ float rsqrt_rounded(double D) {
#pragma clang fp contract(off)
#if defined(__HIP_DEVICE_COMPILE__)
    const double y0 = (double)__builtin_amdgcn_rsqf((float)D);
#else
    const double y0 = (double)(1.f / sqrtf((float)D));
#endif
    return (float)(y0 * (1.5 - (0.5 * D) * (y0 * y0)));
}

// x^2 + b x + c = 0 (solve_cubic.h:13-35; root2real<float> of vk_p3p.hpp)
VKF_HD __forceinline__ bool root2real(float b, float c, float& r1, float& r2) {
#pragma clang fp contract(off)
    const float v = (float)((double)(b * b) - 4.0 * (double)c);  // one double subtraction: what `b * b - 4.0 * c` is
    if (v < 0) { r1 = r2 = 0.5f * b; return false; }
    const float y = sqrtf(v);
    if (b < 0) { r1 = 0.5f * (-b + y); r2 = 0.5f * (-b - y); }
    else { const float c2 = 2.f * c; r1 = c2 / (-b + y); r2 = c2 / (-b - y); }  // 2.0 * c is exact: the double division rounds like the float one
    return true;
}

// eig_known0<float> of vk_p3p.hpp (solve_eig0.h:11-80)
VKF_HD __forceinline__ void eig_known0(const float* x, float& e1, float& e2, V3<float>& v1, V3<float>& v2) {
#pragma clang fp contract(off)
    const float x01s = x[1] * x[1];
    const float b = -x[0] - x[4] - x[8];
    const float c = -x01s - x[2] * x[2] - x[5] * x[5] + x[0] * (x[4] + x[8]) + x[4] * x[8];
    root2real(b, c, e1, e2);
    if (fabsf(e1) < fabsf(e2)) { const float t = e1; e1 = e2; e2 = t; }
    const float mx0011 = -x[0] * x[4];
    const float prec0 = x[1] * x[5] - x[2] * x[4];
    const float prec1 = x[1] * x[2] - x[0] * x[5];
    {
        const float tmp = 1.f / (e1 * (x[0] + x[4]) + mx0011 - e1 * e1 + x01s);
        float a1 = -(e1 * x[2] + prec0) * tmp, a2 = -(e1 * x[5] + prec1) * tmp;
        const float rn = rsqrt_rounded((double)(a1 * a1 + a2 * a2) + 1.0);
        a1 *= rn; a2 *= rn;
        v1 = { a1, a2, rn };
    }
    {
        const float tmp = 1.f / (e2 * (x[0] + x[4]) + mx0011 - e2 * e2 + x01s);
        float a1 = -(e2 * x[2] + prec0) * tmp, a2 = -(e2 * x[5] + prec1) * tmp;
        const float rn = rsqrt_rounded((double)(a1 * a1 + a2 * a2) + 1.0);
        a1 *= rn; a2 *= rn;
        v2 = { a1, a2, rn };
    }
}

// refine_lambda<float> of vk_p3p.hpp (refine_lambda.h:5-102)
VKF_HD __forceinline__ void refine_lambda(V3<float>& L, float a12, float a13, float a23, float b12, float b13, float b23) {
#pragma clang fp contract(off)
#pragma unroll 1
    for (int i = 0; i < 5; ++i) {
        const float l1 = L.x, l2 = L.y, l3 = L.z;
        const float r1 = l1 * l1 + l2 * l2 + b12 * l1 * l2 - a12;
        const float r2 = l1 * l1 + l3 * l3 + b13 * l1 * l3 - a13;
        const float r3 = l2 * l2 + l3 * l3 + b23 * l2 * l3 - a23;
        if ((double)(fabsf(r1) + fabsf(r2) + fabsf(r3)) < 1e-10) break;
        // (2.0) * l + b * l': the double sum of an exact product and a float product, rounded to float = the float sum of the two
        const float v0 = 2.f * l1 + b12 * l2, v1 = 2.f * l2 + b12 * l1;
        const float v3 = 2.f * l1 + b13 * l3, v5 = 2.f * l3 + b13 * l1;
        const float v7 = 2.f * l2 + b23 * l3, v8 = 2.f * l3 + b23 * l2;
        const float det = 1.f / (-v0 * v5 * v7 - v1 * v3 * v8);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        s0 += (-v5 * v7) * r1; s0 += (-v1 * v8) * r2; s0 += (v1 * v5) * r3;
        s1 += (-v3 * v8) * r1; s1 += (v0 * v8) * r2; s1 += (-v0 * v5) * r3;
        s2 += (v3 * v7) * r1; s2 += (-v0 * v7) * r2; s2 += (-v1 * v3) * r3;
        const V3<float> n = { l1 - s0 * det, l2 - s1 * det, l3 - s2 * det };
        const float q1 = n.x * n.x + n.y * n.y + b12 * n.x * n.y - a12;
        const float q2 = n.x * n.x + n.z * n.z + b13 * n.x * n.z - a13;
        const float q3 = n.y * n.y + n.z * n.z + b23 * n.y * n.z - a23;
        if (fabsf(q1) + fabsf(q2) + fabsf(q3) > fabsf(r1) + fabsf(r2) + fabsf(r3)) break;
        L = n;
    }
}

// Candidate `only` (block only >> 1: s = +v / -v; root only & 1: tau1 / tau2) of the up-to-four P3P solutions of points 0..2 and its
// reprojection error on point 3: lambdatwist_p4p<float>(.., only, err_out) of vk_p3p.hpp, value for value.
VKF_HD static bool lambdatwist_candidate(const float* yu, const float* yv, const float (*xp)[3], float fxf, float fyf, float cxf, float cyf, float* Rout, float* tout,
                                         int only, float* err_out) {
#pragma clang fp contract(off)
    typedef float S;
    V3<S> y1 = normalized<S>({ (S)((yu[0] - cxf) / fxf), (S)((yv[0] - cyf) / fyf), S(1.0) });
    V3<S> y2 = normalized<S>({ (S)((yu[1] - cxf) / fxf), (S)((yv[1] - cyf) / fyf), S(1.0) });
    V3<S> y3 = normalized<S>({ (S)((yu[2] - cxf) / fxf), (S)((yv[2] - cyf) / fyf), S(1.0) });
    V3<S> x1 = { xp[0][0], xp[0][1], xp[0][2] }, x2 = { xp[1][0], xp[1][1], xp[1][2] }, x3 = { xp[2][0], xp[2][1], xp[2][2] };
    S b12 = -2.f * (dot(y1, y2)), b13 = -2.f * (dot(y1, y3)), b23 = -2.f * (dot(y2, y3));  // (-2.0 * float: exact)
    V3<S> d12 = sub(x1, x2), d13 = sub(x1, x3), d23 = sub(x2, x3), dc = cross(d12, d13);
    S a12 = dot(d12, d12), a13 = dot(d13, d13), a23 = dot(d23, d23);
    S c31 = -0.5f * b13, c23 = -0.5f * b23, c12 = -0.5f * b12;
    // the sums the reference takes in double (a double literal in the expression) stay in double: additions and multiplications are cheap
    S blob = (c12 * c23 * c31 - 1.0);
    S s31 = 1.0 - c31 * c31, s23 = 1.0 - c23 * c23, s12 = 1.0 - c12 * c12;
    S p3 = (a13 * (a23 * s31 - a13 * s23));
    S p2 = 2.0 * blob * a23 * a13 + a13 * (2.0 * a12 + a13) * s23 + a23 * (a23 - a12) * s31;
    S p1 = a23 * (a13 - a23) * s12 - a12 * a12 * s23 - 2.0 * a12 * (blob * a23 + a13 * s23);
    S p0 = a12 * (a12 * s23 - a23 * s12);
    p3 = 1.f / p3;  // 1.0 / p3 rounded to float
    p2 *= p3; p1 *= p3; p0 *= p3;
    S g = cubic_root<S>(p2, p1, p0);  // vk_p3p.hpp: the start value and the 50 float Newton steps as they are

    S A[9];
    A[0] = a23 * (1.0 - g); A[1] = (a23 * b12) * 0.5f; A[2] = (a23 * b13 * g) * (-0.5f);
    A[4] = a23 - a12 + a13 * g; A[5] = b23 * (a13 * g - a12) * 0.5f; A[8] = g * (a13 - a23) - a12;
    A[3] = A[1]; A[6] = A[2]; A[7] = A[5];
    S e1, e2; V3<S> v1, v2;
    eig_known0(A, e1, e2, v1, v2);
    S v = sqrtf(-e2 / e1 > 0 ? -e2 / e1 : S(0));

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
    bool have = false;
    S err = S(0);
    const S s = (only >> 1) == 0 ? v : -v;
    S w2 = S(1.0) / (s * v2.x - v1.x);
    S w0 = (v1.y - s * v2.y) * w2;
    S w1 = (v1.z - s * v2.z) * w2;
    S a = S(1.0) / ((a13 - a12) * w1 * w1 - a12 * b13 * w1 - a12);
    S b = (a13 * b12 * w1 - a12 * b13 * w0 - S(2.0) * w0 * w1 * (a12 - a13)) * a;
    S c = ((a13 - a12) * w0 * w0 + a13 * b12 * w0 + a13) * a;
    if ((double)(b * b) - 4.0 * (double)c >= 0) {
        S tau1, tau2;
        root2real(b, c, tau1, tau2);
        const S tau = (only & 1) == 0 ? tau1 : tau2;
        if (tau > 0) {
            S d = a23 / (tau * (b23 + tau) + S(1.0));
            if (d > 0) {
                S l2 = sqrtf(d), l3 = tau * l2, l1 = w0 * l2 + w1 * l3;
                if (l1 >= 0) {
                    V3<S> L = { l1, l2, l3 };
                    refine_lambda(L, a12, a13, a23, b12, b13, b23);
                    V3<S> ry1 = scale(y1, L.x), ry2 = scale(y2, L.y), ry3 = scale(y3, L.z);
                    V3<S> yd1 = sub(ry1, ry2), yd2 = sub(ry1, ry3), yc = cross(yd1, yd2);
                    const S Y[9] = { yd1.x, yd2.x, yc.x, yd1.y, yd2.y, yc.y, yd1.z, yd2.z, yc.z };
                    S R[9];
#pragma unroll
                    for (int r = 0; r < 3; r++)
#pragma unroll
                        for (int cc = 0; cc < 3; cc++) {
                            S sum = S(0);
                            sum += Y[r * 3] * Xi[cc]; sum += Y[r * 3 + 1] * Xi[3 + cc]; sum += Y[r * 3 + 2] * Xi[6 + cc];
                            R[r * 3 + cc] = sum;
                        }
                    S t[3];
                    const S ry[3] = { ry1.x, ry1.y, ry1.z };
#pragma unroll
                    for (int r = 0; r < 3; r++) {
                        S sum = S(0);
                        sum += R[r * 3] * x1.x; sum += R[r * 3 + 1] * x1.y; sum += R[r * 3 + 2] * x1.z;
                        t[r] = ry[r] - sum;
                    }
                    const float* x4 = xp[3];
                    S X = R[0] * x4[0] + R[1] * x4[1] + R[2] * x4[2] + t[0];
                    S Yp = R[3] * x4[0] + R[4] * x4[1] + R[5] * x4[2] + t[1];
                    S Z = R[6] * x4[0] + R[7] * x4[1] + R[8] * x4[2] + t[2];
                    S mu = cxf + fxf * X / Z, mv = cyf + fyf * Yp / Z;
                    err = (mu - yu[3]) * (mu - yu[3]) + (mv - yv[3]) * (mv - yv[3]);
#pragma unroll
                    for (int k = 0; k < 9; k++) Rout[k] = R[k];
                    tout[0] = t[0]; tout[1] = t[1]; tout[2] = t[2];
                    have = true;
                }
            }
        }
    }
    *err_out = err;
    return have;
}

// polar factor by Newton's iteration X <- (X + X^-T) / 2 (what nearest_rotation of vk_p3p.hpp runs in fp64; the reference: U V^T of an approximate
// fp32 SVD, rodrigues.h:82-108).  A P3P solution is orthonormal up to the rounding of its construction, so the iteration is at its fixed point
// (to fp32 resolution) after two or three steps; a degenerate 4-tuple's near-singular matrix runs into the step bound and stays what it is -- an
// outlier of the pool whichever way it is rounded.
VKF_HD __forceinline__ void nearest_rotation(float* X) {
#pragma clang fp contract(fast)
    auto rcp = [](float x) { return 1.f / x; };
#pragma unroll 1
    for (int it = 0; it < 12; it++) {
        float c[9];
        c[0] = fmaf(X[4], X[8], -X[5] * X[7]); c[1] = fmaf(X[5], X[6], -X[3] * X[8]); c[2] = fmaf(X[3], X[7], -X[4] * X[6]);
        c[3] = fmaf(X[2], X[7], -X[1] * X[8]); c[4] = fmaf(X[0], X[8], -X[2] * X[6]); c[5] = fmaf(X[1], X[6], -X[0] * X[7]);
        c[6] = fmaf(X[1], X[5], -X[2] * X[4]); c[7] = fmaf(X[2], X[3], -X[0] * X[5]); c[8] = fmaf(X[0], X[4], -X[1] * X[3]);
        const float det = fmaf(X[0], c[0], fmaf(X[1], c[1], X[2] * c[2]));
        if (!(fabsf(det) > 1e-30f)) break;
        const float hid = 0.5f * rcp(det);
        float delta = 0.f;
#pragma unroll
        for (int i = 0; i < 9; i++) { const float y = fmaf(c[i], hid, 0.5f * X[i]); delta += fabsf(y - X[i]); X[i] = y; }
        if (delta < 2e-6f) break;  // quadratic convergence: the next step would move X by ~delta^2, far below an fp32 ulp of its entries
    }
}

}  // namespace p3pf
}  // namespace vk
