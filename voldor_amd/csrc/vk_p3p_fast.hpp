// voldor_amd/csrc/vk_p3p_fast.hpp -- the fast window pipeline's LambdaTwist P3P + 4th-point test in plain fp32 (round 5).
//
// vk_p3p.hpp restates the reference's instantiation lambdatwist_p4p<float> WITH its double literals (lambdatwist/lambdatwist_p3p.h:38-137,
// solve_cubic.h:17-34,160-207, solve_eig0.h:40-67, refine_lambda.h:31-57): a third of the 3 712 VALU instructions a wave of k_solve issues
// are fp64 (half rate, and an IEEE fp64 division is ~35 dependent instructions), kept so that strict mode and the host-pointer API
// reproduce the reference kernel's translations to the bit.  The fast pipeline is not bit-identical to the reference anyway (hardware
// transcendentals, re-associated sums: held to it as a distribution, tests/test_gpu_ensemble.py), and a pose hypothesis is one sample of
// 8192 in a kernel density estimate -- so here the same algorithm, step for step and branch for branch, runs on fp32 fused multiply-adds,
// v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 (1 ulp) instead of IEEE divisions, and the polar factor of nearest_rotation on fp32.  The Newton loop of
// the cubic keeps the reference's 50 steps and its exit test: what the ~1.4 % of cubics that never settle end on after 50 steps is part of
// the pool's distribution (a 12-step cap fails the ensemble test: vk_p3p.hpp).
// k_solve takes this path in fast mode with the float LambdaTwist solver (vk_debug_switch "solve_fp32" = 0: the restated rounding sequence).
#pragma once
#include <hip/hip_runtime.h>

namespace vk {
namespace p3pf {

struct F3 { float x, y, z; };
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float sqrt1(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) { return { fmaf(a.y, b.z, -a.z * b.y), fmaf(a.z, b.x, -a.x * b.z), fmaf(a.x, b.y, -a.y * b.x) }; }
__device__ __forceinline__ F3 sub3(F3 a, F3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
__device__ __forceinline__ F3 scale3(F3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
__device__ __forceinline__ F3 unit3(F3 a) { return scale3(a, rsq(dot3(a, a))); }

// x^2 + b x + c = 0 (solve_cubic.h:13-35)
__device__ __forceinline__ bool root2real(float b, float c, float& r1, float& r2) {
    const float v = fmaf(b, b, -4.f * c);
    if (v < 0.f) { r1 = r2 = 0.5f * b; return false; }
    const float y = sqrt1(v);
    if (b < 0.f) { r1 = 0.5f * (y - b); r2 = 0.5f * (-b - y); }
    else { const float c2 = 2.f * c; r1 = c2 * rcp(y - b); r2 = c2 * rcp(-b - y); }
    return true;
}
// one real root of x^3 + b x^2 + c x + d (solve_cubic.h:154-210): the reference's start value, 50 Newton steps, its exit test
__device__ __forceinline__ float cubic_root(float b, float c, float d) {
    float r0;
    const float disc = fmaf(b, b, -3.f * c);
    if (disc >= 0.f) {
        const float v = sqrt1(disc);
        const float t1 = (-b - v) * (1.f / 3.f);
        float k = fmaf(fmaf(t1 + b, t1, c), t1, d);
        if (k > 0.f) r0 = t1 - sqrt1(-k * rcp(fmaf(3.f, t1, b)));
        else {
            const float t2 = (v - b) * (1.f / 3.f);
            k = fmaf(fmaf(t2 + b, t2, c), t2, d);
            r0 = t2 + sqrt1(-k * rcp(fmaf(3.f, t2, b)));
        }
    } else {
        r0 = -b * (1.f / 3.f);
        if (fabsf(fmaf(fmaf(3.f, r0, 2.f * b), r0, c)) < 1e-4f) r0 += 1.f;
    }
    const float b2 = 2.f * b;
#pragma unroll 1
    for (int cnt = 0; cnt < 50; ++cnt) {
        const float fx = fmaf(fmaf(r0 + b, r0, c), r0, d);
        if (cnt < 7 || fabsf(fx) > 1e-7f) {
            const float fpx = fmaf(fmaf(3.f, r0, b2), r0, c);
            r0 = fmaf(-fx, rcp(fpx), r0);
        } else
            break;
    }
    return r0;
}
// symmetric 3x3 with one zero eigenvalue (solve_eig0.h:11-80): the two other eigenvalues and their vectors
__device__ __forceinline__ void eig_known0(const float* x, float& e1, float& e2, F3& v1, F3& v2) {
    const float x01s = x[1] * x[1];
    const float b = -x[0] - x[4] - x[8];
    const float c = fmaf(x[4], x[8], fmaf(x[0], x[4] + x[8], -x01s - x[2] * x[2] - x[5] * x[5]));
    root2real(b, c, e1, e2);
    if (fabsf(e1) < fabsf(e2)) { const float t = e1; e1 = e2; e2 = t; }
    const float mx0011 = -x[0] * x[4];
    const float prec0 = fmaf(x[1], x[5], -x[2] * x[4]);
    const float prec1 = fmaf(x[1], x[2], -x[0] * x[5]);
    const float s04 = x[0] + x[4];
    {
        const float tmp = rcp(fmaf(e1, s04, mx0011) - fmaf(e1, e1, -x01s));
        float a1 = -fmaf(e1, x[2], prec0) * tmp, a2 = -fmaf(e1, x[5], prec1) * tmp;
        const float rn = rsq(fmaf(a1, a1, fmaf(a2, a2, 1.f)));
        v1 = { a1 * rn, a2 * rn, rn };
    }
    {
        const float tmp = rcp(fmaf(e2, s04, mx0011) - fmaf(e2, e2, -x01s));
        float a1 = -fmaf(e2, x[2], prec0) * tmp, a2 = -fmaf(e2, x[5], prec1) * tmp;
        const float rn = rsq(fmaf(a1, a1, fmaf(a2, a2, 1.f)));
        v2 = { a1 * rn, a2 * rn, rn };
    }
}
// 5 Gauss-Newton steps on the three law-of-cosines residuals (refine_lambda.h:5-102)
__device__ __forceinline__ void refine_lambda(F3& L, float a12, float a13, float a23, float b12, float b13, float b23) {
#pragma unroll 1
    for (int i = 0; i < 5; ++i) {
        const float l1 = L.x, l2 = L.y, l3 = L.z;
        const float r1 = fmaf(l1, l1, fmaf(l2, l2, fmaf(b12 * l1, l2, -a12)));
        const float r2 = fmaf(l1, l1, fmaf(l3, l3, fmaf(b13 * l1, l3, -a13)));
        const float r3 = fmaf(l2, l2, fmaf(l3, l3, fmaf(b23 * l2, l3, -a23)));
        const float rs = fabsf(r1) + fabsf(r2) + fabsf(r3);
        if (rs < 1e-10f) break;
        const float v0 = fmaf(2.f, l1, b12 * l2), v1 = fmaf(2.f, l2, b12 * l1);
        const float v3 = fmaf(2.f, l1, b13 * l3), v5 = fmaf(2.f, l3, b13 * l1);
        const float v7 = fmaf(2.f, l2, b23 * l3), v8 = fmaf(2.f, l3, b23 * l2);
        const float det = rcp(-v0 * v5 * v7 - v1 * v3 * v8);
        const float s0 = fmaf(v1 * v5, r3, fmaf(-v1 * v8, r2, (-v5 * v7) * r1));
        const float s1 = fmaf(-v0 * v5, r3, fmaf(v0 * v8, r2, (-v3 * v8) * r1));
        const float s2 = fmaf(-v1 * v3, r3, fmaf(-v0 * v7, r2, (v3 * v7) * r1));
        const F3 n = { fmaf(-s0, det, l1), fmaf(-s1, det, l2), fmaf(-s2, det, l3) };
        const float q1 = fmaf(n.x, n.x, fmaf(n.y, n.y, fmaf(b12 * n.x, n.y, -a12)));
        const float q2 = fmaf(n.x, n.x, fmaf(n.z, n.z, fmaf(b13 * n.x, n.z, -a13)));
        const float q3 = fmaf(n.y, n.y, fmaf(n.z, n.z, fmaf(b23 * n.y, n.z, -a23)));
        if (fabsf(q1) + fabsf(q2) + fabsf(q3) > rs) break;
        L = n;
    }
}

// Candidate `only` (block only >> 1: s = +v / -v; root only & 1: tau1 / tau2) of the up-to-four P3P solutions of points 0..2 and its
// reprojection error on point 3 (lambdatwist_p4p.h:5-62, lambdatwist_p3p.h:19-294), like lambdatwist_p4p<float>(.., only, err_out) of vk_p3p.hpp:
// k_solve spreads the candidates of a hypothesis over four lanes and folds them with the reference's "first, then strictly better" rule.
__device__ static bool lambdatwist_candidate(const float* yu, const float* yv, const float (*xp)[3], float fx, float fy, float cx, float cy, float* Rout, float* tout,
                                             int only, float* err_out) {
    const float ifx = rcp(fx), ify = rcp(fy);
    const F3 y1 = unit3({ (yu[0] - cx) * ifx, (yv[0] - cy) * ify, 1.f });
    const F3 y2 = unit3({ (yu[1] - cx) * ifx, (yv[1] - cy) * ify, 1.f });
    const F3 y3 = unit3({ (yu[2] - cx) * ifx, (yv[2] - cy) * ify, 1.f });
    const F3 x1 = { xp[0][0], xp[0][1], xp[0][2] }, x2 = { xp[1][0], xp[1][1], xp[1][2] }, x3 = { xp[2][0], xp[2][1], xp[2][2] };
    const float b12 = -2.f * dot3(y1, y2), b13 = -2.f * dot3(y1, y3), b23 = -2.f * dot3(y2, y3);
    const F3 d12 = sub3(x1, x2), d13 = sub3(x1, x3), d23 = sub3(x2, x3), dc = cross3(d12, d13);
    const float a12 = dot3(d12, d12), a13 = dot3(d13, d13), a23 = dot3(d23, d23);
    const float c31 = -0.5f * b13, c23 = -0.5f * b23, c12 = -0.5f * b12;
    const float blob = fmaf(c12 * c23, c31, -1.f);
    const float s31 = fmaf(-c31, c31, 1.f), s23 = fmaf(-c23, c23, 1.f), s12 = fmaf(-c12, c12, 1.f);
    float p3 = a13 * fmaf(a23, s31, -a13 * s23);
    float p2 = fmaf(2.f * blob * a23, a13, fmaf(a13 * fmaf(2.f, a12, a13), s23, a23 * (a23 - a12) * s31));
    float p1 = fmaf(a23 * (a13 - a23), s12, fmaf(-a12 * a12, s23, -2.f * a12 * fmaf(blob, a23, a13 * s23)));
    float p0 = a12 * fmaf(a12, s23, -a23 * s12);
    p3 = rcp(p3);
    p2 *= p3; p1 *= p3; p0 *= p3;
    const float g = cubic_root(p2, p1, p0);

    float A[9];
    A[0] = a23 * (1.f - g); A[1] = (a23 * b12) * 0.5f; A[2] = (a23 * b13 * g) * (-0.5f);
    A[4] = fmaf(a13, g, a23 - a12); A[5] = b23 * fmaf(a13, g, -a12) * 0.5f; A[8] = fmaf(g, a13 - a23, -a12);
    A[3] = A[1]; A[6] = A[2]; A[7] = A[5];
    float e1, e2; F3 v1, v2;
    eig_known0(A, e1, e2, v1, v2);
    const float ratio = -e2 * rcp(e1);
    const float v = sqrt1(ratio > 0.f ? ratio : 0.f);

    // X^-1, X = [d12 d13 d12 x d13] (columns), adjugate form (matrix.h:636-656)
    const float Xm[9] = { d12.x, d13.x, dc.x, d12.y, d13.y, dc.y, d12.z, d13.z, dc.z };
    float Xi[9];
    {
        const float M0 = fmaf(Xm[4], Xm[8], -Xm[5] * Xm[7]), M1 = fmaf(Xm[2], Xm[7], -Xm[1] * Xm[8]), M2 = fmaf(Xm[1], Xm[5], -Xm[2] * Xm[4]);
        const float M3 = fmaf(Xm[5], Xm[6], -Xm[3] * Xm[8]), M4 = fmaf(Xm[0], Xm[8], -Xm[2] * Xm[6]), M5 = fmaf(Xm[2], Xm[3], -Xm[0] * Xm[5]);
        const float M6 = fmaf(Xm[3], Xm[7], -Xm[4] * Xm[6]), M7 = fmaf(Xm[1], Xm[6], -Xm[0] * Xm[7]), M8 = fmaf(Xm[0], Xm[4], -Xm[1] * Xm[3]);
        const float idet = rcp(fmaf(Xm[0], M0, fmaf(Xm[1], M3, Xm[2] * M6)));
        Xi[0] = M0 * idet; Xi[1] = M1 * idet; Xi[2] = M2 * idet; Xi[3] = M3 * idet; Xi[4] = M4 * idet;
        Xi[5] = M5 * idet; Xi[6] = M6 * idet; Xi[7] = M7 * idet; Xi[8] = M8 * idet;
    }
    bool have = false;
    float err = 0.f;
    const float s = (only >> 1) == 0 ? v : -v;
    const float w2 = rcp(fmaf(s, v2.x, -v1.x));
    const float w0 = fmaf(-s, v2.y, v1.y) * w2;
    const float w1 = fmaf(-s, v2.z, v1.z) * w2;
    const float a = rcp(fmaf((a13 - a12) * w1, w1, fmaf(-a12 * b13, w1, -a12)));
    const float b = fmaf(a13 * b12, w1, fmaf(-a12 * b13, w0, -2.f * w0 * w1 * (a12 - a13))) * a;
    const float c = fmaf((a13 - a12) * w0, w0, fmaf(a13 * b12, w0, a13)) * a;
    if (fmaf(b, b, -4.f * c) >= 0.f) {
        float tau1, tau2;
        root2real(b, c, tau1, tau2);
        const float tau = (only & 1) == 0 ? tau1 : tau2;
        if (tau > 0.f) {
            const float d = a23 * rcp(fmaf(tau, b23 + tau, 1.f));
            if (d > 0.f) {
                const float l2 = sqrt1(d), l3 = tau * l2, l1 = fmaf(w0, l2, w1 * l3);
                if (l1 >= 0.f) {
                    F3 L = { l1, l2, l3 };
                    refine_lambda(L, a12, a13, a23, b12, b13, b23);
                    const F3 ry1 = scale3(y1, L.x), ry2 = scale3(y2, L.y), ry3 = scale3(y3, L.z);
                    const F3 yd1 = sub3(ry1, ry2), yd2 = sub3(ry1, ry3), yc = cross3(yd1, yd2);
                    const float Y[9] = { yd1.x, yd2.x, yc.x, yd1.y, yd2.y, yc.y, yd1.z, yd2.z, yc.z };  // R = Y X^-1
                    float R[9];
#pragma unroll
                    for (int r = 0; r < 3; r++)
#pragma unroll
                        for (int cc = 0; cc < 3; cc++) R[r * 3 + cc] = fmaf(Y[r * 3 + 2], Xi[6 + cc], fmaf(Y[r * 3 + 1], Xi[3 + cc], Y[r * 3] * Xi[cc]));
                    const float ry[3] = { ry1.x, ry1.y, ry1.z };
                    float t[3];
#pragma unroll
                    for (int r = 0; r < 3; r++) t[r] = ry[r] - fmaf(R[r * 3 + 2], x1.z, fmaf(R[r * 3 + 1], x1.y, R[r * 3] * x1.x));
                    // 4th-point reprojection (lambdatwist_p4p.h:31-41)
                    const float* x4 = xp[3];
                    const float X = fmaf(R[2], x4[2], fmaf(R[1], x4[1], fmaf(R[0], x4[0], t[0])));
                    const float Yp = fmaf(R[5], x4[2], fmaf(R[4], x4[1], fmaf(R[3], x4[0], t[1])));
                    const float Z = fmaf(R[8], x4[2], fmaf(R[7], x4[1], fmaf(R[6], x4[0], t[2])));
                    const float iz = rcp(Z);
                    const float du = fmaf(fx * X, iz, cx) - yu[3], dv = fmaf(fy * Yp, iz, cy) - yv[3];
                    err = fmaf(du, du, dv * dv);
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
__device__ __forceinline__ void nearest_rotation(float* X) {
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
