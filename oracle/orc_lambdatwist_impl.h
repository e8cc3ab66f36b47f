/* oracle/orc_lambdatwist_impl.h -- ORACLE (test infrastructure only, see orc.h).
 * C restatement of the LambdaTwist P3P + 4th-point disambiguation used by the reference:
 *   lambdatwist/lambdatwist_p4p.h:5-62, lambdatwist_p3p.h:19-294, solve_cubic.h:13-35,154-210,
 *   solve_eig0.h:11-80, refine_lambda.h:5-102, matrix.h:615-660 (3x3 inverse).
 * Included twice with LT_T = float (reference GPU path, solve_batch_lambdatwist.cu:22) and
 * LT_T = double (reference CPU path, voldor/geometry.cpp:112).  <tgmath.h> dispatches
 * sqrt/fabs on the ARGUMENT type exactly like the std:: overloads do in the reference, and
 * the double literals (2.0, 0.5, ...) are kept where the reference has them, so the
 * float instantiation reproduces the reference's mixed float/double arithmetic.
 * Pinned against oracle/_ref (the reference header compiled in place). */

#ifndef LT_T
#error "define LT_T and LT_NAME before including"
#endif

typedef struct { LT_T v[3]; } LT_NAME(vec3);

static inline LT_T LT_NAME(dot3)(const LT_T* a, const LT_T* b) {
    LT_T s = (LT_T)0; /* matrix.h:846-851 */
    for (int i = 0; i < 3; i++) s += a[i] * b[i];
    return s;
}
static inline void LT_NAME(normalize3)(LT_T* a) { /* matrix.h:944-947, :403-411 */
    LT_T n = sqrt(LT_NAME(dot3)(a, a));
    LT_T si = (LT_T)1.0 / n;
    for (int i = 0; i < 3; i++) a[i] *= si;
}
static inline void LT_NAME(cross3)(const LT_T* a, const LT_T* b, LT_T* c) { /* matrix.h:860-872 */
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static inline void LT_NAME(matmul3)(const LT_T* a, const LT_T* b, LT_T* c) { /* matrix.h:795-810 */
    for (int r = 0; r < 3; r++)
        for (int col = 0; col < 3; col++) {
            LT_T s = (LT_T)0;
            for (int i = 0; i < 3; i++) s += a[r * 3 + i] * b[i * 3 + col];
            c[r * 3 + col] = s;
        }
}
static inline void LT_NAME(inverse3)(const LT_T* a, LT_T* out) { /* matrix.h:636-656 */
    LT_T M[9];
    M[0] = a[4] * a[8] - a[5] * a[7];
    M[1] = a[2] * a[7] - a[1] * a[8];
    M[2] = a[1] * a[5] - a[2] * a[4];
    M[3] = a[5] * a[6] - a[3] * a[8];
    M[4] = a[0] * a[8] - a[2] * a[6];
    M[5] = a[2] * a[3] - a[0] * a[5];
    M[6] = a[3] * a[7] - a[4] * a[6];
    M[7] = a[1] * a[6] - a[0] * a[7];
    M[8] = a[0] * a[4] - a[1] * a[3];
    LT_T idet = (LT_T)1.0 / (a[0] * M[0] + a[1] * M[3] + a[2] * M[6]);
    for (int i = 0; i < 9; i++) out[i] = M[i] * idet;
}

/* solve_cubic.h:13-35 */
static inline int LT_NAME(root2real)(LT_T b, LT_T c, LT_T* r1, LT_T* r2) {
    LT_T v = b * b - 4.0 * c;
    if (v < 0) { *r1 = *r2 = 0.5 * b; return 0; }
    LT_T y = sqrt(v);
    if (b < 0) { *r1 = 0.5 * (-b + y); *r2 = 0.5 * (-b - y); }
    else { *r1 = 2.0 * c / (-b + y); *r2 = 2.0 * c / (-b - y); }
    return 1;
}

/* solve_cubic.h:154-210 ; numeric limit :87-108 (float 1e-7, double 1e-13) */
static inline LT_T LT_NAME(cubick)(LT_T b, LT_T c, LT_T d) {
    LT_T r0;
    if (b * b >= 3.0 * c) {
        LT_T v = sqrt(b * b - 3.0 * c);
        LT_T t1 = (-b - v) / (3.0);
        LT_T k = ((t1 + b) * t1 + c) * t1 + d;
        if (k > 0.0) {
            r0 = t1 - sqrt(-k / (3.0 * t1 + b));
        } else {
            LT_T t2 = (-b + v) / (3.0);
            k = ((t2 + b) * t2 + c) * t2 + d;
            r0 = t2 + sqrt(-k / (3.0 * t2 + b));
        }
    } else {
        r0 = -b / 3.0;
        if (fabs((((LT_T)3.0 * r0 + (LT_T)2.0 * b) * r0 + c)) < 1e-4) r0 += 1;
    }
    LT_T fx, fpx;
    for (unsigned cnt = 0; cnt < 50; ++cnt) {
        fx = (((r0 + b) * r0 + c) * r0 + d);
        if ((cnt < 7 || fabs(fx) > LT_NUMERIC_LIMIT)) {
            fpx = (((LT_T)3.0 * r0 + (LT_T)2.0 * b) * r0 + c);
            r0 -= fx / fpx;
        } else
            break;
    }
    return r0;
}

/* solve_eig0.h:11-80 ; x row-major 3x3, E row-major, L[3] */
static inline void LT_NAME(eigwithknown0)(const LT_T* x, LT_T* E, LT_T* L) {
    L[2] = 0;
    LT_T v3[3] = { x[3] * x[7] - x[6] * x[4], x[6] * x[1] - x[7] * x[0], x[4] * x[0] - x[3] * x[1] };
    LT_NAME(normalize3)(v3);
    LT_T x01_squared = x[1] * x[1];
    LT_T b = -x[0] - x[4] - x[8];
    LT_T c = -x01_squared - x[2] * x[2] - x[5] * x[5] + x[0] * (x[4] + x[8]) + x[4] * x[8];
    LT_T e1, e2;
    LT_NAME(root2real)(b, c, &e1, &e2);
    if (fabs(e1) < fabs(e2)) { LT_T t = e1; e1 = e2; e2 = t; }
    L[0] = e1; L[1] = e2;
    LT_T mx0011 = -x[0] * x[4];
    LT_T prec_0 = x[1] * x[5] - x[2] * x[4];
    LT_T prec_1 = x[1] * x[2] - x[0] * x[5];
    LT_T e = e1;
    LT_T tmp = 1.0 / (e * (x[0] + x[4]) + mx0011 - e * e + x01_squared);
    LT_T a1 = -(e * x[2] + prec_0) * tmp;
    LT_T a2 = -(e * x[5] + prec_1) * tmp;
    LT_T rnorm = ((LT_T)1.0) / sqrt(a1 * a1 + a2 * a2 + 1.0);
    a1 *= rnorm; a2 *= rnorm;
    LT_T tmp2 = 1.0 / (e2 * (x[0] + x[4]) + mx0011 - e2 * e2 + x01_squared);
    LT_T a21 = -(e2 * x[2] + prec_0) * tmp2;
    LT_T a22 = -(e2 * x[5] + prec_1) * tmp2;
    LT_T rnorm2 = 1.0 / sqrt(a21 * a21 + a22 * a22 + 1.0);
    a21 *= rnorm2; a22 *= rnorm2;
    E[0] = a1;    E[1] = a21;    E[2] = v3[0];
    E[3] = a2;    E[4] = a22;    E[5] = v3[1];
    E[6] = rnorm; E[7] = rnorm2; E[8] = v3[2];
}

/* refine_lambda.h:5-102 (5 iterations) */
static inline void LT_NAME(gauss_newton_refineL)(LT_T* L, LT_T a12, LT_T a13, LT_T a23, LT_T b12,
                                                 LT_T b13, LT_T b23, int iterations) {
    for (int i = 0; i < iterations; ++i) {
        LT_T l1 = L[0], l2 = L[1], l3 = L[2];
        LT_T r1 = l1 * l1 + l2 * l2 + b12 * l1 * l2 - a12;
        LT_T r2 = l1 * l1 + l3 * l3 + b13 * l1 * l3 - a13;
        LT_T r3 = l2 * l2 + l3 * l3 + b23 * l2 * l3 - a23;
        if (fabs(r1) + fabs(r2) + fabs(r3) < 1e-10) break;
        LT_T dr1dl1 = (2.0) * l1 + b12 * l2;
        LT_T dr1dl2 = (2.0) * l2 + b12 * l1;
        LT_T dr2dl1 = (2.0) * l1 + b13 * l3;
        LT_T dr2dl3 = (2.0) * l3 + b13 * l1;
        LT_T dr3dl2 = (2.0) * l2 + b23 * l3;
        LT_T dr3dl3 = (2.0) * l3 + b23 * l2;
        LT_T r[3] = { r1, r2, r3 };
        LT_T v0 = dr1dl1, v1 = dr1dl2, v3 = dr2dl1, v5 = dr2dl3, v7 = dr3dl2, v8 = dr3dl3;
        LT_T det = (1.0) / (-v0 * v5 * v7 - v1 * v3 * v8);
        LT_T Ji[9] = { -v5 * v7, -v1 * v8, v1 * v5, -v3 * v8, v0 * v8, -v0 * v5, v3 * v7, -v0 * v7, -v1 * v3 };
        LT_T L1[3];
        for (int k = 0; k < 3; k++) {
            LT_T s = (LT_T)0;
            for (int j = 0; j < 3; j++) s += Ji[k * 3 + j] * r[j];
            L1[k] = L[k] - s * det; /* operator*(T s, Matrix): a(i)*s, matrix.h:1217-1224 */
        }
        {
            LT_T m1 = L1[0], m2 = L1[1], m3 = L1[2];
            LT_T r11 = m1 * m1 + m2 * m2 + b12 * m1 * m2 - a12;
            LT_T r12 = m1 * m1 + m3 * m3 + b13 * m1 * m3 - a13;
            LT_T r13 = m2 * m2 + m3 * m3 + b23 * m2 * m3 - a23;
            if (fabs(r11) + fabs(r12) + fabs(r13) > fabs(r1) + fabs(r2) + fabs(r3)) break;
            L[0] = L1[0]; L[1] = L1[1]; L[2] = L1[2];
        }
    }
}

/* lambdatwist_p3p.h:19-294.  Rs: up to 4 row-major 3x3, Ts: up to 4 vec3. returns n valid */
static inline int LT_NAME(p3p)(LT_T* y1, LT_T* y2, LT_T* y3, const LT_T* x1, const LT_T* x2,
                               const LT_T* x3, LT_T (*Rs)[9], LT_T (*Ts)[3]) {
    LT_NAME(normalize3)(y1); LT_NAME(normalize3)(y2); LT_NAME(normalize3)(y3);
    LT_T b12 = -2.0 * (LT_NAME(dot3)(y1, y2));
    LT_T b13 = -2.0 * (LT_NAME(dot3)(y1, y3));
    LT_T b23 = -2.0 * (LT_NAME(dot3)(y2, y3));
    LT_T d12[3], d13[3], d23[3], d12xd13[3];
    for (int i = 0; i < 3; i++) { d12[i] = x1[i] - x2[i]; d13[i] = x1[i] - x3[i]; d23[i] = x2[i] - x3[i]; }
    LT_NAME(cross3)(d12, d13, d12xd13);
    LT_T a12 = LT_NAME(dot3)(d12, d12), a13 = LT_NAME(dot3)(d13, d13), a23 = LT_NAME(dot3)(d23, d23);
    LT_T c31 = -0.5 * b13, c23 = -0.5 * b23, c12 = -0.5 * b12;
    LT_T blob = (c12 * c23 * c31 - 1.0);
    LT_T s31_squared = 1.0 - c31 * c31;
    LT_T s23_squared = 1.0 - c23 * c23;
    LT_T s12_squared = 1.0 - c12 * c12;
    LT_T p3 = (a13 * (a23 * s31_squared - a13 * s23_squared));
    LT_T p2 = 2.0 * blob * a23 * a13 + a13 * (2.0 * a12 + a13) * s23_squared + a23 * (a23 - a12) * s31_squared;
    LT_T p1 = a23 * (a13 - a23) * s12_squared - a12 * a12 * s23_squared - 2.0 * a12 * (blob * a23 + a13 * s23_squared);
    LT_T p0 = a12 * (a12 * s23_squared - a23 * s12_squared);
    LT_T g = 0;
    {
        p3 = 1.0 / p3;
        p2 *= p3; p1 *= p3; p0 *= p3;
        g = LT_NAME(cubick)(p2, p1, p0);
    }
    LT_T A00 = a23 * (1.0 - g);
    LT_T A01 = (a23 * b12) * 0.5;
    LT_T A02 = (a23 * b13 * g) * (-0.5);
    LT_T A11 = a23 - a12 + a13 * g;
    LT_T A12 = b23 * (a13 * g - a12) * 0.5;
    LT_T A22 = g * (a13 - a23) - a12;
    LT_T A[9] = { A00, A01, A02, A01, A11, A12, A02, A12, A22 };
    LT_T V[9], L[3];
    LT_NAME(eigwithknown0)(A, V, L);
    LT_T v = sqrt(-L[1] / L[0] > 0 ? -L[1] / L[0] : (LT_T)0);

    int valid = 0;
    LT_T Ls[4][3];
    for (int blk = 0; blk < 2; blk++) {
        LT_T s = blk == 0 ? v : -v;
        LT_T w2 = (LT_T)1.0 / (s * V[1] - V[0]);
        LT_T w0 = (V[3] - s * V[4]) * w2;
        LT_T w1 = (V[6] - s * V[7]) * w2;
        LT_T a = (LT_T)1.0 / ((a13 - a12) * w1 * w1 - a12 * b13 * w1 - a12);
        LT_T b = (a13 * b12 * w1 - a12 * b13 * w0 - (LT_T)2.0 * w0 * w1 * (a12 - a13)) * a;
        LT_T c = ((a13 - a12) * w0 * w0 + a13 * b12 * w0 + a13) * a;
        if (b * b - 4.0 * c >= 0) {
            LT_T tau1, tau2;
            LT_NAME(root2real)(b, c, &tau1, &tau2);
            for (int k = 0; k < 2; k++) {
                LT_T tau = k == 0 ? tau1 : tau2;
                if (tau > 0) {
                    LT_T d = a23 / (tau * (b23 + tau) + (LT_T)1.0);
                    /* the +v block has no d>0 test (lambdatwist_p3p.h:155-187), the -v block
                     * has one (:206-235) */
                    if (blk == 0 || d > 0) {
                        LT_T l2 = sqrt(d);
                        LT_T l3 = tau * l2;
                        LT_T l1 = w0 * l2 + w1 * l3;
                        if (l1 >= 0) { Ls[valid][0] = l1; Ls[valid][1] = l2; Ls[valid][2] = l3; ++valid; }
                    }
                }
            }
        }
    }
    for (int i = 0; i < valid; ++i)
        LT_NAME(gauss_newton_refineL)(Ls[i], a12, a13, a23, b12, b13, b23, 5);

    LT_T X[9] = { d12[0], d13[0], d12xd13[0], d12[1], d13[1], d12xd13[1], d12[2], d13[2], d12xd13[2] };
    LT_T Xi[9];
    LT_NAME(inverse3)(X, Xi);
    for (int i = 0; i < valid; ++i) {
        LT_T ry1[3], ry2[3], ry3[3], yd1[3], yd2[3], yd1xd2[3];
        for (int k = 0; k < 3; k++) { ry1[k] = y1[k] * Ls[i][0]; ry2[k] = y2[k] * Ls[i][1]; ry3[k] = y3[k] * Ls[i][2]; }
        for (int k = 0; k < 3; k++) { yd1[k] = ry1[k] - ry2[k]; yd2[k] = ry1[k] - ry3[k]; }
        LT_NAME(cross3)(yd1, yd2, yd1xd2);
        LT_T Y[9] = { yd1[0], yd2[0], yd1xd2[0], yd1[1], yd2[1], yd1xd2[1], yd1[2], yd2[2], yd1xd2[2] };
        LT_NAME(matmul3)(Y, Xi, Rs[i]);
        for (int r = 0; r < 3; r++) {
            LT_T sx = (LT_T)0;
            for (int k = 0; k < 3; k++) sx += Rs[i][r * 3 + k] * x1[k];
            Ts[i][r] = ry1[r] - sx;
        }
    }
    return valid;
}

/* lambdatwist_p4p.h:5-62 ; y: 4x2 pixels, x: 4x3 points (float), outputs float */
static int LT_NAME(p4p)(const float* y, const float* x, float fx, float fy, float cx, float cy,
                        float* R9, float* t3) {
    LT_T vy1[3] = { (LT_T)((y[0] - cx) / fx), (LT_T)((y[1] - cy) / fy), (LT_T)1.0 };
    LT_T vy2[3] = { (LT_T)((y[2] - cx) / fx), (LT_T)((y[3] - cy) / fy), (LT_T)1.0 };
    LT_T vy3[3] = { (LT_T)((y[4] - cx) / fx), (LT_T)((y[5] - cy) / fy), (LT_T)1.0 };
    LT_T vx1[3] = { (LT_T)x[0], (LT_T)x[1], (LT_T)x[2] };
    LT_T vx2[3] = { (LT_T)x[3], (LT_T)x[4], (LT_T)x[5] };
    LT_T vx3[3] = { (LT_T)x[6], (LT_T)x[7], (LT_T)x[8] };
    LT_T Rs[4][9], Ts[4][3];
    int n = LT_NAME(p3p)(vy1, vy2, vy3, vx1, vx2, vx3, Rs, Ts);
    if (n == 0) return 0;
    const float* x4 = x + 9; const float* y4 = y + 6;
    int ns = 0; LT_T min_reproj = 0;
    for (int i = 0; i < n; i++) {
        LT_T X3p = Rs[i][0] * x4[0] + Rs[i][1] * x4[1] + Rs[i][2] * x4[2] + Ts[i][0];
        LT_T Y3p = Rs[i][3] * x4[0] + Rs[i][4] * x4[1] + Rs[i][5] * x4[2] + Ts[i][1];
        LT_T Z3p = Rs[i][6] * x4[0] + Rs[i][7] * x4[1] + Rs[i][8] * x4[2] + Ts[i][2];
        LT_T mu3p = cx + fx * X3p / Z3p;
        LT_T mv3p = cy + fy * Y3p / Z3p;
        LT_T reproj = (mu3p - y4[0]) * (mu3p - y4[0]) + (mv3p - y4[1]) * (mv3p - y4[1]);
        if (i == 0 || min_reproj > reproj) { ns = i; min_reproj = reproj; }
    }
    for (int k = 0; k < 9; k++) R9[k] = (float)Rs[ns][k];
    for (int k = 0; k < 3; k++) t3[k] = (float)Ts[ns][k];
    return 1;
}
