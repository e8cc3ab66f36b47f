/* voldor_amd/csrc/vk_ref_svd.h -- the reference's projection of a 3x3 matrix onto SO(3), to the bit.
 *
 * gpu-kernels/rodrigues.h:82-108 orthonormalises every P3P rotation as U * V^T of svd3_cuda.h:36-1044 (McAdams, Selle, Tamstorf,
 * Teran, Sifakis: "Computing the SVD of 3x3 matrices with minimal branching and elementary floating point operations", TR1690):
 * an APPROXIMATE fp32 SVD -- four fixed cyclic Jacobi sweeps on A^T A with the rotation carried as a quaternion, a 1.5-step
 * reciprocal square root, column sort, three Givens rotations for Q R.  The default pipeline (product and oracle) takes the exact
 * polar factor instead (deviation D8, 2e-5 from the reference at the 99th percentile).  `--reference_svd 1` (strict mode) evaluates
 * THIS file: the same algorithm in the reference's rounding sequence, so that a strict HIP window equals the reference pipeline's
 * strict window in every bit with no oracle in between (tests/test_gpu_vs_ref_window.py::test_strict_window_equals_the_reference).
 *
 * What fixes the bits (all of it read off svd3_cuda.h; none of its text is used): every product and every sum is rounded on its
 * own (the reference guards its sums with __fadd_rn / __fsub_rn, and the CPU build of it is compiled with contraction off); the
 * association order of each sum; `__frsqrt_rn(x)` as the CPU execution of the reference defines it, (float)(1.0 / sqrt((double)x))
 * (oracle/ref_stubs/cuda_stub_common.h:22; CUDA: IEEE 1/sqrt); selection by comparison exactly where the reference builds bit masks
 * (`>=` / `<=` / `<` false on NaN); std::max semantics for the pivot magnitude.  The three Jacobi conjugations of a sweep and the
 * three Givens rotations are one routine each here, applied to cyclically renamed entries (the reference unrolls them).
 *
 * Plain C: the oracle (gcc -std=gnu11 -ffp-contract=off) includes this header too -- one restatement, pinned against the
 * reference's own rodrigues() compiled in place (tests/golden/ref_rodrigues.npz, tests/test_oracle_vs_golden.py).
 */
#ifndef VK_REF_SVD_H
#define VK_REF_SVD_H

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VRS_FN __host__ __device__ static inline
#else
#include <math.h>
#define VRS_FN static inline
#endif
#if defined(__clang__)
#define VRS_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define VRS_NO_CONTRACT /* gcc: compiled with -ffp-contract=off */
#endif

VRS_FN float vrs_rsqrt(float x) { return (float)(1.0 / sqrt((double)x)); }
/* one Newton step on the reciprocal square root, in the reference's association: (r + r/2) - x * (r * (r * (r/2))) */
VRS_FN float vrs_rsqrt_refined(float x) {
    VRS_NO_CONTRACT
    const float r = vrs_rsqrt(x), h = r * 0.5f;
    float t = r * h;
    t = r * t;
    t = x * t;
    const float u = r + h;
    return u - t;
}
VRS_FN float vrs_max(float a, float b) { return (a < b) ? b : a; } /* std::max */
VRS_FN float vrs_from_bits(unsigned int u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

/* One Jacobi conjugation of the symmetric S = A^T A in the (p,q) plane, r the third index (svd3_cuda.h:108-213 for (1,2),
 * :218-318 for (2,3), :324-424 for (3,1)).  ra = S[r][p]-ish off-diagonal that receives +s*rb, rb the one that receives -s*ra;
 * q4 = (qs, q_r, q_p, q_q): scalar part and the vector parts along r, p, q. */
VRS_FN void vrs_jacobi_conjugate(float* pp, float* qq, float* pq, float* ra, float* rb, float* rr, float* qs, float* q_r, float* q_p, float* q_q) {
    VRS_NO_CONTRACT
    /* approximate Givens half-angle: (ch, sh) ~ (pp - qq, pq / 2), or the pi/8 rotation when that angle would exceed pi/4 */
    float sh = *pq * 0.5f;
    float d = *pp - *qq;
    float t2 = sh * sh;
    const int usable = t2 >= 1.e-20f;
    sh = usable ? sh : 0.f;
    float ch = usable ? d : 1.f;
    float t1 = sh * sh;
    t2 = ch * ch;
    float t3 = t1 + t2;
    const float w = vrs_rsqrt(t3);
    sh = w * sh;
    ch = w * ch;
    t1 = 5.8284273147583007813f * t1; /* 4 gamma^2 = 3 + 2 sqrt 2 */
    const int clamp = t2 <= t1;
    if (clamp) { sh = vrs_from_bits(1053028117u); ch = vrs_from_bits(1064076127u); } /* the reference's sin, cos of pi/8 (its cosine is one ulp above the nearest float) */
    t1 = sh * sh;
    t2 = ch * ch;
    const float c = t2 - t1;
    float s = ch * sh;
    s = s + s;
    /* conjugation; (ch, sh) is not unit, so the untouched part is scaled by (ch^2 + sh^2) per side */
    t3 = t1 + t2;
    *rr = *rr * t3; *ra = *ra * t3; *rb = *rb * t3; *rr = *rr * t3;
    t1 = s * *ra; t2 = s * *rb;
    *ra = c * *ra; *rb = c * *rb;
    *ra = t2 + *ra; *rb = *rb - t1;
    t2 = s * s;
    t1 = *qq * t2; t3 = *pp * t2;
    float t4 = c * c;
    *pp = *pp * t4; *qq = *qq * t4;
    *pp = *pp + t1; *qq = *qq + t3;
    t4 = t4 - t2;
    t2 = *pq + *pq;
    *pq = *pq * t4;
    t4 = c * s;
    t2 = t2 * t4;
    d = d * t4;
    *pp = *pp + t2; *pq = *pq - d; *qq = *qq - t2;
    /* accumulate the rotation about axis r into the quaternion */
    t1 = sh * *q_p; t2 = sh * *q_q; t3 = sh * *q_r;
    sh = sh * *qs;
    *qs = ch * *qs; *q_p = ch * *q_p; *q_q = ch * *q_q; *q_r = ch * *q_r;
    *q_r = *q_r + sh; *qs = *qs - t3; *q_p = *q_p + t2; *q_q = *q_q - t1;
}

/* conditional exchange of columns i, j of B and V when |col i|^2 < |col j|^2; the column named `neg` is then negated (through a
 * multiplication by 1 + (-2 or 0), as the reference does) so that V stays a rotation (svd3_cuda.h:545-705) */
VRS_FN void vrs_sort_pair(float B[3][3], float V[3][3], float* ni, float* nj, int i, int j, int neg) {
    VRS_NO_CONTRACT
    const int sw = *ni < *nj;
    if (sw) {
        for (int r = 0; r < 3; r++) { float t = B[r][i]; B[r][i] = B[r][j]; B[r][j] = t; t = V[r][i]; V[r][i] = V[r][j]; V[r][j] = t; }
        const float t = *ni; *ni = *nj; *nj = t;
    }
    const float f = 1.f + (sw ? -2.f : 0.f);
    for (int r = 0; r < 3; r++) { B[r][neg] = B[r][neg] * f; V[r][neg] = V[r][neg] * f; }
}

/* Givens rotation of rows p, q of B that annihilates B[q][col], accumulated into columns p, q of U (svd3_cuda.h:717-806 for
 * (1,2) on column 1, :809-898 for (1,3) on column 1, :902-991 for (2,3) on column 2) */
VRS_FN void vrs_givens(float B[3][3], float U[3][3], int p, int q, int col) {
    VRS_NO_CONTRACT
    const float apiv = B[p][col], azero = B[q][col];
    float sh = azero * azero;
    sh = (sh >= 1.e-12f) ? azero : 0.f;
    float ch = 0.f - apiv;
    ch = vrs_max(ch, apiv);
    ch = vrs_max(ch, 1.e-12f);
    const int nonneg = apiv >= 0.f;
    float t1 = ch * ch, t2 = sh * sh;
    t2 = t1 + t2;
    t1 = vrs_rsqrt_refined(t2);
    t1 = t1 * t2; /* ~ sqrt(ch^2 + sh^2) */
    ch = ch + t1;
    if (!nonneg) { const float t = ch; ch = sh; sh = t; }
    t1 = ch * ch; t2 = sh * sh;
    t2 = t1 + t2;
    t1 = vrs_rsqrt_refined(t2);
    ch = ch * t1; sh = sh * t1;
    float c = ch * ch, s = sh * sh;
    c = c - s;
    s = sh * ch;
    s = s + s;
    for (int k = 0; k < 3; k++) { /* rows p, q of B */
        const float a = s * B[p][k], b = s * B[q][k];
        B[p][k] = c * B[p][k]; B[q][k] = c * B[q][k];
        B[p][k] = B[p][k] + b; B[q][k] = B[q][k] - a;
    }
    for (int k = 0; k < 3; k++) { /* columns p, q of U */
        const float a = s * U[k][p], b = s * U[k][q];
        U[k][p] = c * U[k][p]; U[k][q] = c * U[k][q];
        U[k][p] = U[k][p] + b; U[k][q] = U[k][q] - a;
    }
}

/* A = U diag(sigma) V^T as the reference computes it.  A, U, V row-major [row][column]. */
VRS_FN void vrs_svd3(const float A[3][3], float U[3][3], float sigma[3], float V[3][3]) {
    VRS_NO_CONTRACT
    /* lower triangle of S = A^T A, each entry (a*a' + b*b') + c*c' down the rows */
    float S[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j <= i; j++) {
            float acc = A[0][i] * A[0][j];
            float t = A[1][i] * A[1][j];
            acc = t + acc;
            t = A[2][i] * A[2][j];
            acc = t + acc;
            S[i][j] = acc;
        }
    float qs = 1.f, qv[3] = { 0.f, 0.f, 0.f };
    for (int sweep = 0; sweep < 4; sweep++) {
        vrs_jacobi_conjugate(&S[0][0], &S[1][1], &S[1][0], &S[2][0], &S[2][1], &S[2][2], &qs, &qv[2], &qv[0], &qv[1]);
        vrs_jacobi_conjugate(&S[1][1], &S[2][2], &S[2][1], &S[1][0], &S[2][0], &S[0][0], &qs, &qv[0], &qv[1], &qv[2]);
        vrs_jacobi_conjugate(&S[2][2], &S[0][0], &S[2][0], &S[2][1], &S[1][0], &S[1][1], &qs, &qv[1], &qv[2], &qv[0]);
    }
    /* unit quaternion -> V */
    float n = qs * qs;
    float t = qv[0] * qv[0];
    n = t + n;
    t = qv[1] * qv[1];
    n = t + n;
    t = qv[2] * qv[2];
    n = t + n;
    const float inv = vrs_rsqrt_refined(n);
    qs = qs * inv; qv[0] = qv[0] * inv; qv[1] = qv[1] * inv; qv[2] = qv[2] * inv;
    {
        const float xx = qv[0] * qv[0], yy = qv[1] * qv[1], zz = qv[2] * qv[2], ss = qs * qs;
        float v22 = ss - xx;
        float v33 = v22 - yy;
        v33 = v33 + zz;
        v22 = v22 + yy;
        v22 = v22 - zz;
        float v11 = ss + xx;
        v11 = v11 - yy;
        v11 = v11 - zz;
        const float x2 = qv[0] + qv[0], y2 = qv[1] + qv[1], z2 = qv[2] + qv[2];
        const float sx = qs * x2, sy = qs * y2, sz = qs * z2;
        const float xy = qv[1] * x2, yz = qv[2] * y2, zx = qv[0] * z2;
        V[0][0] = v11;     V[0][1] = xy - sz; V[0][2] = zx + sy;
        V[1][0] = xy + sz; V[1][1] = v22;     V[1][2] = yz - sx;
        V[2][0] = zx - sy; V[2][1] = yz + sx; V[2][2] = v33;
    }
    /* B = A V, each entry (v1j*a_i1 + v2j*a_i2) + v3j*a_i3 */
    float B[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float acc = V[0][j] * A[i][0];
            float u = V[1][j] * A[i][1];
            acc = acc + u;
            u = V[2][j] * A[i][2];
            acc = acc + u;
            B[i][j] = acc;
        }
    /* sort the columns by decreasing norm */
    float nn[3];
    for (int j = 0; j < 3; j++) {
        float acc = B[0][j] * B[0][j];
        float u = B[1][j] * B[1][j];
        acc = acc + u;
        u = B[2][j] * B[2][j];
        acc = acc + u;
        nn[j] = acc;
    }
    vrs_sort_pair(B, V, &nn[0], &nn[1], 0, 1, 1);
    vrs_sort_pair(B, V, &nn[0], &nn[2], 0, 2, 0);
    vrs_sort_pair(B, V, &nn[1], &nn[2], 1, 2, 2);
    /* B = U R by three Givens rotations */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) U[i][j] = (i == j) ? 1.f : 0.f;
    vrs_givens(B, U, 0, 1, 0);
    vrs_givens(B, U, 0, 2, 0);
    vrs_givens(B, U, 1, 2, 1);
    sigma[0] = B[0][0]; sigma[1] = B[1][1]; sigma[2] = B[2][2];
}

/* rodrigues.h:82-108: R <- U V^T, every entry (u_i1 v_j1 + u_i2 v_j2) + u_i3 v_j3 */
VRS_FN void vrs_project_rotation(float* R9) {
    VRS_NO_CONTRACT
    float A[3][3], U[3][3], V[3][3], sg[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i][j] = R9[i * 3 + j];
    vrs_svd3(A, U, sg, V);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float acc = U[i][0] * V[j][0];
            float u = U[i][1] * V[j][1];
            acc = acc + u;
            u = U[i][2] * V[j][2];
            acc = acc + u;
            R9[i * 3 + j] = acc;
        }
}
#endif
