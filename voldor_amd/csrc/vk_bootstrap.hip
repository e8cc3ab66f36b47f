// voldor_amd/csrc/vk_bootstrap.hip -- monocular bootstrap (voldor/voldor.cpp:151-162):
// estimate_camera_pose_epipolar + estimate_depth_closed_form (voldor/geometry.cpp:267-332).
//
// The reference calls OpenCV 3.4 findEssentialMat(LMEDS, 0.999, 1.0) + recoverPose on 19 200
// stride-4 correspondences, on the CPU (~10 ms).  OpenCV is not part of the reference tree; the
// same LMedS principle is implemented here with a normalised 8-point minimal solver, 256
// hypotheses, the median squared Sampson distance over a <=2048-point scoring subset, a cheirality
// vote, then cam.t = R*t (geometry.cpp:330).  (Deviation D5 in DESIGN.md)
//
// Everything runs on the GPU, in the window's stream, with no host round trip:
//   k_extract_corr   stride-4 correspondences
//   k_boot_hyp       one lane per hypothesis: 8x9 constraint matrix -> null vector by complete-pivoting
//                    elimination (matrices in LDS, [element][lane] so a wave touches consecutive banks)
//                    -> rank-2 projection (3x3 Jacobi)
//   k_boot_score     one workgroup per hypothesis: Sampson distances -> bitonic sort in LDS -> median
//   k_boot_select    argmin, E -> (R,t) candidates, cheirality vote, pose -> PoseBlock / CamState
//   k_depth_closed_form
// The arithmetic is fp64 with no fma contraction and is shared (same source, __host__ __device__)
// with the host entry point vk_estimate_pose_epipolar, so both produce identical bits.
#include "vk_common.hpp"
#include "vk_device.hpp"
#include "vk_internal.hpp"
#include "vk_p3p.hpp"
#include "vk_fivept.hpp"
#include "vk_ref_cv.h"
#include "../../include/voldor_hip.h"
#include <vector>
#include <algorithm>
#include <cmath>

namespace vk {

constexpr int BOOT_HYPS = 256, BOOT_SCORE_MAX = 2048, BOOT_STEP = 4;
// --bootstrap_points 5: the five-point minimal solver (vk_fivept.hpp), the solver behind the reference's cv::findEssentialMat(.., LMEDS,
// 0.999, 1.0) (geometry.cpp:316-326).  OpenCV's LMedS draws round(log(1 - 0.999) / log(1 - (1 - 0.45)^5)) = 134 samples; every sample yields up
// to ten essential matrices and each one is a model of its own (median squared Sampson distance).  192 samples = three waves of lanes.
// five-point LMedS: the number of subsets cv::findEssentialMat(..., LMEDS, 0.999, ...) draws (voldor/geometry.cpp:316-318).  OpenCV 3.4's LMedS registrator fixes it
// before the loop: RANSACUpdateNumIters(confidence 0.999, outlier ratio 0.45, 5 model points, 1000) = round(log(1 - 0.999) / log(1 - 0.55^5)) = 134 (modules/calib3d/
// src/ptsetreg.cpp; restated from the published source, OpenCV is not in the tree).  Up to ten models per subset.
constexpr int BOOT5_SAMPLES = 134, BOOT5_MODELS = BOOT5_SAMPLES * 10;
constexpr int BOOT_MAX_MODELS = BOOT5_MODELS > BOOT_HYPS ? BOOT5_MODELS : BOOT_HYPS;

// ---- shared host/device numerics --------------------------------------------------------------
// Cyclic Jacobi for a symmetric NxN matrix stored with element stride S (A[(i*N+j)*S]):
// A = V diag(e) V^T, eigenvalues ascending.
template <int N, int S>
VK_HD void sym_eig(double* A, double* V, double* e) {
#define AA(i, j) A[((i) * N + (j)) * S]
#define VV(i, j) V[((i) * N + (j)) * S]
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) VV(i, j) = (i == j) ? 1.0 : 0.0;
#pragma unroll 1
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int i = 0; i < N; i++) for (int j = i + 1; j < N; j++) off += AA(i, j) * AA(i, j);
        if (off < 1e-30) break;
#pragma unroll 1
        for (int p = 0; p < N; p++)
#pragma unroll 1
            for (int q = p + 1; q < N; q++) {
                const double apq = AA(p, q);
                if (vk_abs(apq) < 1e-300) continue;
                const double theta = (AA(q, q) - AA(p, p)) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (vk_abs(theta) + vk_sqrt(theta * theta + 1));
                const double c = 1 / vk_sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < N; k++) { double a = AA(k, p), b = AA(k, q); AA(k, p) = c * a - s * b; AA(k, q) = s * a + c * b; }
                for (int k = 0; k < N; k++) { double a = AA(p, k), b = AA(q, k); AA(p, k) = c * a - s * b; AA(q, k) = s * a + c * b; }
                for (int k = 0; k < N; k++) { double a = VV(k, p), b = VV(k, q); VV(k, p) = c * a - s * b; VV(k, q) = s * a + c * b; }
            }
    }
    for (int i = 0; i < N; i++) e[i] = AA(i, i);
    for (int i = 0; i < N; i++) {
        int m = i;
        for (int j = i + 1; j < N; j++) if (e[j] < e[m]) m = j;
        if (m != i) {
            double t = e[i]; e[i] = e[m]; e[m] = t;
            for (int k = 0; k < N; k++) { t = VV(k, i); VV(k, i) = VV(k, m); VV(k, m) = t; }
        }
    }
#undef AA
#undef VV
}
VK_HD inline double det3x3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
// E = U diag(s) V^T with det U = det V = +1 (rank >= 2)
VK_HD inline void svd_3x3(const double* E, double* U, double* s, double* V) {
    double EtE[9], ev[3], Vv[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double a = 0; for (int k = 0; k < 3; k++) a += E[k * 3 + i] * E[k * 3 + j]; EtE[i * 3 + j] = a; }
    sym_eig<3, 1>(EtE, Vv, ev);
    for (int c = 0; c < 3; c++) {
        for (int r = 0; r < 3; r++) V[r * 3 + c] = Vv[r * 3 + (2 - c)];
        s[c] = vk_sqrt(ev[2 - c] > 0 ? ev[2 - c] : 0.0);
    }
    if (det3x3(V) < 0) for (int r = 0; r < 3; r++) V[r * 3 + 2] = -V[r * 3 + 2];
    double u[3][3];
    for (int c = 0; c < 2; c++) {
        double n = 0;
        for (int r = 0; r < 3; r++) { u[c][r] = E[r * 3] * V[c] + E[r * 3 + 1] * V[3 + c] + E[r * 3 + 2] * V[6 + c]; n += u[c][r] * u[c][r]; }
        n = vk_sqrt(n); if (n < 1e-300) n = 1;
        for (int r = 0; r < 3; r++) u[c][r] /= n;
    }
    {
        double d = u[0][0] * u[1][0] + u[0][1] * u[1][1] + u[0][2] * u[1][2], n = 0;
        for (int r = 0; r < 3; r++) { u[1][r] -= d * u[0][r]; n += u[1][r] * u[1][r]; }
        n = vk_sqrt(n); if (n < 1e-300) n = 1;
        for (int r = 0; r < 3; r++) u[1][r] /= n;
    }
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) U[r * 3 + c] = u[c][r];
}

struct BootGeom { int nx, ny, n, step, sstride, ns; double fx, fy, cx, cy; };
VK_HD inline BootGeom boot_geom(int w, int h, float fx, float fy, float cx, float cy) {
    BootGeom g;
    g.step = BOOT_STEP; g.nx = (w + g.step - 1) / g.step; g.ny = (h + g.step - 1) / g.step; g.n = g.nx * g.ny;
    g.sstride = (g.n + BOOT_SCORE_MAX - 1) / BOOT_SCORE_MAX; g.ns = (g.n + g.sstride - 1) / g.sstride;
    g.fx = fx; g.fy = fy; g.cx = cx; g.cy = cy;
    return g;
}
// normalised coordinates of correspondence i: q1 from the pixel grid, q2 from pixel + flow (float sum)
VK_HD inline void boot_corr(const BootGeom& g, const float* p2, int i, double* q1, double* q2) {
    const int x = (i % g.nx) * g.step, y = (i / g.nx) * g.step;
    q1[0] = (x - g.cx) / g.fx; q1[1] = (y - g.cy) / g.fy;
    q2[0] = (p2[i * 2] - g.cx) / g.fx; q2[1] = (p2[i * 2 + 1] - g.cy) / g.fy;
}
// 8x9 epipolar constraint matrix of hypothesis `hy` (8 correspondences drawn with rng3), element stride S
template <int S>
VK_HD void boot_constraint_matrix(const BootGeom& g, const float* p2, int hy, double* A) {
    for (int k = 0; k < 8; k++) {
        const int i = (int)(rng3(RAND_SEED, (uint32_t)hy, 0x100u + (uint32_t)k) % (uint32_t)g.n);
        double q1[2], q2[2];
        boot_corr(g, p2, i, q1, q2);
        const double a[9] = { q2[0] * q1[0], q2[0] * q1[1], q2[0], q2[1] * q1[0], q2[1] * q1[1], q2[1], q1[0], q1[1], 1.0 };
        for (int c = 0; c < 9; c++) A[(k * 9 + c) * S] = a[c];
    }
}
// Null vector of the 8x9 matrix by Gaussian elimination with complete pivoting (double): after 8 pivots the remaining
// column is free, set to 1, the rest follows by back substitution.  ~600 operations instead of the ~30k of a cyclic
// Jacobi eigen-decomposition of A^T A; a minimal sample has an exact null space, so it is the same vector.
template <int S>
VK_HD void boot_null9(double* A, double* f) {
#define AA(r, c) A[((r) * 9 + (c)) * S]
    int perm[9];
    for (int c = 0; c < 9; c++) perm[c] = c;
    // On the device the matrix lives in LDS (run-time row / column indices) and one lane walks it alone: what the step costs is the
    // number of DEPENDENT LDS round trips, so the loops over the trailing block are written with constant bounds and a predicate --
    // all reads of a step are issued together -- in the same order as the plain r-from-k, c-from-k loops (first maximum wins).
#pragma unroll 1
    for (int k = 0; k < 8; k++) {
        int pr = k, pc = k; double best = -1.0;
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int c = 0; c < 9; c++)
                if (r >= k && c >= k) { const double v = vk_abs(AA(r, c)); if (v > best) { best = v; pr = r; pc = c; } }
        if (pr != k) for (int c = 0; c < 9; c++) { const double t = AA(k, c); AA(k, c) = AA(pr, c); AA(pr, c) = t; }
        if (pc != k) {
            for (int r = 0; r < 8; r++) { const double t = AA(r, k); AA(r, k) = AA(r, pc); AA(r, pc) = t; }
            // perm[] is indexed dynamically: swap through selects so that it stays in registers on the device
            int pk = 0, ppc = 0;
#pragma unroll
            for (int c = 0; c < 9; c++) { if (c == k) pk = perm[c]; if (c == pc) ppc = perm[c]; }
#pragma unroll
            for (int c = 0; c < 9; c++) { if (c == k) perm[c] = ppc; else if (c == pc) perm[c] = pk; }
        }
        double piv = AA(k, k);
        if (!(vk_abs(piv) > 1e-300)) piv = 1e-300;  // rank deficient sample
        double rowk[9];
#pragma unroll
        for (int c = 0; c < 9; c++) rowk[c] = AA(k, c);
#pragma unroll
        for (int r = 1; r < 8; r++) {
            if (r > k) {
                const double m = AA(r, k) / piv;
#pragma unroll
                for (int c = 1; c < 9; c++)
                    if (c > k) AA(r, c) -= m * rowk[c];
            }
        }
        AA(k, k) = piv;
    }
    double x[9];
    x[8] = 1.0;
#pragma unroll
    for (int k = 7; k >= 0; k--) {
        double sacc = 0;
#pragma unroll
        for (int c = k + 1; c < 9; c++) sacc += AA(k, c) * x[c];
        x[k] = -sacc / AA(k, k);
    }
    double nrm = 0;
#pragma unroll
    for (int c = 0; c < 9; c++) nrm += x[c] * x[c];
    nrm = vk_sqrt(nrm);
#pragma unroll
    for (int c = 0; c < 9; c++) {
        const double v = x[c] / nrm;
#pragma unroll
        for (int d = 0; d < 9; d++) if (perm[c] == d) f[d] = v;
    }
#undef AA
}
// null vector -> nearest essential matrix U diag(1,1,0) V^T
template <int S>
VK_HD void boot_essential(double* A, double* E) {
    double E0[9], U[9], s[3], Vt[9];
    boot_null9<S>(A, E0);
    svd_3x3(E0, U, s, Vt);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) E[r * 3 + c] = U[r * 3] * Vt[c * 3] + U[r * 3 + 1] * Vt[c * 3 + 1];
}
VK_HD inline double boot_sampson(const double* E, const double* q1, const double* q2) {
    const double x1 = q1[0], y1 = q1[1], x2 = q2[0], y2 = q2[1];
    const double Ex0 = E[0] * x1 + E[1] * y1 + E[2], Ex1 = E[3] * x1 + E[4] * y1 + E[5], Ex2 = E[6] * x1 + E[7] * y1 + E[8];
    const double Et0 = E[0] * x2 + E[3] * y2 + E[6], Et1 = E[1] * x2 + E[4] * y2 + E[7];
    const double num = x2 * Ex0 + y2 * Ex1 + Ex2;
    return num * num / (Ex0 * Ex0 + Ex1 * Ex1 + Et0 * Et0 + Et1 * Et1);
}
// E -> two rotations and the translation direction (recoverPose decomposition)
VK_HD inline void boot_decompose(const double* E, double (*Rc)[9], double* tc) {
    double U[9], s[3], V[9];
    svd_3x3(E, U, s, V);
    const double W[9] = { 0, -1, 0, 1, 0, 0, 0, 0, 1 };
    for (int k = 0; k < 2; k++) {
        double UW[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int j = 0; j < 3; j++) a += U[r * 3 + j] * (k == 0 ? W[j * 3 + c] : W[c * 3 + j]); UW[r * 3 + c] = a; }
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int j = 0; j < 3; j++) a += UW[r * 3 + j] * V[c * 3 + j]; Rc[k][r * 3 + c] = a; }
    }
    tc[0] = U[2]; tc[1] = U[5]; tc[2] = U[8];
}
// does correspondence (q1,q2) triangulate in front of both cameras for candidate (R, sg*tc)?
VK_HD inline bool boot_in_front(const double* R, const double* tc, double sg, const double* q1, const double* q2) {
    const double t[3] = { sg * tc[0], sg * tc[1], sg * tc[2] };
    const double a[3] = { q1[0], q1[1], 1 }, b[3] = { q2[0], q2[1], 1 };
    const double Ra[3] = { R[0] * a[0] + R[1] * a[1] + R[2] * a[2], R[3] * a[0] + R[4] * a[1] + R[5] * a[2], R[6] * a[0] + R[7] * a[1] + R[8] * a[2] };
    const double A11 = Ra[0] * Ra[0] + Ra[1] * Ra[1] + Ra[2] * Ra[2], A12 = -(Ra[0] * b[0] + Ra[1] * b[1] + Ra[2] * b[2]);
    const double A22 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
    const double r1 = -(Ra[0] * t[0] + Ra[1] * t[1] + Ra[2] * t[2]), r2 = b[0] * t[0] + b[1] * t[1] + b[2] * t[2];
    const double det = A11 * A22 - A12 * A12;
    if (vk_abs(det) < 1e-12) return false;
    const double z1 = (r1 * A22 - A12 * r2) / det, z2 = (A11 * r2 - A12 * r1) / det;
    return z1 > 0 && z2 > 0;
}
// chosen candidate -> float pose with the reference's cam.t = R*t (geometry.cpp:330)
VK_HD inline void boot_finish(const double* R, const double* tc, double sg, float* R9, float* t3) {
    float Rf[9], tf[3] = { (float)(sg * tc[0]), (float)(sg * tc[1]), (float)(sg * tc[2]) };
    for (int i = 0; i < 9; i++) Rf[i] = (float)R[i];
    for (int i = 0; i < 9; i++) R9[i] = Rf[i];
    for (int r = 0; r < 3; r++) t3[r] = Rf[r * 3] * tf[0] + Rf[r * 3 + 1] * tf[1] + Rf[r * 3 + 2] * tf[2];
}
// KRKinv and b = K t in float (cv::Mat float products, geometry.cpp:269-270) for the closed-form depth
VK_HD inline void boot_depth_coeffs(const float* K9, const float* R9, const float* t3, float* Mb) {
    const float Kinv[9] = { 1.f / K9[0], 0, -K9[2] / K9[0], 0, 1.f / K9[4], -K9[5] / K9[4], 0, 0, 1 };
    float KR[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float s = 0; for (int k = 0; k < 3; k++) s += K9[i * 3 + k] * R9[k * 3 + j]; KR[i * 3 + j] = s; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float s = 0; for (int k = 0; k < 3; k++) s += KR[i * 3 + k] * Kinv[k * 3 + j]; Mb[i * 3 + j] = s; }
    for (int i = 0; i < 3; i++) Mb[9 + i] = K9[i * 3] * t3[0] + K9[i * 3 + 1] * t3[1] + K9[i * 3 + 2] * t3[2];
}

// ---- host reference of the same procedure (vk_estimate_pose_epipolar) ---------------------------
int lmeds_essential_host(const float* p2, int w, int h, float fx, float fy, float cx, float cy, float* R9, float* t3) {
    const BootGeom g = boot_geom(w, h, fx, fy, cx, cy);
    if (g.n < 8) return 0;
    std::vector<double> errs(g.ns);
    double best_med = INFINITY, bestE[9] = { 0 };
    for (int hy = 0; hy < BOOT_HYPS; hy++) {
        double A[72], E[9];
        boot_constraint_matrix<1>(g, p2, hy, A);
        boot_essential<1>(A, E);
        for (int j = 0; j < g.ns; j++) {
            double q1[2], q2[2];
            boot_corr(g, p2, j * g.sstride, q1, q2);
            double v = boot_sampson(E, q1, q2);
            errs[j] = (v == v) ? v : INFINITY;
        }
        std::nth_element(errs.begin(), errs.begin() + g.ns / 2, errs.end());  // the ns/2-th order statistic
        const double med = errs[g.ns / 2];
        if (med < best_med) { best_med = med; memcpy(bestE, E, sizeof E); }
    }
    double Rc[2][9], tc[3];
    boot_decompose(bestE, Rc, tc);
    int best = 0, best_cnt = -1;
    for (int cand = 0; cand < 4; cand++) {
        int cnt = 0;
        for (int j = 0; j < g.ns; j++) {
            double q1[2], q2[2];
            boot_corr(g, p2, j * g.sstride, q1, q2);
            if (boot_in_front(Rc[cand >> 1], tc, (cand & 1) ? -1.0 : 1.0, q1, q2)) cnt++;
        }
        if (cnt > best_cnt) { best_cnt = cnt; best = cand; }
    }
    boot_finish(Rc[best >> 1], tc, (best & 1) ? -1.0 : 1.0, R9, t3);
    return 1;
}

// the five correspondences of sample `hy` -> up to ten essential matrices (unused slots NaN)
VK_HD inline void boot5_sample(const BootGeom& g, const float* p2, int hy, double (*Es)[9]) {
    double q1[5][2], q2[5][2];
    for (int k = 0; k < 5; k++) {
        const int i = (int)(rng3(RAND_SEED, (uint32_t)hy, 0x200u + (uint32_t)k) % (uint32_t)g.n);
        boot_corr(g, p2, i, q1[k], q2[k]);
    }
    const int n = fivept::solve(q1, q2, Es);
    const double qnan = __builtin_nan("");
    for (int m = n; m < 10; m++) for (int c = 0; c < 9; c++) Es[m][c] = qnan;
}
int lmeds_essential_host5(const float* p2, int w, int h, float fx, float fy, float cx, float cy, float* R9, float* t3) {
    const BootGeom g = boot_geom(w, h, fx, fy, cx, cy);
    if (g.n < 8) return 0;
    std::vector<double> errs(g.ns);
    double best_med = INFINITY, bestE[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 0 };
    for (int hy = 0; hy < BOOT5_SAMPLES; hy++) {
        double Es[10][9];
        boot5_sample(g, p2, hy, Es);
        for (int m = 0; m < 10; m++) {
            if (!(Es[m][0] == Es[m][0])) continue;
            for (int j = 0; j < g.ns; j++) {
                double q1[2], q2[2];
                boot_corr(g, p2, j * g.sstride, q1, q2);
                double v = boot_sampson(Es[m], q1, q2);
                errs[j] = (v == v) ? v : INFINITY;
            }
            std::nth_element(errs.begin(), errs.begin() + g.ns / 2, errs.end());
            const double med = errs[g.ns / 2];
            if (med < best_med) { best_med = med; memcpy(bestE, Es[m], sizeof bestE); }  // first strict minimum in (sample, model) order
        }
    }
    double Rc[2][9], tc[3];
    boot_decompose(bestE, Rc, tc);
    int best = 0, best_cnt = -1;
    for (int cand = 0; cand < 4; cand++) {
        int cnt = 0;
        for (int j = 0; j < g.ns; j++) {
            double q1[2], q2[2];
            boot_corr(g, p2, j * g.sstride, q1, q2);
            if (boot_in_front(Rc[cand >> 1], tc, (cand & 1) ? -1.0 : 1.0, q1, q2)) cnt++;
        }
        if (cnt > best_cnt) { best_cnt = cnt; best = cand; }
    }
    boot_finish(Rc[best >> 1], tc, (best & 1) ? -1.0 : 1.0, R9, t3);
    return 1;
}

// ---- kernels --------------------------------------------------------------------------------------
__global__ static void k_extract_corr(const float2* __restrict__ flow, float* __restrict__ out, int w, int h, int step, int nx, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (i % nx) * step, y = (i / nx) * step;
    float2 f = flow[y * w + x];
    out[i * 2] = (float)x + f.x;
    out[i * 2 + 1] = (float)y + f.y;
}

// one lane per hypothesis; the 8x9 constraint matrices of 64 hypotheses in LDS, [element][lane] (rows and columns are
// swapped with run-time indices during pivoting)
__global__ __launch_bounds__(64) static void k_boot_hyp(const float* __restrict__ p2, BootGeom g, double* __restrict__ Es) {
    __shared__ double sA[72 * 64];
    const int hy = blockIdx.x * 64 + threadIdx.x;
    double* A = sA + threadIdx.x;
    double E[9];
    boot_constraint_matrix<64>(g, p2, hy, A);
    boot_essential<64>(A, E);
    for (int k = 0; k < 9; k++) Es[(size_t)hy * 9 + k] = E[k];
}

// Five-point samples (round 6: a sample over a workgroup instead of a lane).  Rounds 4-5 ran one sample per lane on private arrays -- 134-192 lanes of the chip
// walking ~10^5 dependent fp64 operations each (the arrays in scratch memory): 1.9 ms, which kept the five-point bootstrap optional.  Now one 128-thread
// workgroup per sample, wave b on basis b of the null space (vk_fivept.hpp: the system is solved in two bases), the work arrays in LDS:
//   lane 0    the five correspondences, the null space, the ten cubics, their elimination, the degree-10 polynomial (the serial part: ~5 k flops)
//   lanes < deg   the Aberth sweeps, one root per lane, the other roots through wave shuffles (registers only: a wave leaves the loop on its own)
//   lane 0    the real starting values in root order
//   lane k    starting value k: null vector of B(z), Gauss-Newton on the ten cubics, the residual test -> a model or nothing; survivors keep their order (ballot)
//   then wave 0 merges the two bases' models (duplicates dropped, ten at most) and the workgroup writes them.
// The same functions as the host build (fivept::solve), the same operations on the same operands in the same order: the host bits
// (tests/test_fivept.py::test_five_point_bootstrap_kernels_give_the_host_bits).
struct Boot5Wave {
    double NA[5][9]; int perm[9];            // the epipolar system, eliminated in place
    double N[4][9], e[9][4];                 // the basis, the entries of E as polynomials
    fivept::BasisWork W;
    double A[10][20];
    double c[11];
    fivept::Cx z[10];
    double zs[20];
    double Es[10][9];
    int deg, nz, ne, ok;
};
// the lanes of ONE wave meet: LDS instructions of a wave execute in order, so an earlier store of any lane is seen by a later load of any other; the
// fences keep the compiler from moving accesses across the meeting point (no workgroup barrier: the two waves of a sample go their own ways)
__device__ __forceinline__ void wave_meet() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// gauss_jordan_10x20 (vk_fivept.hpp) with the 200 entries over the lanes of a wave: per pivot column the first largest entry at or below the diagonal
// (a butterfly over (value, row): larger value, lower row among equals -- the serial loop's strict >), the row swap, the pivot row divided, every other
// row minus its multiple: entry by entry the operations of the serial routine on the same operands.  Uniform result over the wave.
__device__ __forceinline__ bool gauss_jordan_10x20_wave(double (*A)[20], int lane) {
    for (int k = 0; k < 10; k++) {
        const double akk = vk_abs(A[k][k]);
        if (!(akk == akk)) return false;  // (the serial search starts from it: a NaN there fails the pivot test)
        double v = (lane >= k && lane < 10) ? vk_abs(A[lane][k]) : -1.0;
        if (!(v == v)) v = -1.0;          // (a NaN further down is never "greater")
        int r = lane < 10 ? lane : 99;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const double ov = __shfl_xor(v, d, 64); const int orr = __shfl_xor(r, d, 64);
            if (ov > v || (ov == v && orr < r)) { v = ov; r = orr; }
        }
        const int pr = __shfl(r, 0, 64);
        const double best = __shfl(v, 0, 64);
        if (!(best > 1e-14)) return false;
        if (pr != k && lane < 20) { const double t = A[k][lane]; A[k][lane] = A[pr][lane]; A[pr][lane] = t; }
        wave_meet();
        const double piv = A[k][k];
        wave_meet();  // (every lane holds the pivot before lane k overwrites it)
        if (lane >= k && lane < 20) A[k][lane] /= piv;
        wave_meet();
        // entries (r, c), c >= k, r != k: four per lane
        double mr[4], nv[4];
        int idx[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int e = lane + 64 * u, rr = e / 20, cc = e - rr * 20;
            idx[u] = (e < 200 && rr != k && cc >= k) ? e : -1;
            mr[u] = idx[u] >= 0 ? A[rr][k] : 0.0;
        }
        wave_meet();  // (the multipliers are read before column k is cleared)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (idx[u] >= 0) {
                const int rr = idx[u] / 20, cc = idx[u] - rr * 20;
                nv[u] = A[rr][cc];
                if (mr[u] != 0.0) nv[u] -= mr[u] * A[k][cc];
                A[rr][cc] = nv[u];
            }
        }
        wave_meet();
    }
    return true;
}
// fivept::ns_eliminate with the 45 entries over the lanes of a wave (complete pivoting: the first largest entry in row-major order -- a butterfly over
// (value, entry index): larger value, lower index among equals), the same operations entry by entry.  Uniform result.
__device__ __forceinline__ bool ns_eliminate_wave(double (*A)[9], int* perm, int lane) {
    if (lane < 9) perm[lane] = lane;
    const int er = lane / 9, ec = lane - er * 9;  // my entry (lanes < 45)
    wave_meet();
    for (int k = 0; k < 5; k++) {
        double v = (lane < 45 && er >= k && ec >= k) ? vk_abs(A[er][ec]) : -2.0;
        if (!(v == v)) v = -2.0;  // (the serial `v > best` never takes a NaN)
        int e = lane;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double ov = __shfl_xor(v, d, 64); const int oe = __shfl_xor(e, d, 64);
            if (ov > v || (ov == v && oe < e)) { v = ov; e = oe; }
        }
        if (!(v > 1e-12)) return false;
        const int pr = e / 9, pc = e - pr * 9;
        if (pr != k && lane < 9) { const double t = A[k][lane]; A[k][lane] = A[pr][lane]; A[pr][lane] = t; }
        wave_meet();
        if (pc != k) {
            if (lane < 5) { const double t = A[lane][k]; A[lane][k] = A[lane][pc]; A[lane][pc] = t; }
            if (lane == 5) { const int t = perm[k]; perm[k] = perm[pc]; perm[pc] = t; }
        }
        wave_meet();
        const double piv = A[k][k];
        wave_meet();
        if (lane >= k && lane < 9) A[k][lane] /= piv;
        wave_meet();
        const bool mine = lane < 45 && er != k && ec >= k;
        const double m = mine ? A[er][k] : 0.0;
        wave_meet();  // (the multipliers are read before column k is cleared)
        if (mine && m != 0.0) A[er][ec] -= m * A[k][ec];
        wave_meet();
    }
    return true;
}
__global__ __launch_bounds__(128) static void k_boot_hyp5(const float* __restrict__ p2, BootGeom g, double* __restrict__ Es) {
    __shared__ Boot5Wave S[2];
    const int hy = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    Boot5Wave& w = S[wv];
    if (lane == 0) {
        w.deg = 0; w.nz = 0; w.ne = 0; w.ok = 0;
        double q1[5][2], q2[5][2];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int i = (int)(rng3(RAND_SEED, (uint32_t)hy, 0x200u + (uint32_t)k) % (uint32_t)g.n);
            boot_corr(g, p2, i, q1[k], q2[k]);
        }
        fivept::epipolar_rows(q1, q2, w.NA);  // (the elimination works on the system itself)
    }
    wave_meet();
    {
        const bool ok = ns_eliminate_wave(w.NA, w.perm, lane);
        if (lane == 0 && ok) {
            if (wv == 0) w.ok = fivept::ns_finish(w.NA, w.perm, w.N) ? 1 : 0;
            else {
                double (*N0)[9] = reinterpret_cast<double (*)[9]>(&w.A[0][0]);  // (free until the elimination of the constraints)
                w.ok = fivept::ns_finish(w.NA, w.perm, N0) ? 1 : 0;
                if (w.ok) fivept::second_basis(N0, w.N);
            }
            if (w.ok) fivept::entry_polys(w.N, w.e);
        }
    }
    wave_meet();
    if (w.ok) {  // (uniform per wave)
        if (lane < 10) fivept::constraint_row(w.e, lane, w.W.A0[lane]);
        wave_meet();
        for (int e = lane; e < 200; e += 64) w.A[e / 20][e % 20] = w.W.A0[e / 20][e % 20];
        wave_meet();
        const bool ok = gauss_jordan_10x20_wave(w.A, lane);
        if (lane == 0 && ok) { fivept::basis_poly(w.A, w.W); w.deg = fivept::poly_trim(w.W.n, 10, w.c); }
        wave_meet();
    }
    const int deg = w.deg;
    if (deg >= 1) {  // (uniform per wave; nothing below meets the other wave before the barrier after this block)
        fivept::RootState st = { fivept::aberth_start(lane < 10 ? lane : 0), lane >= deg };
        for (int sw = 0; sw < fivept::ABERTH_SWEEPS; sw++) {
            fivept::Cx zs[10];
#pragma unroll
            for (int j = 0; j < 10; j++) zs[j] = { __shfl(st.z.re, j, 64), __shfl(st.z.im, j, 64) };
            if (lane < deg) st = fivept::aberth_move(w.c, deg, zs, st, lane);
            if (__ballot(!st.done) == 0ull) break;
        }
        if (lane < deg) w.z[lane] = st.z;
        wave_meet();
        if (lane == 0) w.nz = fivept::real_candidates(w.c, deg, w.z, w.zs);
        wave_meet();
    }
    {
        double E[9];
        bool ok = false;
        if (lane < w.nz) ok = fivept::polish_candidate(w.N, w.W, w.zs[lane], E);
        const unsigned long long m = __ballot(ok);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (ok && rank < 10) { for (int cc = 0; cc < 9; cc++) w.Es[rank][cc] = E[cc]; }
        if (lane == 0) w.ne = min(10, __popcll(m));
    }
    __syncthreads();  // the one meeting of the two waves
    if (threadIdx.x == 0) S[0].ne = fivept::merge_models(S[0].Es, S[0].ne, S[1].Es, S[1].ne);
    __syncthreads();
    if (threadIdx.x < 90) {
        const int m = threadIdx.x / 9, k = threadIdx.x % 9;
        Es[((size_t)hy * 10 + m) * 9 + k] = m < S[0].ne ? S[0].Es[m][k] : __builtin_nan("");
    }
}

// one workgroup per hypothesis: squared Sampson distances of the scoring subset and their median = the element of rank ns / 2 in
// ascending order (what a full sort would leave at s[ns / 2]).  The distances are >= 0 (NaN counted as +inf), so their bit patterns
// order like the values: a most-significant-digit-first radix SELECT finds that element in 8 passes of 8 bits over values that stay in
// registers -- a histogram of the still-matching candidates, a scan to the bin that holds the rank -- instead of a 66-step bitonic
// sort of 2048 doubles in LDS.
__global__ __launch_bounds__(256) static void k_boot_score(const float* __restrict__ p2, BootGeom g, const double* __restrict__ Es,
                                                            double* __restrict__ med) {
    constexpr int PER = BOOT_SCORE_MAX / 256;
    __shared__ unsigned hist[256];
    __shared__ unsigned s_wave[4];
    __shared__ unsigned s_bin, s_rank;
    const int hy = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double E[9];
    for (int k = 0; k < 9; k++) E[k] = Es[(size_t)hy * 9 + k];
    if (!(E[0] == E[0])) { if (tid == 0) med[hy] = INFINITY; return; }  // an unused model slot of a five-point sample (uniform: the whole workgroup leaves)
    unsigned long long key[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int j = tid + u * 256;
        double v = INFINITY;  // padding sorts to the end; NaN distances are ordered as +inf as well
        if (j < g.ns) {
            double q1[2], q2[2];
            boot_corr(g, p2, j * g.sstride, q1, q2);
            v = boot_sampson(E, q1, q2);
            if (!(v == v)) v = INFINITY;
        }
        key[u] = (unsigned long long)__double_as_longlong(v);
    }
    unsigned long long prefix = 0ull, mask = 0ull;
    unsigned rank = (unsigned)(g.ns / 2);  // rank among the candidates that match `prefix` under `mask`
#pragma unroll 1
    for (int shift = 56; shift >= 0; shift -= 8) {
        hist[tid] = 0u;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; u++)
            if ((key[u] & mask) == prefix) atomicAdd(&hist[(unsigned)(key[u] >> shift) & 255u], 1u);
        __syncthreads();
        // inclusive scan of the 256 bins (thread = bin); the bin whose range contains `rank` is the next digit
        const unsigned mine = hist[tid];
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        unsigned base = 0u;
        for (int k = 0; k < wv; k++) base += s_wave[k];
        incl += base;
        if (rank < incl && rank >= incl - mine) { s_bin = (unsigned)tid; s_rank = rank - (incl - mine); }  // exactly one bin
        __syncthreads();
        prefix |= (unsigned long long)s_bin << shift;
        mask |= 0xffull << shift;
        rank = s_rank;
    }
    if (tid == 0) med[hy] = __longlong_as_double((long long)prefix);
}

// argmin over hypotheses, decomposition, cheirality vote, pose write-back (single workgroup)
__global__ __launch_bounds__(256) static void k_boot_select(const float* __restrict__ p2, BootGeom g, const double* __restrict__ Es,
                                                             const double* __restrict__ med, PoseBlock* P, CamState* cam0, float* __restrict__ Mb, int strict, int n_models) {
    __shared__ double sRc[2][9], stc[3];
    __shared__ int s_cnt[4][4];
    const int tid = threadIdx.x;
    // first strict minimum of the medians (hypothesis 0 if none is below +inf): every thread takes one, ties go to the lower index
    __shared__ double s_bm[4];
    __shared__ int s_bi[4];
    {
        double bm = INFINITY;
        int bi = n_models;  // never "below": loses against everything, like the sequential scan
        for (int i = tid; i < n_models; i += 256) { const double m = med[i]; if (m < bm) { bm = m; bi = i; } }  // (ascending indices: ties stay with the lower one)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double om = __shfl_xor(bm, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (om < bm || (om == bm && oi < bi)) { bm = om; bi = oi; }
        }
        if ((tid & 63) == 0) { s_bm[tid >> 6] = bm; s_bi[tid >> 6] = bi; }
    }
    __syncthreads();
    if (tid == 0) {
        int best = n_models;
        double bm = INFINITY;
        for (int k = 0; k < 4; k++) if (s_bm[k] < bm || (s_bm[k] == bm && s_bi[k] < best)) { bm = s_bm[k]; best = s_bi[k]; }
        double E[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 0 }, Rc[2][9], tc[3];
        if (best >= n_models) best = (n_models == BOOT_HYPS) ? 0 : -1;  // no finite median: the 8-point path falls back to hypothesis 0, the five-point path to a fixed E
        if (best >= 0) for (int k = 0; k < 9; k++) E[k] = Es[(size_t)best * 9 + k];
        boot_decompose(E, Rc, tc);
        for (int k = 0; k < 9; k++) { sRc[0][k] = Rc[0][k]; sRc[1][k] = Rc[1][k]; }
        for (int k = 0; k < 3; k++) stc[k] = tc[k];
    }
    __syncthreads();
    int cnt[4] = { 0, 0, 0, 0 };
    for (int j = tid; j < g.ns; j += 256) {
        double q1[2], q2[2];
        boot_corr(g, p2, j * g.sstride, q1, q2);
#pragma unroll
        for (int cand = 0; cand < 4; cand++)
            if (boot_in_front(sRc[cand >> 1], stc, (cand & 1) ? -1.0 : 1.0, q1, q2)) cnt[cand]++;
    }
#pragma unroll
    for (int cand = 0; cand < 4; cand++) {
        int c = cnt[cand];
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((tid & 63) == 0) s_cnt[cand][tid >> 6] = c;
    }
    __syncthreads();
    if (tid == 0) {
        int best = 0, best_cnt = -1;
        for (int cand = 0; cand < 4; cand++) {
            const int c = s_cnt[cand][0] + s_cnt[cand][1] + s_cnt[cand][2] + s_cnt[cand][3];
            if (c > best_cnt) { best_cnt = c; best = cand; }
        }
        float R[9], t[3], rv[3];
        boot_finish(sRc[best >> 1], stc, (best & 1) ? -1.0 : 1.0, R, t);
        if (strict) vrcv_rvec_of_R32(R, rv, 1);  // reference mode: Camera::rvec() of the float matrix (vk_ref_cv.h), what the first mean shift's displacement test sees
        else rotmat_to_angle_axis(R, rv, false);
        for (int k = 0; k < 9; k++) P->Rs[0][k] = R[k];
        for (int k = 0; k < 3; k++) { P->ts[0][k] = t[k]; cam0->rvec[k] = rv[k]; cam0->t[k] = t[k]; }
        const float K9[9] = { (float)g.fx, 0, (float)g.cx, 0, (float)g.fy, (float)g.cy, 0, 0, 1 };
        boot_depth_coeffs(K9, R, t, Mb);
    }
}

// geometry.cpp:267-285; M = K R K^-1 (row-major), b = K t
__global__ __launch_bounds__(256) static void k_depth_closed_form(const float2* __restrict__ flow, float* __restrict__ depth, int w, int h,
                                                                   const float* __restrict__ Mb, float min_depth, float max_depth) {
#pragma clang fp contract(off)  // same fp32 op sequence as the un-fused host loop of geometry.cpp:272-284
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float b1 = Mb[9], b2 = Mb[10], b3 = Mb[11];
    float2 d = flow[y * w + x];
    float w1 = Mb[0] * x + Mb[1] * y + Mb[2], w2 = Mb[3] * x + Mb[4] * y + Mb[5], w3 = Mb[6] * x + Mb[7] * y + Mb[8];
    float a1 = x + d.x, a2 = y + d.y;
    float zn = (a1 * b3 - b1) * (w1 - a1 * w3) + (a2 * b3 - b2) * (w2 - a2 * w3);
    float zd = (w1 - a1 * w3) * (w1 - a1 * w3) + (w2 - a2 * w3) * (w2 - a2 * w3);
    depth[y * w + x] = fminf(fmaxf(zn / zd, min_depth), max_depth);
}

int bootstrap_device(Context* c, ImageSet& S, int w, int h, float fx, float fy, float cx, float cy, CamState* cam0_dev, bool strict, int points) {
    const BootGeom g = boot_geom(w, h, fx, fy, cx, cy);
    const size_t bytes = sizeof(double) * (BOOT_MAX_MODELS * 10 + 4) + sizeof(float) * (2 * (size_t)g.n + 16) + 64;
    if (int e = c->tmp.reserve(bytes)) return e;
    char* base = c->tmp.as<char>();
    double* d_E = reinterpret_cast<double*>(base);                       // [models][9]
    double* d_med = d_E + BOOT_MAX_MODELS * 9;                            // [models]
    float* d_Mb = reinterpret_cast<float*>(d_med + BOOT_MAX_MODELS);      // [12] (+4 pad)
    float* d_p2 = d_Mb + 16;                                              // [n][2]
    if (g.n >= 8) {
        hipLaunchKernelGGL(k_extract_corr, dim3((g.n + 255) / 256), dim3(256), 0, c->stream, S.flows.as<float2>(), d_p2, w, h, g.step, g.nx, g.n);
        const int n_models = points == 5 ? BOOT5_MODELS : BOOT_HYPS;
        if (points == 5) hipLaunchKernelGGL(k_boot_hyp5, dim3(BOOT5_SAMPLES), dim3(128), 0, c->stream, d_p2, g, d_E);
        else hipLaunchKernelGGL(k_boot_hyp, dim3(BOOT_HYPS / 64), dim3(64), 0, c->stream, d_p2, g, d_E);
        hipLaunchKernelGGL(k_boot_score, dim3(n_models), dim3(256), 0, c->stream, d_p2, g, d_E, d_med);
        hipLaunchKernelGGL(k_boot_select, dim3(1), dim3(256), 0, c->stream, d_p2, g, d_E, d_med, S.pb(), cam0_dev, d_Mb, strict ? 1 : 0, n_models);
    } else {  // too few correspondences: identity pose (the host path returns failure and keeps R=I, t=0)
        const float K9[9] = { fx, 0, cx, 0, fy, cy, 0, 0, 1 }, R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, t[3] = { 0, 0, 0 };
        float Mb[12];
        boot_depth_coeffs(K9, R, t, Mb);
        VK_CHECK(hipMemcpyAsync(d_Mb, Mb, sizeof Mb, hipMemcpyHostToDevice, c->stream));
        VK_CHECK(hipStreamSynchronize(c->stream));
    }
    hipLaunchKernelGGL(k_depth_closed_form, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, c->stream, S.flows.as<float2>(),
                       S.depth.as<float>(), w, h, d_Mb, 1e-2f, 1e10f);
    VK_CHECK_LAST();
    return 0;
}

}  // namespace vk

extern "C" {
// host path of the same procedure (no GPU needed): bit-identical to the kernels above
int vk_estimate_pose_epipolar(const float* h_flow, const float* h_K, int w, int h, float* h_o_R9, float* h_o_t3) {
    const int step = vk::BOOT_STEP, nx = (w + step - 1) / step, ny = (h + step - 1) / step;
    std::vector<float> p2((size_t)nx * ny * 2);
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {
            const int x = i * step, y = j * step;
            p2[(size_t)(j * nx + i) * 2] = (float)x + h_flow[((size_t)y * w + x) * 2];
            p2[(size_t)(j * nx + i) * 2 + 1] = (float)y + h_flow[((size_t)y * w + x) * 2 + 1];
        }
    return vk::lmeds_essential_host(p2.data(), w, h, h_K[0], h_K[4], h_K[2], h_K[5], h_o_R9, h_o_t3) ? 0 : 1;
}
// the same with the five-point minimal solver (vk_fivept.hpp; config key --bootstrap_points 5): host path, no GPU needed
int vk_estimate_pose_epipolar5(const float* h_flow, const float* h_K, int w, int h, float* h_o_R9, float* h_o_t3) {
    const int step = vk::BOOT_STEP, nx = (w + step - 1) / step, ny = (h + step - 1) / step;
    std::vector<float> p2((size_t)nx * ny * 2);
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {
            const int x = i * step, y = j * step;
            p2[(size_t)(j * nx + i) * 2] = (float)x + h_flow[((size_t)y * w + x) * 2];
            p2[(size_t)(j * nx + i) * 2 + 1] = (float)y + h_flow[((size_t)y * w + x) * 2 + 1];
        }
    return vk::lmeds_essential_host5(p2.data(), w, h, h_K[0], h_K[4], h_K[2], h_K[5], h_o_R9, h_o_t3) ? 0 : 1;
}
// the five-point solver alone: q1, q2 = five normalised correspondences [5][2] (image 1, image 2); Es receives up to ten essential matrices
// (row-major, Frobenius norm sqrt 2); returns their number.  Host code.
int vk_fivept_solve(const double* q1, const double* q2, double* Es) {
    return vk::fivept::solve(reinterpret_cast<const double (*)[2]>(q1), reinterpret_cast<const double (*)[2]>(q2), reinterpret_cast<double (*)[9]>(Es));
}
// (vk_debug.h) the degree-10 polynomial of a sample and the real roots the solver found
__attribute__((visibility("default"))) int vk_fivept_debug(const double* q1, const double* q2, double* poly11, double* roots10) {
    double Es[10][9]; int nr = 0;
    vk::fivept::solve(reinterpret_cast<const double (*)[2]>(q1), reinterpret_cast<const double (*)[2]>(q2), Es, poly11, roots10, &nr);
    return nr;
}
// GPU path on a host flow: runs the bootstrap kernels, returns pose and closed-form depth
int vk_bootstrap_gpu(const float* h_flow, const float* h_K, int w, int h, float* h_o_R9, float* h_o_t3, float* h_o_depth) { return vk_bootstrap_gpu_points(h_flow, h_K, w, h, 8, h_o_R9, h_o_t3, h_o_depth); }
int vk_bootstrap_gpu_points(const float* h_flow, const float* h_K, int w, int h, int points, float* h_o_R9, float* h_o_t3, float* h_o_depth) {
    if (points != 5 && points != 8) return (int)hipErrorInvalidValue;
    vk::Context* c = vk::default_context();
    if (!c) return (int)hipErrorNoDevice;
    vk::ImageSet& S = c->od;
    const size_t npx = (size_t)w * h;
    if (int e = S.ensure_pose()) return e;
    if (int e = S.flows.reserve(sizeof(float) * 2 * npx)) return e;
    if (int e = S.depth.reserve(sizeof(float) * npx)) return e;
    if (int e = c->cams.reserve(sizeof(vk::CamState) * vk::MAX_FRAMES)) return e;
    VK_CHECK(hipMemcpyAsync(S.flows.p, h_flow, sizeof(float) * 2 * npx, hipMemcpyHostToDevice, c->stream));
    if (int e = vk::bootstrap_device(c, S, w, h, h_K[0], h_K[4], h_K[2], h_K[5], c->cams.as<vk::CamState>(), false, points)) return e;
    if (h_o_R9) VK_CHECK(hipMemcpyAsync(h_o_R9, S.pb()->Rs[0], sizeof(float) * 9, hipMemcpyDeviceToHost, c->stream));
    if (h_o_t3) VK_CHECK(hipMemcpyAsync(h_o_t3, S.pb()->ts[0], sizeof(float) * 3, hipMemcpyDeviceToHost, c->stream));
    if (h_o_depth) VK_CHECK(hipMemcpyAsync(h_o_depth, S.depth.p, sizeof(float) * npx, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}
int vk_estimate_depth_closed_form(const float* h_flow, const float* h_K, const float* h_R9, const float* h_t3, int w, int h,
                                  float* h_o_depth) {
    vk::Context* c = vk::default_context();
    if (!c) return (int)hipErrorNoDevice;
    const size_t npx = (size_t)w * h;
    if (int e = c->tmp.reserve(sizeof(float) * (3 * npx + 16))) return e;
    float* d_flow = c->tmp.as<float>(); float* d_depth = d_flow + 2 * npx; float* d_s = d_depth + npx;
    float Mb[12];
    vk::boot_depth_coeffs(h_K, h_R9, h_t3, Mb);
    VK_CHECK(hipMemcpyAsync(d_flow, h_flow, sizeof(float) * 2 * npx, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipMemcpyAsync(d_s, Mb, sizeof Mb, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    hipLaunchKernelGGL(vk::k_depth_closed_form, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, c->stream,
                       reinterpret_cast<const float2*>(d_flow), d_depth, w, h, d_s, 1e-2f, 1e10f);
    VK_CHECK_LAST();
    VK_CHECK(hipMemcpyAsync(h_o_depth, d_depth, sizeof(float) * npx, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}
}
