/* oracle/orc_pose.c -- ORACLE (test infrastructure only, see orc.h).
 * Pose half of the EM loop: 2D-3D correspondence collection, compaction, batched P3P
 * (LambdaTwist / AP3P), rotation -> angle-axis, mean-shift, robust Gaussian fit.
 * Restated from gpu-kernels/{collect_p3p_instances.cu,solve_batch_lambdatwist.cu,
 * solve_batch_ap3p.cu,rodrigues.h,meanshift.cu,fit_robust_gaussian.cu,aux_funs.cpp},
 * the lambdatwist headers and voldor/geometry.cpp; citations inline. */
#include "orc.h"
#include "orc_math.h"
#include "../voldor_amd/csrc/vk_ref_svd.h"
#include <tgmath.h>
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- LambdaTwist: two instantiations of the restated template ---- */
#define LT_CAT2(a, b) a##b
#define LT_CAT(a, b) LT_CAT2(a, b)

#define LT_T float
#define LT_NAME(n) LT_CAT(ltf_, n)
#define LT_NUMERIC_LIMIT ((float)1e-7)
#include "orc_lambdatwist_impl.h"
#undef LT_T
#undef LT_NAME
#undef LT_NUMERIC_LIMIT

#define LT_T double
#define LT_NAME(n) LT_CAT(ltd_, n)
#define LT_NUMERIC_LIMIT (1e-13)
#include "orc_lambdatwist_impl.h"
#undef LT_T
#undef LT_NAME
#undef LT_NUMERIC_LIMIT

int orc_lambdatwist_p4p(const float* y8, const float* x12, float fx, float fy, float cx, float cy,
                        int use_double, float* R9, float* t3) {
    return use_double ? ltd_p4p(y8, x12, fx, fy, cx, cy, R9, t3)
                      : ltf_p4p(y8, x12, fx, fy, cx, cy, R9, t3);
}

/* ------------------------------------------------------------------ collect_p3p
 * collect_p3p_instances.cu:38-55 helpers, :57-67 rigidness sum, :70-145 map kernel */
extern void orc_bilinear2(const float* img, int w, int h, float x, float y, float* ox, float* oy);
extern void orc_fetch2(const float* stack, int f, int n_layers, int w, int h, float x, float y, float* ox, float* oy);  /* at_tex: D2 or CUDA's filter (orc_set_reference_tex) */
extern int orc_get_reference_rng(void);
extern const uint32_t* orc_xorwow_jumps(void);
#include "../voldor_amd/csrc/vk_ref_cuda.h"

void orc_collect_p3p(const float* flows, const float* rig, const float* depth, const float* K,
                     const float (*Rs)[9], const float (*ts)[3], float* p2_map, float* p3_map,
                     int N, int w, int h, int active_idx, float rigidness_thresh,
                     float rigidness_sum_thresh, float sample_min_depth, float sample_max_depth,
                     int max_trace_on_flow) {
    const float K4[4] = { K[0], K[2], K[4], K[5] };
    const float K4i[4] = { 1.f / K[0], -K[2] / K[0], 1.f / K[4], -K[5] / K[4] };
    const int npx = w * h;
    const float qnan = __builtin_nanf("");
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            const int pi = y * w + x;
            p2_map[pi * 2] = p2_map[pi * 2 + 1] = qnan;
            p3_map[pi * 3] = p3_map[pi * 3 + 1] = p3_map[pi * 3 + 2] = qnan;
            float d = depth[pi];
            if (d < sample_min_depth || (sample_max_depth > 0 && d > sample_max_depth)) continue;
            float rsum = 0; /* :57-67 */
            for (int i = 0; i < N; i++) rsum += rig[(size_t)i * npx + pi];
            if (rsum < rigidness_sum_thresh && rigidness_sum_thresh > N + 1) continue; /* sic, :88-90 */
            int n_trace = 0;
            float prod = 1;
            int lo = max_trace_on_flow > 0 ? (active_idx - max_trace_on_flow + 1 > 0 ? active_idx - max_trace_on_flow + 1 : 0) : 0;
            for (int i = active_idx; i >= lo; i--) {
                prod *= rig[(size_t)i * npx + pi];
                if (prod > rigidness_thresh) n_trace++;
                else break;
            }
            if (n_trace <= 0) continue;
            int out = 0;
            float px = 0, py = 0, o[3];
            o[0] = (K4i[0] * x + K4i[1]) * d; o[1] = (K4i[2] * y + K4i[3]) * d; o[2] = d;
            for (int i = 0; i <= active_idx; i++) {
                if (i >= active_idx - n_trace + 1) {
                    if (i == active_idx - n_trace + 1) {
                        px = (K4[0] * o[0] + K4[1] * o[2]) / o[2];
                        py = (K4[2] * o[1] + K4[3] * o[2]) / o[2];
                    }
                    if (px > 0 && px < w && py > 0 && py < h) { /* strict, :120 */
                        float fx_, fy_;
                        orc_fetch2(flows, i, N, w, h, px, py, &fx_, &fy_);
                        px += fx_; py += fy_;
                    } else { out = 1; break; }
                }
                if (i < active_idx) {
                    const float* R = Rs[i]; const float* t = ts[i];
                    float a = o[0] * R[0] + o[1] * R[1] + o[2] * R[2];
                    float b = o[0] * R[3] + o[1] * R[4] + o[2] * R[5];
                    float c = o[0] * R[6] + o[1] * R[7] + o[2] * R[8];
                    o[0] = a + t[0]; o[1] = b + t[1]; o[2] = c + t[2];
                }
            }
            /* geometry.cpp:73 drops entries whose sum is not finite; do it here so that map validity and
             * compacted-list membership are the same set */
            if (!out && o[2] > sample_min_depth && (sample_max_depth <= 0 || o[2] < sample_max_depth) &&
                isfinite(px + py + o[0] + o[1] + o[2])) {
                p2_map[pi * 2] = px; p2_map[pi * 2 + 1] = py;
                p3_map[pi * 3] = o[0]; p3_map[pi * 3 + 1] = o[1]; p3_map[pi * 3 + 2] = o[2];
            }
        }
    }
}

/* voldor/geometry.cpp:68-80 */
int orc_compact_p3p(const float* p2_map, const float* p3_map, int npx, float* pts2, float* pts3) {
    int n = 0;
    for (int i = 0; i < npx; i++) {
        float s = p2_map[i * 2] + p2_map[i * 2 + 1] + p3_map[i * 3] + p3_map[i * 3 + 1] + p3_map[i * 3 + 2];
        if (isfinite(s)) {
            pts2[n * 2] = p2_map[i * 2]; pts2[n * 2 + 1] = p2_map[i * 2 + 1];
            pts3[n * 3] = p3_map[i * 3]; pts3[n * 3 + 1] = p3_map[i * 3 + 1]; pts3[n * 3 + 2] = p3_map[i * 3 + 2];
            n++;
        }
    }
    return n;
}

/* solve_batch_lambdatwist.cu:16-19 with deviations D1 (RNG) and D3 (clamp). The reference
 * re-seeds per call (:80-81), so the pattern depends only on (idx, n_pts). */
void orc_pose_sample_indices(int idx, int n_pts, int out4[4]) {
    vrc_xorwow st;
    const int xw = orc_get_reference_rng();
    if (xw) vrc_xorwow_init(orc_xorwow_jumps(), 233ull, (uint32_t)idx, &st);  /* curand_init(RAND_SEED, idx, 0, ..) per call (:44-48) */
    for (int k = 0; k < 4; k++) {
        float u = xw ? vrc_uniform(vrc_xorwow_next(&st)) : orc_u01(orc_rng(233u, (uint32_t)idx, (uint32_t)k));
        int i = (int)(u * (float)n_pts);
        if (i > n_pts - 1) i = n_pts - 1;
        out4[k] = i;
    }
}

/* Window-pipeline variant of the draw (deviation D3b, DESIGN.md "sampling stability"): the same
 * uniform distribution over the valid correspondences, drawn by rejection over the NaN-marked map
 * instead of indexing the compacted list, so that a one-pixel change of the valid set only
 * perturbs the hypotheses that hit that pixel.  <=256 tries per point, else the hypothesis fails. */
#define ORC_DRAW_MAX_TRIES 256
int orc_pose_sample_pixels(int idx, int npx, const float* p2_map, int out4[4]) {
    for (int k = 0; k < 4; k++) {
        int found = -1;
        for (int j = 0; j < ORC_DRAW_MAX_TRIES; j++) {
            uint32_t r = orc_rng(233u, (uint32_t)idx, (uint32_t)(k * ORC_DRAW_MAX_TRIES + j));
            int cand = (int)(((uint64_t)r * (uint64_t)npx) >> 32);
            if (isfinite(p2_map[(size_t)cand * 2])) { found = cand; break; }
        }
        if (found < 0) return 0;
        out4[k] = found;
    }
    return 1;
}

/* ------------------------------------------------------------------ rotation helpers
 * rodrigues.h:5-79 (Ceres RotationMatrixToAngleAxis in float) */
void orc_rotmat_to_angle_axis(const float* R9, float* aa) {
    const float (*R)[3] = (const float (*)[3])R9;
    aa[0] = R[2][1] - R[1][2];
    aa[1] = R[0][2] - R[2][0];
    aa[2] = R[1][0] - R[0][1];
    float costheta = fminf(fmaxf((R[0][0] + R[1][1] + R[2][2] - 1.f) * 0.5f, -1.f), 1.f);
    float sintheta = fminf(sqrtf(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]) * 0.5f, 1.f);
    const float theta = om_atan2f(sintheta, costheta);
    if ((sintheta > FLT_EPSILON) || (sintheta < -FLT_EPSILON)) {
        const float r = theta / (2.f * sintheta);
        aa[0] *= r; aa[1] *= r; aa[2] *= r;
        return;
    }
    if (costheta > 0) { aa[0] *= 0.5f; aa[1] *= 0.5f; aa[2] *= 0.5f; return; }
    const float inv_one_minus_costheta = 1.f / (1.f - costheta);
    for (int i = 0; i < 3; ++i) {
        aa[i] = theta * sqrtf((R[i][i] - costheta) * inv_one_minus_costheta);
        if (((sintheta < 0) && (aa[i] > 0)) || ((sintheta > 0) && (aa[i] < 0))) aa[i] = -aa[i];
    }
}

/* rodrigues.h:82-114 projects R to SO(3) as U*V^T of an approximate SVD (svd3_cuda.h, McAdams
 * et al., 4 Jacobi sweeps, float) before the angle-axis conversion.  The oracle computes the
 * same polar factor exactly (Newton iteration X <- (X + X^-T)/2 in double); agreement with the
 * reference's approximate SVD is checked against oracle/_ref at 1e-5 (tests). */
static void polar_rotation(const float* R9, float* Q9) {
    double X[9], Y[9];
    for (int i = 0; i < 9; i++) X[i] = R9[i];
    for (int it = 0; it < 30; it++) {
        double c[9];
        c[0] = X[4] * X[8] - X[5] * X[7]; c[1] = X[5] * X[6] - X[3] * X[8]; c[2] = X[3] * X[7] - X[4] * X[6];
        c[3] = X[2] * X[7] - X[1] * X[8]; c[4] = X[0] * X[8] - X[2] * X[6]; c[5] = X[1] * X[6] - X[0] * X[7];
        c[6] = X[1] * X[5] - X[2] * X[4]; c[7] = X[2] * X[3] - X[0] * X[5]; c[8] = X[0] * X[4] - X[1] * X[3];
        double det = X[0] * c[0] + X[1] * c[1] + X[2] * c[2];
        if (!(fabs(det) > 1e-300)) break;
        const double idet = 1.0 / det;
        double delta = 0;
        for (int i = 0; i < 9; i++) { Y[i] = 0.5 * (X[i] + c[i] * idet); delta += fabs(Y[i] - X[i]); } /* cof/det = X^-T */
        memcpy(X, Y, sizeof X);
        if (delta < 1e-10) break; /* quadratic convergence: the next step would move X by ~delta^2 */
    }
    for (int i = 0; i < 9; i++) Q9[i] = (float)X[i];
}
/* Test hook: tests/test_oracle_vs_ref_window.py installs the reference's own rodrigues() (oracle/_ref, ref_rodrigues) here
 * to take the approximate-SVD difference out of a whole-window comparison with the reference pipeline.  NULL = the oracle's. */
static void (*g_rodrigues_hook)(const float* R9, float* rvec3, float* Rproj9) = NULL;
void orc_set_rodrigues_hook(void (*fn)(const float*, float*, float*)) { g_rodrigues_hook = fn; }
/* Reference-SVD mode (config key --reference_svd 1 of the product; orc_set_reference_svd here): the projection is U V^T of the
 * reference's approximate fp32 SVD, restated to the bit in voldor_amd/csrc/vk_ref_svd.h (svd3_cuda.h:36-1044, rodrigues.h:82-108) and
 * pinned against the reference's own rodrigues() compiled in place (tests/test_oracle_vs_golden.py).  Retires D8 for parity runs. */
static int g_reference_svd = 0;
void orc_set_reference_svd(int on) { g_reference_svd = on; }
int orc_get_reference_svd(void) { return g_reference_svd; }
void orc_reference_project_rotation(const float* R9, float* Q9) { memcpy(Q9, R9, 9 * sizeof(float)); vrs_project_rotation(Q9); }
void orc_rodrigues(const float* R9, float* rvec3) {
    if (g_rodrigues_hook) { g_rodrigues_hook(R9, rvec3, NULL); return; }
    float Q[9];
    if (g_reference_svd) { orc_reference_project_rotation(R9, Q); orc_rotmat_to_angle_axis(Q, rvec3); return; }
    polar_rotation(R9, Q);
    orc_rotmat_to_angle_axis(Q, rvec3);
}

/* cv::Rodrigues (vector -> matrix), OpenCV 3.4 modules/calib3d/src/calibration.cpp
 * cvRodrigues2: double arithmetic, theta < DBL_EPSILON -> identity;
 * R = c*I + (1-c)*r r^T + s*[r]_x.  Used at voldor/geometry.cpp:258, voldor.cpp:63. */
void orc_rvec_to_rotmat(const float* rvec3, float* R9) {
    double rx = rvec3[0], ry = rvec3[1], rz = rvec3[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R9[i] = (i % 4 == 0) ? 1.f : 0.f;
        return;
    }
    double c = om_cos(theta), s = om_sin(theta), c1 = 1. - c, it = 1. / theta;
    rx *= it; ry *= it; rz *= it;
    double rrt[9] = { rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz };
    double rxm[9] = { 0, -rz, ry, rz, 0, -rx, -ry, rx, 0 };
    for (int i = 0; i < 9; i++) R9[i] = (float)(c * (i % 4 == 0 ? 1. : 0.) + c1 * rrt[i] + s * rxm[i]);
}

/* ------------------------------------------------------------------ AP3P
 * solve_batch_ap3p.cu:9-26 complex helpers on top of CUDA's cuComplex.h (cuCabsf, cuCdivf
 * restated from the published header), :28-82 solveQuartic, :85-98 polish, :152-292
 * computePoses, :294-328 solve_all, :331-378 4th point selection. */
typedef struct { float x, y; } cplx;
static float c_abs(cplx z) { /* cuComplex.h cuCabsf */
    float a = fabsf(z.x), b = fabsf(z.y), v, w, t;
    if (a > b) { v = a; w = b; } else { v = b; w = a; }
    t = w / v; t = 1.0f + t * t; t = v * sqrtf(t);
    if ((v == 0.0f) || (v > 3.402823466e38f) || (w > 3.402823466e38f)) t = v + w;
    return t;
}
static cplx c_div(cplx x, cplx y) { /* cuComplex.h cuCdivf */
    cplx q; float s = fabsf(y.x) + fabsf(y.y); float oos = 1.0f / s;
    float ars = x.x * oos, ais = x.y * oos, brs = y.x * oos, bis = y.y * oos;
    s = (brs * brs) + (bis * bis); oos = 1.0f / s;
    q.x = ((ars * brs) + (ais * bis)) * oos;
    q.y = ((ais * brs) - (ars * bis)) * oos;
    return q;
}
static cplx c_sqrt(cplx x) { /* :9-15 */
    cplx o;
    o.x = sqrtf(c_abs(x) * (x.x / c_abs(x) + 1.0f) / 2.0f);
    o.y = sqrtf(c_abs(x) * (1.0f - x.x / c_abs(x)) / 2.0f);
    o.y = -fabsf(o.y);
    return o;
}
static cplx c_pow(cplx z, float p) { /* :17-20 */
    float theta = om_atan2f(z.y, z.x);
    cplx o = { om_powf(c_abs(z), p) * om_cosf(p * theta), om_powf(c_abs(z), p) * om_sinf(p * theta) };
    return o;
}
static void solve_quartic(const float* f, float* roots) { /* :28-82, literal incl. the double sqrt at :57 */
    const float a4 = f[0], a3 = f[1], a2 = f[2], a1 = f[3], a0 = f[4];
    float a4_2 = a4 * a4, a3_2 = a3 * a3, a4_3 = a4_2 * a4, a2a4 = a2 * a4;
    float p4 = (8 * a2a4 - 3 * a3_2) / (8 * a4_2);
    float q4 = (a3_2 * a3 - 4 * a2a4 * a3 + 8 * a1 * a4_2) / (8 * a4_3);
    float r4 = (256 * a0 * a4_3 - 3 * (a3_2 * a3_2) - 64 * a1 * a3 * a4_2 + 16 * a2a4 * a3_2) / (256 * (a4_3 * a4));
    float p3 = ((p4 * p4) / 12 + r4) / 3;
    float q3 = (72 * r4 * p4 - 2 * p4 * p4 * p4 - 27 * q4 * q4) / 432;
    float t;
    cplx w = { q3 * q3 - p3 * p3 * p3, 0 };
    w = c_sqrt(w);
    if (q3 >= 0) { w.x = -w.x - q3; w.y = -w.y; }
    else { w = c_sqrt(w); w.x = w.x - q3; }
    if (w.y == 0.0f) { w.x = om_cbrtf(w.x); t = 2.0f * (w.x + p3 / w.x); }
    else { w = c_pow(w, (1.0f / 3.0f)); t = 4.0f * w.x; }
    cplx arg = { -2 * p4 / 3 + t, 0 };
    cplx sqrt_2m = c_sqrt(arg);
    float B_4A = -a3 / (4 * a4);
    cplx complex1 = { 4 * p4 / 3 + t, 0 };
    cplx num = { 2 * q4, 0 };
    cplx complex2 = c_div(num, sqrt_2m);
    float sqrt_2m_rh = sqrt_2m.x * 0.5f;
    cplx s1 = { -(complex1.x + complex2.x), -(complex1.y + complex2.y) };
    float sqrt1 = c_sqrt(s1).x * 0.5f;
    roots[0] = B_4A + sqrt_2m_rh + sqrt1;
    roots[1] = B_4A + sqrt_2m_rh - sqrt1;
    cplx s2 = { -(complex1.x - complex2.x), -(complex1.y - complex2.y) };
    float sqrt2 = c_sqrt(s2).x * 0.5f;
    roots[2] = B_4A - sqrt_2m_rh + sqrt2;
    roots[3] = B_4A - sqrt_2m_rh - sqrt2;
}
static void polish_quartic(const float* c, float* r) { /* :85-98 */
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) {
            float err = (((c[0] * r[j] + c[1]) * r[j] + c[2]) * r[j] + c[3]) * r[j] + c[4];
            float der = ((4 * c[0] * r[j] + 3 * c[1]) * r[j] + 2 * c[2]) * r[j] + c[3];
            r[j] -= err / der;
        }
}
static void v_cross(const float* a, const float* b, float* r) { /* :100-104 */
    r[0] = a[1] * b[2] - a[2] * b[1]; r[1] = -(a[0] * b[2] - a[2] * b[0]); r[2] = a[0] * b[1] - a[1] * b[0];
}
static float v_dot(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static float v_norm(const float* a) { return sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
static void m_mult(const float a[3][3], const float b[3][3], float r[3][3]) { /* :134-146 */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
}
/* :152-292 ; fv/wp: columns are the 3 bearings / world points */
static int ap3p_compute_poses(const float fv[3][3], const float wp[3][3], float sR[4][3][3], float sT[4][3]) {
    float w1[3] = { wp[0][0], wp[1][0], wp[2][0] }, w2[3] = { wp[0][1], wp[1][1], wp[2][1] }, w3[3] = { wp[0][2], wp[1][2], wp[2][2] };
    float u0[3] = { w1[0] - w2[0], w1[1] - w2[1], w1[2] - w2[2] };
    float nu0 = v_norm(u0);
    float k1[3] = { u0[0] / nu0, u0[1] / nu0, u0[2] / nu0 };
    float b1[3] = { fv[0][0], fv[1][0], fv[2][0] }, b2[3] = { fv[0][1], fv[1][1], fv[2][1] }, b3[3] = { fv[0][2], fv[1][2], fv[2][2] };
    float k3[3]; v_cross(b1, b2, k3);
    float nk3 = v_norm(k3);
    k3[0] /= nk3; k3[1] /= nk3; k3[2] /= nk3;
    float tz[3]; v_cross(b1, k3, tz);
    float v1[3]; v_cross(b1, b3, v1);
    float v2[3]; v_cross(b2, b3, v2);
    float u1[3] = { w1[0] - w3[0], w1[1] - w3[1], w1[2] - w3[2] };
    float u1k1 = v_dot(u1, k1);
    float k3b3 = v_dot(k3, b3);
    float f11 = k3b3;
    float f13 = v_dot(k3, v1);
    float f15 = -u1k1 * f11;
    float nl[3]; v_cross(u1, k1, nl);
    float delta = v_norm(nl);
    nl[0] /= delta; nl[1] /= delta; nl[2] /= delta;
    f11 *= delta; f13 *= delta;
    float u2k1 = u1k1 - nu0;
    float f21 = v_dot(tz, v2);
    float f22 = nk3 * k3b3;
    float f23 = v_dot(k3, v2);
    float f24 = u2k1 * f22;
    float f25 = -u2k1 * f21;
    f21 *= delta; f22 *= delta; f23 *= delta;
    float g1 = f13 * f22;
    float g2 = f13 * f25 - f15 * f23;
    float g3 = f11 * f23 - f13 * f21;
    float g4 = -f13 * f24;
    float g5 = f11 * f22;
    float g6 = f11 * f25 - f15 * f21;
    float g7 = -f15 * f24;
    float coeffs[5] = { g5 * g5 + g1 * g1 + g3 * g3,
                        2 * (g5 * g6 + g1 * g2 + g3 * g4),
                        g6 * g6 + 2 * g5 * g7 + g2 * g2 + g4 * g4 - g1 * g1 - g3 * g3,
                        2 * (g6 * g7 - g1 * g2 - g3 * g4),
                        g7 * g7 - g2 * g2 - g4 * g4 };
    float s[4];
    solve_quartic(coeffs, s);
    polish_quartic(coeffs, s);
    float temp[3]; v_cross(k1, nl, temp);
    float Ck1nl[3][3] = { { k1[0], nl[0], temp[0] }, { k1[1], nl[1], temp[1] }, { k1[2], nl[2], temp[2] } };
    float Cb1k3tzT[3][3] = { { b1[0], b1[1], b1[2] }, { k3[0], k3[1], k3[2] }, { tz[0], tz[1], tz[2] } };
    float b3p[3] = { b3[0] * (delta / k3b3), b3[1] * (delta / k3b3), b3[2] * (delta / k3b3) };
    int nb = 0;
    for (int i = 0; i < 4; ++i) {
        float ctheta1p = s[i];
        if (fabsf(ctheta1p) > 1) continue;
        float stheta1p = sqrtf(1 - ctheta1p * ctheta1p);
        stheta1p = (k3b3 > 0) ? stheta1p : -stheta1p;
        float ctheta3 = g1 * ctheta1p + g2;
        float stheta3 = g3 * ctheta1p + g4;
        float ntheta3 = stheta1p / ((g5 * ctheta1p + g6) * ctheta1p + g7);
        ctheta3 *= ntheta3; stheta3 *= ntheta3;
        float C13[3][3] = { { ctheta3, 0, -stheta3 },
                            { stheta1p * stheta3, ctheta1p, stheta1p * ctheta3 },
                            { ctheta1p * stheta3, -stheta1p, ctheta1p * ctheta3 } };
        float tm[3][3], R[3][3];
        m_mult(Ck1nl, C13, tm);
        m_mult(tm, Cb1k3tzT, R);
        float rp3[3] = { w3[0] * R[0][0] + w3[1] * R[1][0] + w3[2] * R[2][0],
                         w3[0] * R[0][1] + w3[1] * R[1][1] + w3[2] * R[2][1],
                         w3[0] * R[0][2] + w3[1] * R[1][2] + w3[2] * R[2][2] };
        for (int k = 0; k < 3; k++) sT[nb][k] = b3p[k] * stheta1p - rp3[k];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) sR[nb][r][c] = R[c][r]; /* transpose, :279-287 */
        nb++;
    }
    return nb;
}
int orc_ap3p_p4p(const float* y, const float* x, float fx, float fy, float cx, float cy, float* R9, float* t3) {
    float mu[3], mv[3], mk[3];
    for (int i = 0; i < 3; i++) { /* :294-318 */
        float u = (y[i * 2] - cx) / fx, v = (y[i * 2 + 1] - cy) / fy;
        float norm = sqrtf(u * u + v * v + 1);
        mk[i] = 1.f / norm; mu[i] = u * mk[i]; mv[i] = v * mk[i];
    }
    float fv[3][3] = { { mu[0], mu[1], mu[2] }, { mv[0], mv[1], mv[2] }, { mk[0], mk[1], mk[2] } };
    float wp[3][3] = { { x[0], x[3], x[6] }, { x[1], x[4], x[7] }, { x[2], x[5], x[8] } };
    float Rs[4][3][3], ts[4][3];
    int n = ap3p_compute_poses(fv, wp, Rs, ts);
    if (n == 0) return 0;
    const float* x4 = x + 9; const float* y4 = y + 6;
    int ns = 0; float min_reproj = 0;
    for (int i = 0; i < n; i++) { /* :360-372 */
        float X3p = Rs[i][0][0] * x4[0] + Rs[i][0][1] * x4[1] + Rs[i][0][2] * x4[2] + ts[i][0];
        float Y3p = Rs[i][1][0] * x4[0] + Rs[i][1][1] * x4[1] + Rs[i][1][2] * x4[2] + ts[i][1];
        float Z3p = Rs[i][2][0] * x4[0] + Rs[i][2][1] * x4[1] + Rs[i][2][2] * x4[2] + ts[i][2];
        float mu3p = cx + fx * X3p / Z3p, mv3p = cy + fy * Y3p / Z3p;
        float reproj = (mu3p - y4[0]) * (mu3p - y4[0]) + (mv3p - y4[1]) * (mv3p - y4[1]);
        if (i == 0 || min_reproj > reproj) { ns = i; min_reproj = reproj; }
    }
    memcpy(R9, Rs[ns], sizeof(float) * 9);
    memcpy(t3, ts[ns], sizeof(float) * 3);
    return 1;
}

/* ------------------------------------------------------------------ batched pose sampling
 * solve_batch_lambdatwist.cu:11-42 / solve_batch_ap3p.cu:331-378 */
void orc_solve_batch_p3p(const float* pts3, const float* pts2, float* rvecs, float* tvecs,
                         const float* K, int n_pts, int n_poses, int use_ap3p, int use_double) {
    const float fx = K[0], cx = K[2], fy = K[4], cy = K[5];
    const float qnan = __builtin_nanf("");
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < n_poses; idx++) {
        int id[4];
        orc_pose_sample_indices(idx, n_pts, id);
        float y[8], x[12], R[9], t[3];
        for (int k = 0; k < 4; k++) {
            y[k * 2] = pts2[id[k] * 2]; y[k * 2 + 1] = pts2[id[k] * 2 + 1];
            x[k * 3] = pts3[id[k] * 3]; x[k * 3 + 1] = pts3[id[k] * 3 + 1]; x[k * 3 + 2] = pts3[id[k] * 3 + 2];
        }
        int ok = use_ap3p ? orc_ap3p_p4p(y, x, fx, fy, cx, cy, R, t)
                          : orc_lambdatwist_p4p(y, x, fx, fy, cx, cy, use_double, R, t);
        if (!ok) {
            for (int k = 0; k < 3; k++) { rvecs[idx * 3 + k] = qnan; tvecs[idx * 3 + k] = qnan; }
            continue;
        }
        for (int k = 0; k < 3; k++) tvecs[idx * 3 + k] = t[k];
        orc_rodrigues(R, rvecs + idx * 3);
    }
}

/* same as orc_solve_batch_p3p but drawing from the maps (n_valid = number of finite map entries) */
void orc_solve_batch_p3p_maps(const float* p2_map, const float* p3_map, int npx, int n_valid, float* rvecs, float* tvecs,
                              const float* K, int n_poses, int use_ap3p, int use_double) {
    const float fx = K[0], cx = K[2], fy = K[4], cy = K[5];
    const float qnan = __builtin_nanf("");
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < n_poses; idx++) {
        int id[4];
        float y[8], x[12], R[9], t[3];
        int ok = n_valid >= 4 && orc_pose_sample_pixels(idx, npx, p2_map, id);
        if (ok) {
            for (int k = 0; k < 4; k++) {
                y[k * 2] = p2_map[(size_t)id[k] * 2]; y[k * 2 + 1] = p2_map[(size_t)id[k] * 2 + 1];
                x[k * 3] = p3_map[(size_t)id[k] * 3]; x[k * 3 + 1] = p3_map[(size_t)id[k] * 3 + 1]; x[k * 3 + 2] = p3_map[(size_t)id[k] * 3 + 2];
            }
            ok = use_ap3p ? orc_ap3p_p4p(y, x, fx, fy, cx, cy, R, t) : orc_lambdatwist_p4p(y, x, fx, fy, cx, cy, use_double, R, t);
        }
        if (!ok) {
            for (int k = 0; k < 3; k++) { rvecs[idx * 3 + k] = qnan; tvecs[idx * 3 + k] = qnan; }
            continue;
        }
        for (int k = 0; k < 3; k++) tvecs[idx * 3 + k] = t[k];
        orc_rodrigues(R, rvecs + idx * 3);
    }
}

/* ------------------------------------------------------------------ mean-shift
 * meanshift.cu:12-31 kernel, :34-150 host loop.  Sums follow the reference's float shared-memory tree
 * (reduce_vector_sum.h:12-61) term by term -- per level, blocks of 512 rows: thread t starts from x[t] + x[t+256], then a
 * binary tree over strides 128..1; the block sums form the next level -- so that the fixed point, the iteration count and
 * the confidence are the reference's bit for bit (tests/test_oracle_vs_ref_kernels.py).  Init trials use orc_rng instead
 * of host rand() (:76). */
#define ORC_RED_BLOCK 256
static void ref_tree_sum(const float* x, int N, int dims, float* out) { /* x: [N][dims] */
    float* cur = malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * dims);
    memcpy(cur, x, sizeof(float) * (size_t)N * dims);
    int n = N;
    float s[ORC_RED_BLOCK];
    while (n > 1) {
        const int nb = (n + 2 * ORC_RED_BLOCK - 1) / (2 * ORC_RED_BLOCK);
        float* nxt = malloc(sizeof(float) * (size_t)nb * dims);
        for (int b = 0; b < nb; b++)
            for (int d = 0; d < dims; d++) {
                for (int t = 0; t < ORC_RED_BLOCK; t++) {
                    const int idx = b * 2 * ORC_RED_BLOCK + t;
                    s[t] = 0;
                    if (idx < n) { s[t] = cur[(size_t)idx * dims + d]; if (idx + ORC_RED_BLOCK < n) s[t] += cur[(size_t)(idx + ORC_RED_BLOCK) * dims + d]; }
                }
                for (int stride = ORC_RED_BLOCK / 2; stride >= 1; stride >>= 1)
                    for (int t = 0; t < stride; t++) s[t] += s[t + stride];
                nxt[(size_t)b * dims + d] = s[0];
            }
        free(cur); cur = nxt; n = nb;
    }
    for (int d = 0; d < dims; d++) out[d] = cur[d];
    free(cur);
}
static float ms_weights(const float* space, const float* mean, float kernel_var, int N, int dims, float* wsum_x) {
    float* wgt = malloc(sizeof(float) * (size_t)N);
    float* wx = wsum_x ? malloc(sizeof(float) * (size_t)N * dims) : NULL;
    for (int i = 0; i < N; i++) {
        float l2 = 0;
        for (int d = 0; d < dims; d++) { float df = space[i * dims + d] - mean[d]; l2 += df * df; }
        wgt[i] = om_expf(-l2 / (2 * kernel_var));
        if (wx) for (int d = 0; d < dims; d++) wx[(size_t)i * dims + d] = space[i * dims + d] * wgt[i];
    }
    float wsum;
    ref_tree_sum(wgt, N, 1, &wsum);
    if (wx) { ref_tree_sum(wx, N, dims, wsum_x); free(wx); }
    free(wgt);
    return wsum;
}
void orc_meanshift(const float* space, float kernel_var, float* io_mean, float* o_confidence,
                   int* used_iters, int use_external_init_mean, int N, int dims, float epsilon,
                   int max_iters, int max_init_trials, float good_init_confidence) {
    float c_mean[16];
    float sx[16];
    if (use_external_init_mean) memcpy(c_mean, io_mean, sizeof(float) * dims);
    else {
        float best = 0; int best_idx = -1;
        for (int trial = 0; trial < max_init_trials; trial++) { /* :75-95 */
            int idx_rand = (int)(orc_rng(233u, (uint32_t)trial, 0x4D53u) % (uint32_t)N);
            float wsum = ms_weights(space, space + idx_rand * dims, kernel_var, N, dims, NULL);
            if (wsum > best) { best = wsum; best_idx = idx_rand; }
            if (best > good_init_confidence * N) break;
        }
        if (best_idx < 0) best_idx = 0;
        memcpy(c_mean, space + best_idx * dims, sizeof(float) * dims);
    }
    if (used_iters) *used_iters = 0;
    for (int iter = 0; iter < max_iters; iter++) { /* :103-134 */
        float wsum = ms_weights(space, c_mean, kernel_var, N, dims, sx);
        float m[16];
        for (int d = 0; d < dims; d++) m[d] = sx[d] / wsum;
        if (o_confidence) *o_confidence = wsum / N;
        if (used_iters) *used_iters = iter + 1;
        float disp = 0;
        for (int d = 0; d < dims; d++) disp += (io_mean[d] - m[d]) * (io_mean[d] - m[d]); /* vs stale io_mean: SURVEY B-6 */
        disp = sqrtf(disp);
        for (int d = 0; d < dims; d++) io_mean[d] = m[d];
        if (disp < epsilon) break;
        memcpy(c_mean, io_mean, sizeof(float) * dims);
    }
}

/* ------------------------------------------------------------------ robust Gaussian
 * aux_funs.cpp:101-141 on cv::Matx66d: determinant / inverse by LU with partial pivoting
 * (OpenCV 3.4 core/src/lapack.cpp), Ledoit-Wolf shrinkage with fixed lambda. */
static double lu_inverse(const double* A, double* Ainv, int n) { /* returns det; Ainv only if det>0 */
    double a[36], b[36];
    memcpy(a, A, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) b[i * n + j] = (i == j);
    double det = 1;
    for (int i = 0; i < n; i++) {
        int k = i;
        for (int j = i + 1; j < n; j++) if (fabs(a[j * n + i]) > fabs(a[k * n + i])) k = j;
        if (fabs(a[k * n + i]) < DBL_EPSILON) return 0;
        if (k != i) {
            for (int j = 0; j < n; j++) { double t = a[i * n + j]; a[i * n + j] = a[k * n + j]; a[k * n + j] = t;
                                          t = b[i * n + j]; b[i * n + j] = b[k * n + j]; b[k * n + j] = t; }
            det = -det;
        }
        det *= a[i * n + i];
        double d = -1 / a[i * n + i];
        for (int j = i + 1; j < n; j++) {
            double alpha = a[j * n + i] * d;
            for (int c = i + 1; c < n; c++) a[j * n + c] += alpha * a[i * n + c];
            for (int c = 0; c < n; c++) b[j * n + c] += alpha * b[i * n + c];
        }
    }
    if (det > 0 && Ainv) {
        for (int i = n - 1; i >= 0; i--)
            for (int c = 0; c < n; c++) {
                double s = b[i * n + c];
                for (int k = i + 1; k < n; k++) s -= a[i * n + k] * b[k * n + c];
                b[i * n + c] = s / a[i * n + i];
            }
        memcpy(Ainv, b, sizeof(double) * n * n);
    }
    return det;
}

/* fit_robust_gaussian.cu:56-97 e_step, :101-286 host loop */
int orc_fit_robust_gaussian(const float* space, float* io_mean, float* io_covar, float trunc_sigma,
                            float covar_reg_lambda, float* o_density, int* used_iters, int N,
                            int dims, float epsilon, int max_iters) {
    if (dims > 6) return 2;
    const int dc = (dims * dims + dims) / 2;
    float ht_weight = 0, ht_mean[6], ht_covar[21], ht_covar_inv[21];
    double full[36], inv_full[36];
    for (int d = 0; d < dims; d++) ht_mean[d] = io_mean[d];
    for (int d1 = 0; d1 < dims; d1++)
        for (int d2 = 0; d2 <= d1; d2++) ht_covar[(d1 * d1 + d1) / 2 + d2] = io_covar[d1 * dims + d2];
    if (used_iters) *used_iters = 0;
    int iter, reliable = 1;
    for (iter = 0; iter < max_iters; iter++) {
        for (int d1 = 0; d1 < dims; d1++) /* covar_half_to_full :17-25 */
            for (int d2 = 0; d2 <= d1; d2++) {
                full[d1 * dims + d2] = (double)ht_covar[(d1 * d1 + d1) / 2 + d2];
                if (d1 != d2) full[d2 * dims + d1] = full[d1 * dims + d2];
            }
        if (iter > 0 && covar_reg_lambda > 0) { /* aux_funs.cpp:124-141 */
            double tr = 0; for (int d = 0; d < dims; d++) tr += full[d * dims + d];
            double m = tr / (double)dims, lam = (double)covar_reg_lambda;
            for (int i = 0; i < dims; i++)
                for (int j = 0; j < dims; j++)
                    full[i * dims + j] = lam * m * (i == j ? 1.0 : 0.0) + (1 - lam) * full[i * dims + j];
        }
        double det = lu_inverse(full, inv_full, dims);
        if (det <= 0) { reliable = 0; break; }
        for (int d1 = 0; d1 < dims; d1++)
            for (int d2 = 0; d2 <= d1; d2++) {
                ht_covar[(d1 * d1 + d1) / 2 + d2] = (float)full[d1 * dims + d2];
                ht_covar_inv[(d1 * d1 + d1) / 2 + d2] = (float)inv_full[d1 * dims + d2];
            }
        float prev_density = ht_weight / N;
        float* e_w = malloc(sizeof(float) * (size_t)N);
        float* e_x = malloc(sizeof(float) * (size_t)N * dims);
        float* e_c = malloc(sizeof(float) * (size_t)N * dc);
        for (int i = 0; i < N; i++) { /* e_step :56-97 */
            float diff[6];
            for (int d = 0; d < dims; d++) diff[d] = space[i * dims + d] - ht_mean[d];
            float z = 0;
            for (int d1 = 0; d1 < dims; d1++) {
                float tmp = 0;
                for (int d2 = 0; d2 < dims; d2++) {
                    if (d1 >= d2) tmp += ht_covar_inv[(d1 * d1 + d1) / 2 + d2] * diff[d2];
                    else tmp += ht_covar_inv[(d2 * d2 + d2) / 2 + d1] * diff[d2];
                }
                z += tmp * diff[d1];
            }
            z = sqrtf(z);
            const float wgt = z < trunc_sigma ? 1 : 0;
            e_w[i] = wgt;
            for (int d = 0; d < dims; d++) e_x[(size_t)i * dims + d] = wgt * space[i * dims + d];
            for (int d1 = 0; d1 < dims; d1++)
                for (int d2 = 0; d2 <= d1; d2++) e_c[(size_t)i * dc + (d1 * d1 + d1) / 2 + d2] = wgt * diff[d1] * diff[d2];
        }
        ref_tree_sum(e_w, N, 1, &ht_weight); /* m step :213-243, sums in the reference's tree order */
        if (getenv("ORC_PRINT_RG")) fprintf(stderr, "orc rg: iter %d gated %.0f of %d\n", iter, ht_weight, N);
        int stop = 0;
        if (!isfinite(ht_weight)) { reliable = 0; stop = 1; }
        else if (fabsf(ht_weight / N - prev_density) < epsilon) { reliable = 1; stop = 1; }
        if (!stop) {
            ref_tree_sum(e_x, N, dims, ht_mean);
            ref_tree_sum(e_c, N, dc, ht_covar);
            for (int d = 0; d < dims; d++) ht_mean[d] /= ht_weight;
            for (int k = 0; k < dc; k++) ht_covar[k] /= ht_weight;
        }
        free(e_w); free(e_x); free(e_c);
        if (stop) break;
    }
    if (reliable) {
        if (o_density) *o_density = ht_weight / N;
        if (used_iters) *used_iters = iter;
        for (int d1 = 0; d1 < dims; d1++)
            for (int d2 = 0; d2 <= d1; d2++) {
                io_covar[d1 * dims + d2] = ht_covar[(d1 * d1 + d1) / 2 + d2];
                io_covar[d2 * dims + d1] = io_covar[d1 * dims + d2];
            }
        for (int d = 0; d < dims; d++) io_mean[d] = ht_mean[d];
    }
    return reliable ? 0 : 1;
}
