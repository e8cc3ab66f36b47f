/* voldor_amd/csrc/vk_ref_cv.h -- cv::Rodrigues, matrix -> vector, as the reference's host code applies it to a camera's float rotation
 * matrix: Camera::rvec() (voldor/utils.h:49-53), called for the START of every mean shift (voldor/geometry.cpp:183-185; the start also
 * enters the first displacement test, SURVEY B-6) and for the poses the window returns (Camera::pose6(), utils.h:44-47, py_export.cpp:60).
 *
 * Round 4.  Rounds 1-3 kept the rotation VECTOR a pose update produced and used it as the next start and as the output; the reference
 * keeps only the float MATRIX and converts back each time -- a round trip vector -> float matrix -> vector that moves a component by an
 * ulp or two.  A converging mean shift forgets its start, which is why whole windows were bit-identical all the same and the returned
 * poses agreed to 1e-9; a mean shift that runs into its iteration cap (a camera looking at noise: the truncated windows) does not, and
 * the windows drift apart from there.  Reference mode now takes the round trip (strict kernels in vk_strict.hip, the oracle, and the
 * OpenCV stand-in of the emulated reference all call THIS header), and the returned poses are the reference's bit for bit.
 *
 * OpenCV is not part of the reference tree: this is the calib3d algorithm restated from its documented behaviour (orthonormalise the
 * matrix, theta = acos((trace - 1) / 2), axis from the antisymmetric part, the theta ~ pi branch from the diagonal), in double -- the
 * same text that oracle/ref_stubs/minicv carried in rounds 2-3, moved here so that three users share one copy.  The orthonormalisation
 * is the polar factor by Newton's iteration where OpenCV runs an SVD: equal as real numbers, not pinned against OpenCV's bits
 * ("relative to the restatement").  strict != 0: sin / cos / acos from vk_strict_math.h (acos as atan2(sqrt((1 - c)(1 + c)), c));
 * strict == 0 (host only): the C library's.  Plain C: hipcc, gcc -std=gnu11 (oracle) and g++ (reference build) include it. */
#ifndef VK_REF_CV_H
#define VK_REF_CV_H
#include <math.h>
#include <float.h>
#include "vk_strict_math.h"

#if defined(__HIPCC__)
#define VRCV_FN __host__ __device__ static inline
#else
#define VRCV_FN static inline
#endif
#if defined(__clang__)
#define VRCV_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define VRCV_NO_CONTRACT /* gcc: the oracle and the reference build are compiled with -ffp-contract=off */
#endif

VRCV_FN double vrcv_cos(double x, int strict) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)strict; return vsm_cos(x);  /* the device has no C library: reference mode only */
#else
    return strict ? vsm_cos(x) : cos(x);
#endif
}
VRCV_FN double vrcv_sin(double x, int strict) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)strict; return vsm_sin(x);
#else
    return strict ? vsm_sin(x) : sin(x);
#endif
}
VRCV_FN double vrcv_acos(double c, int strict) { /* c in [-1, 1] */
    VRCV_NO_CONTRACT
#if defined(__HIP_DEVICE_COMPILE__)
    (void)strict; return vsm_atan2(sqrt((1. - c) * (1. + c)), c);
#else
    return strict ? vsm_atan2(sqrt((1. - c) * (1. + c)), c) : acos(c);
#endif
}

/* vector -> matrix (cvRodrigues2, double) */
VRCV_FN void vrcv_rvec_to_R(const double r_in[3], double R[9], int strict) {
    VRCV_NO_CONTRACT
    double rx = r_in[0], ry = r_in[1], rz = r_in[2];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) { for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.; return; }
    const double c = vrcv_cos(theta, strict), s = vrcv_sin(theta, strict), c1 = 1. - c, it = 1. / theta;
    rx *= it; ry *= it; rz *= it;
    const double rrt[9] = { rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz };
    const double rxm[9] = { 0, -rz, ry, rz, 0, -rx, -ry, rx, 0 };
    for (int i = 0; i < 9; i++) R[i] = c * (i % 4 == 0 ? 1. : 0.) + c1 * rrt[i] + s * rxm[i];
}
/* R <- U V^T of its SVD, as the polar factor (Newton: X <- (X + X^-T) / 2) */
VRCV_FN void vrcv_orthonormalise(double X[9]) {
    VRCV_NO_CONTRACT
    for (int it = 0; it < 40; it++) {
        const double d = X[0] * (X[4] * X[8] - X[5] * X[7]) - X[1] * (X[3] * X[8] - X[5] * X[6]) + X[2] * (X[3] * X[7] - X[4] * X[6]);
        if (d == 0) return;
        const double id = 1. / d;
        const double invT[9] = { (X[4] * X[8] - X[5] * X[7]) * id, (X[5] * X[6] - X[3] * X[8]) * id, (X[3] * X[7] - X[4] * X[6]) * id,
                                 (X[2] * X[7] - X[1] * X[8]) * id, (X[0] * X[8] - X[2] * X[6]) * id, (X[1] * X[6] - X[0] * X[7]) * id,
                                 (X[1] * X[5] - X[2] * X[4]) * id, (X[2] * X[3] - X[0] * X[5]) * id, (X[0] * X[4] - X[1] * X[3]) * id };
        double delta = 0;
        for (int i = 0; i < 9; i++) { const double n = 0.5 * (X[i] + invT[i]); const double e = fabs(n - X[i]); delta = e > delta ? e : delta; X[i] = n; }
        if (delta < 1e-16) break;
    }
}
/* matrix -> vector */
VRCV_FN void vrcv_R_to_rvec(const double R_in[9], double r[3], int strict) {
    VRCV_NO_CONTRACT
    double R[9];
    for (int i = 0; i < 9; i++) R[i] = R_in[i];
    vrcv_orthonormalise(R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    const double theta = vrcv_acos(c, strict);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; rx = sqrt(t > 0. ? t : 0.);
        t = (R[4] + 1) * 0.5; ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
        if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
        const double k = theta / sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = rx * k; r[1] = ry * k; r[2] = rz * k;
        return;
    }
    const double vth = 1. / (2 * s) * theta;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}
/* Rodrigues(Mat R (3x3, CV_32F), Vec3f& rvec): what Camera::rvec() returns */
VRCV_FN void vrcv_rvec_of_R32(const float R32[9], float rvec[3], int strict) {
    double Rd[9], r[3];
    for (int i = 0; i < 9; i++) Rd[i] = (double)R32[i];
    vrcv_R_to_rvec(Rd, r, strict);
    for (int i = 0; i < 3; i++) rvec[i] = (float)r[i];
}
#endif
