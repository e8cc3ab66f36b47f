// voldor_amd/csrc/vk_lu.hpp -- 6x6 LU inverse on one wave, rows in lanes (shared by the fast mode kernels of vk_pose.hip and the strict
// mode kernel of vk_strict.hip): bit-identical to the serial algorithm behind cv::determinant / cv::Matx::inv (aux_funs.cpp:101-118).
#pragma once
#include <hip/hip_runtime.h>

namespace vk {

// n x n (n<=6) inverse + determinant in double by LU with partial pivoting, the algorithm behind
// cv::determinant / cv::Matx::inv that the reference calls on the host every iteration
// (aux_funs.cpp:101-118).  Lane r (< n) of ONE wave holds row r of A and of the right-hand side B
// (starts as I) in registers; pivot columns and pivot rows are broadcast with v_readlane (the
// source lane is wave-uniform), so there is no LDS or scratch traffic on the critical path.  Within
// an elimination step every element update is independent, hence each element sees exactly the
// operation sequence of the serial algorithm: the result is bit-identical to it.
// Must be called by all 64 lanes of the wave; lanes >= n carry don't-care rows.
__device__ __forceinline__ double readlane_d(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lu_inverse_rows(double (&a)[6], double (&b)[6], int n) {
    const int lane = threadIdx.x & 63;
    double det = 1.0;
    bool singular = false;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        if (i < n && !singular) {
            int k = i;
            double best = fabs(readlane_d(a[i], i));
#pragma unroll
            for (int j = i + 1; j < 6; j++) {
                if (j < n) {
                    const double v = fabs(readlane_d(a[i], j));
                    if (v > best) { best = v; k = j; }
                }
            }
            if (best < 2.220446049250313e-16) singular = true;
            else {
                if (k != i) {  // row swap i <-> k
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        const double ai = readlane_d(a[c], i), ak = readlane_d(a[c], k);
                        const double bi = readlane_d(b[c], i), bk = readlane_d(b[c], k);
                        if (lane == i) { a[c] = ak; b[c] = bk; }
                        else if (lane == k) { a[c] = ai; b[c] = bi; }
                    }
                    det = -det;
                }
                const double piv = readlane_d(a[i], i);
                det *= piv;
                const double d = -1.0 / piv;
                const double alpha = a[i] * d;  // own row, column i
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    const double ric = readlane_d(a[c], i), bic = readlane_d(b[c], i);
                    if (lane > i && lane < n && c < n) {
                        if (c > i) a[c] += alpha * ric;
                        b[c] += alpha * bic;
                    }
                }
            }
        }
    }
    if (singular) return 0.0;
    if (det > 0.0) {
#pragma unroll
        for (int i = 5; i >= 0; i--) {  // back substitution, row i lives in lane i
            if (i < n) {
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double sacc = b[c];
#pragma unroll
                    for (int k = i + 1; k < 6; k++) {
                        if (k < n) sacc -= a[k] * readlane_d(b[c], k);
                    }
                    const double q = sacc / a[i];
                    if (lane == i && c < n) b[c] = q;
                }
            }
        }
    }
    return det;
}

// One robust-Gaussian "prepare" step (fit_robust_gaussian.cu:172-205): packed half -> full, Ledoit-
// Wolf shrinkage with fixed lambda (aux_funs.cpp:124-141), inverse; writes the (regularised) covariance
// and its inverse back in packed form.  Called by all lanes of wave 0; returns false when det <= 0.
__device__ static bool rg_prepare_wave(float* covar_half, float* cinv_half, int dims, bool regularise, float lambda) {
    const int lane = threadIdx.x & 63;
    const int r = lane < dims ? lane : 0;
    double a[6], b[6];
    double tr = 0;
#pragma unroll
    for (int d = 0; d < 6; d++) if (d < dims) tr += (double)covar_half[(d * d + d) / 2 + d];
    const double m = tr / (double)dims, lam = (double)lambda;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        const int hi = r >= c ? r : c, lo = r >= c ? c : r;
        double full = (c < dims) ? (double)covar_half[(hi * hi + hi) / 2 + lo] : 0.0;
        if (regularise) full = lam * m * (r == c ? 1.0 : 0.0) + (1 - lam) * full;
        a[c] = full;
        b[c] = (r == c) ? 1.0 : 0.0;
    }
    double keep[6];
#pragma unroll
    for (int c = 0; c < 6; c++) keep[c] = a[c];
    const double det = lu_inverse_rows(a, b, dims);
    if (det <= 0) return false;
    if (lane < dims) {
#pragma unroll
        for (int c = 0; c < 6; c++) {
            if (c <= r && c < dims) {
                covar_half[(r * r + r) / 2 + c] = (float)keep[c];
                cinv_half[(r * r + r) / 2 + c] = (float)b[c];
            }
        }
    }
    return true;
}

// The same inverse with the RIGHT-HAND SIDE dealt out as well: lane 6 r + c (< 36) holds row r of A (six values, shared by the six lanes of the
// row) and ONE element B[r][c].  lu_inverse_rows has every lane update all six columns of B and, in the back substitution, perform all 36 fp64
// divisions although it keeps six of them (~3500 instructions per call: most of a strict refit iteration, measured round 4); here a lane updates
// its own element (the pivot row's element of its column arrives by ds_bpermute) and divides once per level.  Every element still sees exactly the
// operation sequence of the serial algorithm: bit-identical (tests/test_gpu_strict.py holds the strict kernels that use it to the serial LU).
// Must be called by all 64 lanes of the wave.  a: row r of A (in), bb: B[r][c] (in: identity; out: the inverse's element); returns det.
__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ double lu_inverse_rc(double (&a)[6], double& bb, int n) {
    const int lane = threadIdx.x & 63, r = lane / 6, c = lane % 6;
    const bool in = lane < 36 && r < n && c < n;
    double det = 1.0;
    bool singular = false;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        if (i < n && !singular) {
            int k = i;
            double best = fabs(readlane_d(a[i], 6 * i));
#pragma unroll
            for (int j = i + 1; j < 6; j++) {
                if (j < n) {
                    const double v = fabs(readlane_d(a[i], 6 * j));
                    if (v > best) { best = v; k = j; }
                }
            }
            if (best < 2.220446049250313e-16) singular = true;
            else {
                if (k != i) {  // row swap i <-> k (wave-uniform decision)
#pragma unroll
                    for (int cc = 0; cc < 6; cc++) {
                        const double ai = readlane_d(a[cc], 6 * i), ak = readlane_d(a[cc], 6 * k);
                        if (r == i) a[cc] = ak; else if (r == k) a[cc] = ai;
                    }
                    const double bi = shfl_d(bb, 6 * i + c), bk = shfl_d(bb, 6 * k + c);
                    if (r == i) bb = bk; else if (r == k) bb = bi;
                    det = -det;
                }
                const double piv = readlane_d(a[i], 6 * i);
                det *= piv;
                const double d = -1.0 / piv;
                const double alpha = a[i] * d;  // own row, column i
                const double bic = shfl_d(bb, 6 * i + c);
#pragma unroll
                for (int cc = 0; cc < 6; cc++) {
                    const double ric = readlane_d(a[cc], 6 * i);
                    if (r > i && r < n && cc < n && cc > i) a[cc] += alpha * ric;
                }
                if (in && r > i) bb += alpha * bic;
            }
        }
    }
    if (singular) return 0.0;
    if (det > 0.0) {
#pragma unroll
        for (int i = 5; i >= 0; i--) {  // back substitution, level i: the elements of row i
            if (i < n) {
                double sacc = bb;
#pragma unroll
                for (int k = i + 1; k < 6; k++) {
                    const double bkc = shfl_d(bb, 6 * k + c);  // the solved element (k, c)
                    if (k < n) sacc -= a[k] * bkc;
                }
                const double q = sacc / a[i];
                if (in && r == i) bb = q;
            }
        }
    }
    return det;
}
// rg_prepare_wave on that layout (strict mode kernel, vk_strict.hip)
__device__ static bool rg_prepare_wave_rc(float* covar_half, float* cinv_half, int dims, bool regularise, float lambda) {
    const int lane = threadIdx.x & 63;
    const int r = lane < 36 ? lane / 6 : 0, c0 = lane < 36 ? lane % 6 : 0;
    const int rr = r < dims ? r : 0;
    double a[6];
    double tr = 0;
#pragma unroll
    for (int d = 0; d < 6; d++) if (d < dims) tr += (double)covar_half[(d * d + d) / 2 + d];
    const double m = tr / (double)dims, lam = (double)lambda;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        const int hi = rr >= c ? rr : c, lo = rr >= c ? c : rr;
        double full = (c < dims) ? (double)covar_half[(hi * hi + hi) / 2 + lo] : 0.0;
        if (regularise) full = lam * m * (rr == c ? 1.0 : 0.0) + (1 - lam) * full;
        a[c] = full;
    }
    double keep = 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) if (c == c0) keep = a[c];
    double bb = (rr == c0) ? 1.0 : 0.0;
    const double det = lu_inverse_rc(a, bb, dims);
    if (det <= 0) return false;
    if (lane < 36 && r < dims && c0 <= r) {
        covar_half[(r * r + r) / 2 + c0] = (float)keep;
        cinv_half[(r * r + r) / 2 + c0] = (float)bb;
    }
    return true;
}

}  // namespace vk
