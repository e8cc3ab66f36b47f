// tests/cxx/vk_testhooks.hip -- TEST LIBRARY (voldor_amd/lib/libvoldor_hip_test.so), not part of the product.
//
// (1) Host instantiations of the __host__ __device__ per-lane math of the product headers (vk_p3p.hpp, vk_device.hpp,
//     vk_strict_model.hpp), so the CPU-only test tier can compare the exact source the GPU lanes run with the oracle and
//     the reference's golden vectors.
// (2) Device-vs-host probes: the same functions evaluated by a trivial kernel, one lane per element.  Strict mode rests on
//     "IEEE + - * / sqrt and conversions give the same bits on gfx950 and on the host"; these probes test exactly that on
//     the GPU box (tests/test_gpu_strict.py), and they locate where a device build starts to differ from the host build.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../voldor_amd/csrc/vk_p3p.hpp"
#include "../../voldor_amd/csrc/vk_device.hpp"
#include "../../voldor_amd/csrc/vk_strict_math.h"
#include "../../voldor_amd/csrc/vk_strict_model.hpp"
#include "../../voldor_amd/csrc/vk_ref_cv.h"

#define VKT_API extern "C" __attribute__((visibility("default")))

// ---- (1) host instantiations ------------------------------------------------------------------------------------
VKT_API int vk_host_lambdatwist_p4p(const float* y8, const float* x12, float fx, float fy, float cx, float cy, int use_double,
                                    float* R9, float* t3) {
    float yu[4], yv[4], xp[4][3];
    for (int k = 0; k < 4; k++) { yu[k] = y8[k * 2]; yv[k] = y8[k * 2 + 1]; for (int d = 0; d < 3; d++) xp[k][d] = x12[k * 3 + d]; }
    bool ok = use_double ? vk::lambdatwist_p4p<double>(yu, yv, xp, fx, fy, cx, cy, R9, t3)
                         : vk::lambdatwist_p4p<float>(yu, yv, xp, fx, fy, cx, cy, R9, t3);
    return ok ? 1 : 0;
}
VKT_API int vk_host_ap3p_p4p(const float* y8, const float* x12, float fx, float fy, float cx, float cy, float* R9, float* t3) {
    float yu[4], yv[4], xp[4][3];
    for (int k = 0; k < 4; k++) { yu[k] = y8[k * 2]; yv[k] = y8[k * 2 + 1]; for (int d = 0; d < 3; d++) xp[k][d] = x12[k * 3 + d]; }
    return vk::ap3p_p4p(yu, yv, xp, fx, fy, cx, cy, R9, t3) ? 1 : 0;
}
VKT_API void vk_host_rodrigues(const float* R9, float* rvec3) {
    float R[9];
    for (int i = 0; i < 9; i++) R[i] = R9[i];
    vk::nearest_rotation(R);
    vk::rotmat_to_angle_axis(R, rvec3);
}
VKT_API void vk_host_rvec_to_rotmat(const float* rvec3, float* R9) { vk::angle_axis_to_rotmat(rvec3, R9); }
VKT_API unsigned vk_host_rng(unsigned seed, unsigned stream, unsigned counter) { return vk::rng3(seed, stream, counter); }
VKT_API float vk_host_u01(unsigned r) { return vk::u01(r); }

// ---- (2) elementary probes: out[i] = op(a[i], b[i]) as double (float results widen exactly) ------------------------
__host__ __device__ static double probe_op(int op, float a, float b) {
#pragma clang fp contract(off)
    const double da = (double)a, db = (double)b;
    switch (op) {
        case 0: return (double)(a / b);
        case 1: return (double)sqrtf(fabsf(a));
        case 2: return da / db;
        case 3: return ::sqrt(fabs(da));
        case 4: return (double)(float)(1.0 / ::sqrt(da * da + db * db + 1.0));
        case 5: return (double)(a * b + a);
        case 6: return (double)vsm_expf(a);
        case 7: return (double)vsm_logf(fabsf(a));
        case 8: return (double)vsm_powf(fabsf(a), b);
        case 9: return (double)vsm_atan2f(a, b);
        case 10: return (double)vsm_sinf(a);
        case 11: return (double)vsm_cosf(a);
        case 12: return (double)vsm_cbrtf(a);
        case 13: return (double)floorf(a);
        case 14: return (double)(float)(int)a;
        case 15: return (double)(1.0f / a);
        case 16: return vsm_exp(da);
        case 17: return vsm_log(fabs(da));
        case 18: return (double)(float)(da * db);                       // double product rounded to float
        case 19: return (double)((float)(da) * 0.5);                    // float * double literal
        case 20: return (double)vk::strict::rigidness(a, b, a * 0.9f + 0.1f, b * 1.1f - 0.05f, 0.15f, 1.f);
        case 21: return (double)vk::strict::depth_rigidness(fabsf(a) + 0.1f, fabsf(b) + 0.1f, 160.f, 0.15f, 1.f);
        case 22: return (double)(-1.5f * vsm_logf(vk::strict::rigidness(a, b, a * 0.9f + 0.1f, b * 1.1f - 0.05f, 0.15f, 1.f)));
        case 23: return (double)(fabsf(a) / (2.f * fabsf(b) + 1e-3f));
        case 24: return (double)sqrtf(a * a + b * b);
        case 25: return (double)vk::strict::rig_core(fabsf(a), fabsf(b), 0.15f);        // the straight-line model and its call-by-call form: equal bits, device == host
        case 26: return (double)vk::strict::rig_core_plain(fabsf(a), fabsf(b), 0.15f);
        default: return 0.0;
    }
}
__global__ static void k_probe(int op, const float* a, const float* b, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = probe_op(op, a[i], b[i]);
}
VKT_API int vkt_probe_ops_count(void) { return 27; }
VKT_API void vkt_probe_host(int op, const float* a, const float* b, double* out, int n) {
    for (int i = 0; i < n; i++) out[i] = probe_op(op, a[i], b[i]);
}
VKT_API int vkt_probe_device(int op, const float* a, const float* b, double* out, int n) {
    float *da = nullptr, *db = nullptr; double* dout = nullptr;
    if (hipMalloc(&da, sizeof(float) * n) != hipSuccess || hipMalloc(&db, sizeof(float) * n) != hipSuccess ||
        hipMalloc(&dout, sizeof(double) * n) != hipSuccess) return 1;
    (void)hipMemcpy(da, a, sizeof(float) * n, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b, sizeof(float) * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3((n + 255) / 256), dim3(256), 0, 0, op, da, db, dout, n);
    const int rc = (int)hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return rc;
}

// ---- LambdaTwist: sequential path (`only` = -1, what the host build runs) on the device, one lane per 4-tuple, with the
// intermediate values of vk_p3p.hpp's debug taps.  y8[n][8], x12[n][12] -> ok[n], R9[n][9], t3[n][3], dbg[n][96]
constexpr int VKT_DBG = 96;
template <typename S>
__host__ __device__ static void p4p_one(const float* y8, const float* x12, float fx, float fy, float cx, float cy, int* ok, float* R9, float* t3,
                                        double* dbg) {
    float yu[4], yv[4], xp[4][3];
    for (int k = 0; k < 4; k++) { yu[k] = y8[k * 2]; yv[k] = y8[k * 2 + 1]; for (int d = 0; d < 3; d++) xp[k][d] = x12[k * 3 + d]; }
    S d[VKT_DBG];
    for (int i = 0; i < VKT_DBG; i++) d[i] = S(0);
    float R[9] = { 0 }, t[3] = { 0 };
    *ok = vk::lambdatwist_p4p<S>(yu, yv, xp, fx, fy, cx, cy, R, t, -1, (S*)nullptr, d) ? 1 : 0;
    for (int i = 0; i < 9; i++) R9[i] = R[i];
    for (int i = 0; i < 3; i++) t3[i] = t[i];
    for (int i = 0; i < VKT_DBG; i++) dbg[i] = (double)d[i];
}
template <typename S>
__global__ static void k_p4p(const float* y8, const float* x12, float fx, float fy, float cx, float cy, int* ok, float* R9, float* t3, double* dbg,
                             int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p4p_one<S>(y8 + (size_t)i * 8, x12 + (size_t)i * 12, fx, fy, cx, cy, ok + i, R9 + (size_t)i * 9, t3 + (size_t)i * 3, dbg + (size_t)i * VKT_DBG);
}
VKT_API int vkt_p4p_dbg_len(void) { return VKT_DBG; }
VKT_API void vkt_p4p_host(const float* y8, const float* x12, int n, float fx, float fy, float cx, float cy, int use_double, int* ok, float* R9,
                          float* t3, double* dbg) {
    for (int i = 0; i < n; i++) {
        if (use_double) p4p_one<double>(y8 + (size_t)i * 8, x12 + (size_t)i * 12, fx, fy, cx, cy, ok + i, R9 + (size_t)i * 9, t3 + (size_t)i * 3, dbg + (size_t)i * VKT_DBG);
        else p4p_one<float>(y8 + (size_t)i * 8, x12 + (size_t)i * 12, fx, fy, cx, cy, ok + i, R9 + (size_t)i * 9, t3 + (size_t)i * 3, dbg + (size_t)i * VKT_DBG);
    }
}
VKT_API int vkt_p4p_device(const float* y8, const float* x12, int n, float fx, float fy, float cx, float cy, int use_double, int* ok, float* R9,
                           float* t3, double* dbg) {
    float *dy = nullptr, *dx = nullptr, *dR = nullptr, *dt = nullptr; int* dok = nullptr; double* dd = nullptr;
    if (hipMalloc(&dy, sizeof(float) * 8 * n) || hipMalloc(&dx, sizeof(float) * 12 * n) || hipMalloc(&dR, sizeof(float) * 9 * n) ||
        hipMalloc(&dt, sizeof(float) * 3 * n) || hipMalloc(&dok, sizeof(int) * n) || hipMalloc(&dd, sizeof(double) * VKT_DBG * n)) return 1;
    (void)hipMemcpy(dy, y8, sizeof(float) * 8 * n, hipMemcpyHostToDevice);
    (void)hipMemcpy(dx, x12, sizeof(float) * 12 * n, hipMemcpyHostToDevice);
    if (use_double) hipLaunchKernelGGL(k_p4p<double>, dim3((n + 63) / 64), dim3(64), 0, 0, dy, dx, fx, fy, cx, cy, dok, dR, dt, dd, n);
    else hipLaunchKernelGGL(k_p4p<float>, dim3((n + 63) / 64), dim3(64), 0, 0, dy, dx, fx, fy, cx, cy, dok, dR, dt, dd, n);
    int rc = (int)hipMemcpy(ok, dok, sizeof(int) * n, hipMemcpyDeviceToHost);
    rc |= (int)hipMemcpy(R9, dR, sizeof(float) * 9 * n, hipMemcpyDeviceToHost);
    rc |= (int)hipMemcpy(t3, dt, sizeof(float) * 3 * n, hipMemcpyDeviceToHost);
    rc |= (int)hipMemcpy(dbg, dd, sizeof(double) * VKT_DBG * n, hipMemcpyDeviceToHost);
    (void)hipFree(dy); (void)hipFree(dx); (void)hipFree(dR); (void)hipFree(dt); (void)hipFree(dok); (void)hipFree(dd);
    return rc;
}
// rotation matrix -> nearest rotation -> angle axis (strict = software atan2), device vs host
__global__ static void k_rodrigues(const float* R9, float* rv, int n, int strict) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float R[9];
    for (int k = 0; k < 9; k++) R[k] = R9[(size_t)i * 9 + k];
    vk::nearest_rotation(R);
    vk::rotmat_to_angle_axis(R, rv + (size_t)i * 3, strict != 0);
}
VKT_API void vkt_rodrigues_host(const float* R9, float* rv, int n, int strict) {
    for (int i = 0; i < n; i++) {
        float R[9];
        for (int k = 0; k < 9; k++) R[k] = R9[(size_t)i * 9 + k];
        vk::nearest_rotation(R);
        vk::rotmat_to_angle_axis(R, rv + (size_t)i * 3, strict != 0);
    }
}
VKT_API int vkt_rodrigues_device(const float* R9, float* rv, int n, int strict) {
    float *dR = nullptr, *dv = nullptr;
    if (hipMalloc(&dR, sizeof(float) * 9 * n) || hipMalloc(&dv, sizeof(float) * 3 * n)) return 1;
    (void)hipMemcpy(dR, R9, sizeof(float) * 9 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_rodrigues, dim3((n + 63) / 64), dim3(64), 0, 0, dR, dv, n, strict);
    const int rc = (int)hipMemcpy(rv, dv, sizeof(float) * 3 * n, hipMemcpyDeviceToHost);
    (void)hipFree(dR); (void)hipFree(dv);
    return rc;
}
// strict residual model on the host (vs the oracle in strict mode, CPU tier)
// cv::Rodrigues(matrix -> vector) as reference mode applies it to a camera's float matrix (vk_ref_cv.h): host build and gfx950 build
VKT_API void vkt_cv_rvec_of_R_host(const float* R9, float* rv, int n, int strict) {
    for (int i = 0; i < n; i++) vrcv_rvec_of_R32(R9 + 9 * i, rv + 3 * i, strict);
}
__global__ static void k_cv_rvec_of_R(const float* R9, float* rv, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) vrcv_rvec_of_R32(R9 + 9 * i, rv + 3 * i, 1);
}
VKT_API int vkt_cv_rvec_of_R_device(const float* R9, float* rv, int n) {
    float *dR = nullptr, *dv = nullptr;
    if (hipMalloc(&dR, sizeof(float) * 9 * n) || hipMalloc(&dv, sizeof(float) * 3 * n)) return 1;
    (void)hipMemcpy(dR, R9, sizeof(float) * 9 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_cv_rvec_of_R, dim3((n + 63) / 64), dim3(64), 0, 0, dR, dv, n);
    const int rc = (int)hipMemcpy(rv, dv, sizeof(float) * 3 * n, hipMemcpyDeviceToHost);
    (void)hipFree(dR); (void)hipFree(dv);
    return rc;
}
VKT_API float vkt_strict_rigidness(float dx1, float dy1, float dx2, float dy2, float lambda, float abs_rf) {
    return vk::strict::rigidness(dx1, dy1, dx2, dy2, lambda, abs_rf);
}
VKT_API float vkt_strict_depth_rigidness(float d1, float d2, float basefocal, float omega, float abs_rf) {
    return vk::strict::depth_rigidness(d1, d2, basefocal, omega, abs_rf);
}
VKT_API void vkt_strict_rvec_to_rotmat(const float* rv, float* R9) { vk::angle_axis_to_rotmat(rv, R9, true); }

// ---- the fp32 pre-filter of the strict passes against the strict residual model, same inputs (vk_device.hpp filt_neglog): out_s = -vsm_logf(strict::rigidness),
// out_f = the filter's value, one lane per element
__global__ static void k_filter_pair(const float* dx1, const float* dy1, const float* ox, const float* oy, int n, float lambda, float arf, float* out_s, float* out_f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float ia = 1.f / arf;
    const vk::ObsTerms T = vk::obs_terms(ox[i], oy[i], ia * ia, __builtin_amdgcn_logf(0.25f * lambda * lambda));
    out_f[i] = vk::filt_neglog(T, dx1[i], dy1[i], ox[i], oy[i], 0.25f * ia * ia);
    out_s[i] = -vsm_logf(vk::strict::rigidness(dx1[i], dy1[i], ox[i], oy[i], lambda, arf));
}
VKT_API int vkt_filter_pair_device(const float* dx1, const float* dy1, const float* ox, const float* oy, int n, float lambda, float arf, float* out_s, float* out_f) {
    float* d[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    for (int k = 0; k < 6; k++) if (hipMalloc(&d[k], sizeof(float) * n) != hipSuccess) return 1;
    const float* src[4] = { dx1, dy1, ox, oy };
    for (int k = 0; k < 4; k++) (void)hipMemcpy(d[k], src[k], sizeof(float) * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_filter_pair, dim3((n + 255) / 256), dim3(256), 0, 0, d[0], d[1], d[2], d[3], n, lambda, arf, d[4], d[5]);
    int rc = (int)hipMemcpy(out_s, d[4], sizeof(float) * n, hipMemcpyDeviceToHost);
    rc |= (int)hipMemcpy(out_f, d[5], sizeof(float) * n, hipMemcpyDeviceToHost);
    for (int k = 0; k < 6; k++) (void)hipFree(d[k]);
    return rc;
}

// ---- pow_m2 (vk_strict_model.hpp): the shortcut against the plain call, host build; returns the number of elements whose bits differ, *n_fast = the ones the shortcut answered
VKT_API long vkt_pow_m2_host(const float* x, long n, long* n_fast) {
    long bad = 0, fast = 0;
    for (long i = 0; i < n; i++) {
        const float a = vk::strict::pow_m2(x[i]), b = vsm_powf(x[i], -2.f);
        unsigned ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
        if (ua != ub && !(a != a && b != b)) bad++;
        if (x[i] >= 1.0f && x[i] <= 1e18f) { const double xd = (double)x[i]; if (vk::strict::rounds_alike(1.0 / (xd * xd))) fast++; }
    }
    *n_fast = fast;
    return bad;
}
// the largest distance, in units of the last place of the shortcut's double, between the shortcut's double and the plain call's (the margin of 2^17 rests on it)
VKT_API double vkt_pow_m2_gap_host(const float* x, long n) {
    double worst = 0.0;
    for (long i = 0; i < n; i++) {
        if (!(x[i] >= 1.0f && x[i] <= 1e18f)) continue;
        const double xd = (double)x[i], y = 1.0 / (xd * xd), l = vsm_log(xd), f = l == 0.0 ? 1.0 : vsm_exp(-2.0 * l);
        const double gap = (double)(long long)(vsm_bits(y) - vsm_bits(f));
        if (fabs(gap) > worst) worst = fabs(gap);
    }
    return worst;
}

// the same for a depth prior's term: -vsm_logf(strict::depth_rigidness(d1, d2)) against ln2 log2(1 + depth_ratio(d1, d2))
__global__ static void k_filter_pair_depth(const float* d1, const float* d2, int n, float basefocal, float omega, float arf, float* out_s, float* out_f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_f[i] = 0.6931471805599453f * vk::fast_log2(1.f + vk::depth_ratio(d1[i], d2[i], basefocal, omega, 1.f / arf));
    out_s[i] = -vsm_logf(vk::strict::depth_rigidness(d1[i], d2[i], basefocal, omega, arf));
}
VKT_API int vkt_filter_pair_depth_device(const float* d1, const float* d2, int n, float basefocal, float omega, float arf, float* out_s, float* out_f) {
    float* d[4] = { nullptr, nullptr, nullptr, nullptr };
    for (int k = 0; k < 4; k++) if (hipMalloc(&d[k], sizeof(float) * n) != hipSuccess) return 1;
    (void)hipMemcpy(d[0], d1, sizeof(float) * n, hipMemcpyHostToDevice);
    (void)hipMemcpy(d[1], d2, sizeof(float) * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_filter_pair_depth, dim3((n + 255) / 256), dim3(256), 0, 0, d[0], d[1], n, basefocal, omega, arf, d[2], d[3]);
    int rc = (int)hipMemcpy(out_s, d[2], sizeof(float) * n, hipMemcpyDeviceToHost);
    rc |= (int)hipMemcpy(out_f, d[3], sizeof(float) * n, hipMemcpyDeviceToHost);
    for (int k = 0; k < 4; k++) (void)hipFree(d[k]);
    return rc;
}

// rig_core (vk_strict_model.hpp: the residual model in one straight line) against rig_core_plain (call by call), host build: elements whose bits differ
VKT_API long vkt_rig_core_host(const float* mag, const float* diff, const float* k, long n) {
    long bad = 0;
    for (long i = 0; i < n; i++) {
        const float a = vk::strict::rig_core(mag[i], diff[i], k[i]), b = vk::strict::rig_core_plain(mag[i], diff[i], k[i]);
        unsigned ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
        if (ua != ub && !(a != a && b != b)) bad++;
    }
    return bad;
}
