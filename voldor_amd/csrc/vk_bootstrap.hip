// voldor_amd/csrc/vk_bootstrap.hip -- monocular bootstrap (voldor/voldor.cpp:151-162):
// estimate_camera_pose_epipolar + estimate_depth_closed_form (voldor/geometry.cpp:267-332).
//
// The reference calls OpenCV 3.4 findEssentialMat(LMEDS, 0.999, 1.0) + recoverPose on 19 200
// stride-4 correspondences, on the CPU.  OpenCV is not part of the reference tree; the same
// LMedS principle is implemented here with a normalised 8-point minimal solver, 256 hypotheses,
// median squared Sampson distance over a <=2048-point scoring subset, cheirality vote, then
// cam.t = R*t (geometry.cpp:330).  (Deviation D5 in DESIGN.md; SURVEY.md §8(f)-1 "next" row.)
// Correspondence extraction and the closed-form depth map run on the GPU; the 256-hypothesis
// LMedS itself is a few hundred kFLOP and stays on the host like in the reference.
#include "vk_common.hpp"
#include "vk_device.hpp"
#include "vk_internal.hpp"
#include "vk_p3p.hpp"
#include "../../include/voldor_hip.h"
#include <vector>
#include <algorithm>
#include <cmath>

namespace vk {

__global__ static void k_extract_corr(const float2* __restrict__ flow, float* __restrict__ out, int w, int h, int step, int nx, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (i % nx) * step, y = (i / nx) * step;
    float2 f = flow[y * w + x];
    out[i * 2] = (float)x + f.x;
    out[i * 2 + 1] = (float)y + f.y;
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

// ---- host LMedS ---------------------------------------------------------------------------------
namespace {
// cyclic Jacobi, ascending eigenvalues, columns of V = eigenvectors
void sym_eig(double* A, int n, double* V, double* e) {
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-30) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                const double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; k++) { double a = A[k * n + p], b = A[k * n + q]; A[k * n + p] = c * a - s * b; A[k * n + q] = s * a + c * b; }
                for (int k = 0; k < n; k++) { double a = A[p * n + k], b = A[q * n + k]; A[p * n + k] = c * a - s * b; A[q * n + k] = s * a + c * b; }
                for (int k = 0; k < n; k++) { double a = V[k * n + p], b = V[k * n + q]; V[k * n + p] = c * a - s * b; V[k * n + q] = s * a + c * b; }
            }
    }
    for (int i = 0; i < n; i++) e[i] = A[i * n + i];
    for (int i = 0; i < n; i++) {
        int m = i;
        for (int j = i + 1; j < n; j++) if (e[j] < e[m]) m = j;
        if (m != i) {
            std::swap(e[i], e[m]);
            for (int k = 0; k < n; k++) std::swap(V[k * n + i], V[k * n + m]);
        }
    }
}
double det3x3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
// E = U diag(s) V^T with det U = det V = +1
void svd_3x3(const double* E, double* U, double* s, double* V) {
    double EtE[9], ev[3], Vv[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double a = 0; for (int k = 0; k < 3; k++) a += E[k * 3 + i] * E[k * 3 + j]; EtE[i * 3 + j] = a; }
    sym_eig(EtE, 3, Vv, ev);
    for (int c = 0; c < 3; c++) {
        for (int r = 0; r < 3; r++) V[r * 3 + c] = Vv[r * 3 + (2 - c)];
        s[c] = std::sqrt(ev[2 - c] > 0 ? ev[2 - c] : 0.0);
    }
    if (det3x3(V) < 0) for (int r = 0; r < 3; r++) V[r * 3 + 2] = -V[r * 3 + 2];
    double u[3][3];
    for (int c = 0; c < 2; c++) {
        double n = 0;
        for (int r = 0; r < 3; r++) { u[c][r] = E[r * 3] * V[c] + E[r * 3 + 1] * V[3 + c] + E[r * 3 + 2] * V[6 + c]; n += u[c][r] * u[c][r]; }
        n = std::sqrt(n); if (n < 1e-300) n = 1;
        for (int r = 0; r < 3; r++) u[c][r] /= n;
    }
    {
        double d = u[0][0] * u[1][0] + u[0][1] * u[1][1] + u[0][2] * u[1][2], n = 0;
        for (int r = 0; r < 3; r++) { u[1][r] -= d * u[0][r]; n += u[1][r] * u[1][r]; }
        n = std::sqrt(n); if (n < 1e-300) n = 1;
        for (int r = 0; r < 3; r++) u[1][r] /= n;
    }
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) U[r * 3 + c] = u[c][r];
}
}  // namespace

// p2: [n][2] = pixel + flow on the stride grid (float). Returns 1 on success; R9/t3 float.
int lmeds_essential_host(const float* p2, int nx, int ny, int step, float fxf, float fyf, float cxf, float cyf, float* R9, float* t3) {
    const int n = nx * ny;
    if (n < 8) return 0;
    const double fx = fxf, fy = fyf, cx = cxf, cy = cyf;
    std::vector<double> q1((size_t)n * 2), q2((size_t)n * 2);
    for (int i = 0; i < n; i++) {
        const int x = (i % nx) * step, y = (i / nx) * step;
        q1[i * 2] = (x - cx) / fx; q1[i * 2 + 1] = (y - cy) / fy;
        q2[i * 2] = (p2[i * 2] - cx) / fx; q2[i * 2 + 1] = (p2[i * 2 + 1] - cy) / fy;
    }
    const int HYPS = 256, SCORE_MAX = 2048;
    const int sstride = (n + SCORE_MAX - 1) / SCORE_MAX, ns = (n + sstride - 1) / sstride;
    std::vector<double> errs(ns);
    double best_med = INFINITY, bestE[9] = { 0 };
    for (int hy = 0; hy < HYPS; hy++) {
        double AtA[81] = { 0 };
        for (int k = 0; k < 8; k++) {
            const int i = (int)(rng3(RAND_SEED, (uint32_t)hy, 0x100u + (uint32_t)k) % (uint32_t)n);
            const double a[9] = { q2[i * 2] * q1[i * 2], q2[i * 2] * q1[i * 2 + 1], q2[i * 2],
                                  q2[i * 2 + 1] * q1[i * 2], q2[i * 2 + 1] * q1[i * 2 + 1], q2[i * 2 + 1],
                                  q1[i * 2], q1[i * 2 + 1], 1.0 };
            for (int r = 0; r < 9; r++) for (int c = 0; c < 9; c++) AtA[r * 9 + c] += a[r] * a[c];
        }
        double V[81], ev[9], E0[9], U[9], s[3], Vt[9], E[9];
        sym_eig(AtA, 9, V, ev);
        for (int r = 0; r < 9; r++) E0[r] = V[r * 9];
        svd_3x3(E0, U, s, Vt);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) E[r * 3 + c] = U[r * 3] * Vt[c * 3] + U[r * 3 + 1] * Vt[c * 3 + 1];
        for (int j = 0; j < ns; j++) {
            const int i = j * sstride;
            const double x1 = q1[i * 2], y1 = q1[i * 2 + 1], x2 = q2[i * 2], y2 = q2[i * 2 + 1];
            const double Ex0 = E[0] * x1 + E[1] * y1 + E[2], Ex1 = E[3] * x1 + E[4] * y1 + E[5], Ex2 = E[6] * x1 + E[7] * y1 + E[8];
            const double Et0 = E[0] * x2 + E[3] * y2 + E[6], Et1 = E[1] * x2 + E[4] * y2 + E[7];
            const double num = x2 * Ex0 + y2 * Ex1 + Ex2;
            errs[j] = num * num / (Ex0 * Ex0 + Ex1 * Ex1 + Et0 * Et0 + Et1 * Et1);
        }
        std::nth_element(errs.begin(), errs.begin() + ns / 2, errs.end());  // the ns/2-th order statistic
        const double med = errs[ns / 2];
        if (med < best_med) { best_med = med; memcpy(bestE, E, sizeof E); }
    }
    double U[9], s[3], V[9];
    svd_3x3(bestE, U, s, V);
    const double W[9] = { 0, -1, 0, 1, 0, 0, 0, 0, 1 };
    double Rc[2][9];
    for (int k = 0; k < 2; k++) {
        double UW[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int j = 0; j < 3; j++) a += U[r * 3 + j] * (k == 0 ? W[j * 3 + c] : W[c * 3 + j]); UW[r * 3 + c] = a; }
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int j = 0; j < 3; j++) a += UW[r * 3 + j] * V[c * 3 + j]; Rc[k][r * 3 + c] = a; }
    }
    const double tc[3] = { U[2], U[5], U[8] };
    int best = 0, best_cnt = -1;
    for (int cand = 0; cand < 4; cand++) {
        const double* R = Rc[cand >> 1];
        const double sg = (cand & 1) ? -1.0 : 1.0;
        const double t[3] = { sg * tc[0], sg * tc[1], sg * tc[2] };
        int cnt = 0;
        for (int j = 0; j < ns; j++) {
            const int i = j * sstride;
            const double a[3] = { q1[i * 2], q1[i * 2 + 1], 1 }, b[3] = { q2[i * 2], q2[i * 2 + 1], 1 };
            const double Ra[3] = { R[0] * a[0] + R[1] * a[1] + R[2] * a[2], R[3] * a[0] + R[4] * a[1] + R[5] * a[2], R[6] * a[0] + R[7] * a[1] + R[8] * a[2] };
            const double A11 = Ra[0] * Ra[0] + Ra[1] * Ra[1] + Ra[2] * Ra[2], A12 = -(Ra[0] * b[0] + Ra[1] * b[1] + Ra[2] * b[2]);
            const double A22 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
            const double r1 = -(Ra[0] * t[0] + Ra[1] * t[1] + Ra[2] * t[2]), r2 = b[0] * t[0] + b[1] * t[1] + b[2] * t[2];
            const double det = A11 * A22 - A12 * A12;
            if (std::fabs(det) < 1e-12) continue;
            const double z1 = (r1 * A22 - A12 * r2) / det, z2 = (A11 * r2 - A12 * r1) / det;
            if (z1 > 0 && z2 > 0) cnt++;
        }
        if (cnt > best_cnt) { best_cnt = cnt; best = cand; }
    }
    const double* R = Rc[best >> 1];
    const double sg = (best & 1) ? -1.0 : 1.0;
    float Rf[9], tf[3] = { (float)(sg * tc[0]), (float)(sg * tc[1]), (float)(sg * tc[2]) };
    for (int i = 0; i < 9; i++) Rf[i] = (float)R[i];
    memcpy(R9, Rf, sizeof Rf);
    for (int r = 0; r < 3; r++) t3[r] = Rf[r * 3] * tf[0] + Rf[r * 3 + 1] * tf[1] + Rf[r * 3 + 2] * tf[2];  // cam.t = R*t (:330)
    return 1;
}

// host angle-axis of a rotation matrix (float), same formula as the device one (rodrigues.h:5-79)
void host_rotmat_to_angle_axis(const float* R, float* aa) {
    float a0 = R[7] - R[5], a1 = R[2] - R[6], a2 = R[3] - R[1];
    float costheta = fminf(fmaxf((R[0] + R[4] + R[8] - 1.f) * 0.5f, -1.f), 1.f);
    float sintheta = fminf(sqrtf(a0 * a0 + a1 * a1 + a2 * a2) * 0.5f, 1.f);
    const float theta = atan2f(sintheta, costheta);
    if (sintheta > 1.1920929e-07f) { const float r = theta / (2.f * sintheta); aa[0] = a0 * r; aa[1] = a1 * r; aa[2] = a2 * r; return; }
    if (costheta > 0.f) { aa[0] = a0 * 0.5f; aa[1] = a1 * 0.5f; aa[2] = a2 * 0.5f; return; }
    const float inv = 1.f / (1.f - costheta);
    aa[0] = theta * sqrtf((R[0] - costheta) * inv); aa[1] = theta * sqrtf((R[4] - costheta) * inv); aa[2] = theta * sqrtf((R[8] - costheta) * inv);
}

int closed_form_depth_device(Context* c, const float2* flow0, float* depth, int w, int h, const float* K9, const float* R9,
                             const float* t3, float* scratch12_dev) {
    // KRKinv and b = K t in float (cv::Mat float products, geometry.cpp:269-270)
    const float Kinv[9] = { 1.f / K9[0], 0, -K9[2] / K9[0], 0, 1.f / K9[4], -K9[5] / K9[4], 0, 0, 1 };
    float KR[9], Mb[12];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float s = 0; for (int k = 0; k < 3; k++) s += K9[i * 3 + k] * R9[k * 3 + j]; KR[i * 3 + j] = s; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float s = 0; for (int k = 0; k < 3; k++) s += KR[i * 3 + k] * Kinv[k * 3 + j]; Mb[i * 3 + j] = s; }
    for (int i = 0; i < 3; i++) Mb[9 + i] = K9[i * 3] * t3[0] + K9[i * 3 + 1] * t3[1] + K9[i * 3 + 2] * t3[2];
    VK_CHECK(hipMemcpyAsync(scratch12_dev, Mb, sizeof Mb, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    hipLaunchKernelGGL(k_depth_closed_form, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, c->stream, flow0, depth, w, h,
                       scratch12_dev, 1e-2f, 1e10f);
    VK_CHECK_LAST();
    return 0;
}

int bootstrap_device(Context* c, ImageSet& S, int w, int h, float fx, float fy, float cx, float cy, CamState* cam0_dev) {
    const int step = 4, nx = (w + step - 1) / step, ny = (h + step - 1) / step, n = nx * ny;
    if (int e = c->tmp.reserve(sizeof(float) * (2 * (size_t)n + 16))) return e;
    float* d_p2 = c->tmp.as<float>();
    hipLaunchKernelGGL(k_extract_corr, dim3((n + 255) / 256), dim3(256), 0, c->stream, S.flows.as<float2>(), d_p2, w, h, step, nx, n);
    VK_CHECK_LAST();
    std::vector<float> p2((size_t)n * 2);
    VK_CHECK(hipMemcpyAsync(p2.data(), d_p2, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    float R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, t[3] = { 0, 0, 0 }, rv[3] = { 0, 0, 0 };
    if (lmeds_essential_host(p2.data(), nx, ny, step, fx, fy, cx, cy, R, t)) host_rotmat_to_angle_axis(R, rv);
    VK_CHECK(hipMemcpyAsync(S.pb()->Rs[0], R, sizeof R, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipMemcpyAsync(S.pb()->ts[0], t, sizeof t, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipMemcpyAsync(cam0_dev->rvec, rv, sizeof rv, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipMemcpyAsync(cam0_dev->t, t, sizeof t, hipMemcpyHostToDevice, c->stream));
    const float K9[9] = { fx, 0, cx, 0, fy, cy, 0, 0, 1 };
    return closed_form_depth_device(c, S.flows.as<float2>(), S.depth.as<float>(), w, h, K9, R, t, d_p2 + 2 * (size_t)n);
}

}  // namespace vk

extern "C" {
int vk_estimate_pose_epipolar(const float* h_flow, const float* h_K, int w, int h, float* h_o_R9, float* h_o_t3) {
    const int step = 4, nx = (w + step - 1) / step, ny = (h + step - 1) / step;
    std::vector<float> p2((size_t)nx * ny * 2);
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {
            const int x = i * step, y = j * step;
            p2[(size_t)(j * nx + i) * 2] = (float)x + h_flow[((size_t)y * w + x) * 2];
            p2[(size_t)(j * nx + i) * 2 + 1] = (float)y + h_flow[((size_t)y * w + x) * 2 + 1];
        }
    return vk::lmeds_essential_host(p2.data(), nx, ny, step, h_K[0], h_K[4], h_K[2], h_K[5], h_o_R9, h_o_t3) ? 0 : 1;
}
int vk_estimate_depth_closed_form(const float* h_flow, const float* h_K, const float* h_R9, const float* h_t3, int w, int h,
                                  float* h_o_depth) {
    vk::Context* c = vk::default_context();
    if (!c) return (int)hipErrorNoDevice;
    const size_t npx = (size_t)w * h;
    if (int e = c->tmp.reserve(sizeof(float) * (3 * npx + 16))) return e;
    float* d_flow = c->tmp.as<float>(); float* d_depth = d_flow + 2 * npx; float* d_s = d_depth + npx;
    VK_CHECK(hipMemcpyAsync(d_flow, h_flow, sizeof(float) * 2 * npx, hipMemcpyHostToDevice, c->stream));
    if (int e = vk::closed_form_depth_device(c, reinterpret_cast<const float2*>(d_flow), d_depth, w, h, h_K, h_R9, h_t3, d_s)) return e;
    VK_CHECK(hipMemcpyAsync(h_o_depth, d_depth, sizeof(float) * npx, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}
}
