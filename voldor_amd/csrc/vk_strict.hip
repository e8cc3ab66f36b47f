// voldor_amd/csrc/vk_strict.hip -- "strict math" variants of the stages whose fast kernels re-associate arithmetic.
//
// Strict mode (config key --strict_math 1, or vk_set_strict_math) exists to PIN parity: every stage evaluates the
// reference's expressions in the reference's order (the order the CPU oracle restates, file:line below), transcendentals come
// from vk_strict_math.h on both sides, so a whole window of the HIP pipeline can be compared with the oracle bit for bit
// (tests/test_gpu_strict.py) instead of "within the estimator's sampling noise".  Speed is secondary here (a window takes a few
// times longer than in the fast mode); the fast kernels are then held to the strict ones by measured distances.
//   * fb_smooth: one lane per line, the step-by-step recurrence of gpu-kernels/fb_smooth.h:26-69 (the fast kernels compose
//     projective maps and use v_rcp_f32, deviation D7)
//   * mode finding: mean-shift (meanshift.cu:34-150) and robust Gaussian (fit_robust_gaussian.cu:101-286) with every sum taken
//     in the order of the reference's shared-memory tree (reduce_vector_sum.h:12-61: blocks of 512 rows, thread t starts from
//     x[t] + x[t+256], strides 128..1, block sums form the next level), the 6x6 algebra of aux_funs.cpp:101-141 as a serial
//     fp64 LU on one lane, and the scalings of geometry.cpp:186-263 with cv::Mat's `/=` = multiply by (float)(1./s)
// The per-pixel depth kernels have their strict variants as template instantiations in vk_depth.hip.
#include "vk_common.hpp"
#include "vk_device.hpp"
#include "vk_p3p.hpp"
#include "vk_strict_math.h"
#include "vk_internal.hpp"

namespace vk {

// ---- fb_smooth, gpu-kernels/fb_smooth.h:26-69 -----------------------------------------------------------------------------
// pass 0: rows (line = row, stride 1), pass 1: columns (line = column, stride w).  F = forward messages of the line (scratch,
// same layout as the maps).  The backward recurrence and the posterior are fused: step i of the backward chain needs the raw
// e1[i], which is overwritten by the posterior only after it has been used.
__global__ __launch_bounds__(64) static void k_fb_strict(float* __restrict__ maps, float* __restrict__ fwd, int w, int h, int pass, float e0, float p,
                                                          const int* __restrict__ n_dev) {
#pragma clang fp contract(off)
    if (n_dev && (int)blockIdx.y >= *n_dev) return;
    const int l = blockIdx.x * 64 + threadIdx.x;
    const int n = pass == 0 ? w : h, lines = pass == 0 ? h : w, stride = pass == 0 ? 1 : w;
    if (l >= lines) return;
    const size_t base = (size_t)blockIdx.y * w * h + (pass == 0 ? (size_t)l * w : (size_t)l);
    float* e1 = maps + base;
    float* F = fwd + base;
    float prev = e1[0];
    for (int i = 0; i < n; i++) {  // FB_MSG_L2R / T2B (:27-36, :47-55)
        const float s0 = (prev * (1.f - p) + (1.f - prev) * p) * e0;
        const float s1 = (prev * p + (1.f - prev) * (1.f - p)) * e1[(size_t)i * stride];
        prev = s1 / (s0 + s1);
        F[(size_t)i * stride] = prev;
    }
    prev = e1[(size_t)(n - 1) * stride];
    for (int i = n - 1; i >= 0; i--) {  // FB_MSG_R2L / B2T (:37-46, :56-64), then FB_POSTERIOR (:65-69)
        const float e = e1[(size_t)i * stride];
        const float s0 = prev * e * (1.f - p) + (1.f - prev) * p * e0;
        const float s1 = prev * e * p + (1.f - prev) * (1.f - p) * e0;
        prev = s1 / (s0 + s1);
        const float f = F[(size_t)i * stride];
        const float q0 = (1.f - f) * (1.f - prev), q1 = f * prev;
        e1[(size_t)i * stride] = q1 / (q0 + q1);
    }
}
int fb_smooth_strict_device(Context* c, float* maps, int n_maps, int w, int h, float s0_ems_prob, float no_change_prob, const int* n_dev) {
    if (n_maps <= 0) return 0;
    if (int e = c->fb_scratch.reserve(sizeof(float) * (size_t)n_maps * w * h)) return e;
    hipLaunchKernelGGL(k_fb_strict, dim3((h + 63) / 64, n_maps), dim3(64), 0, c->stream, maps, c->fb_scratch.as<float>(), w, h, 0, s0_ems_prob,
                       no_change_prob, n_dev);
    hipLaunchKernelGGL(k_fb_strict, dim3((w + 63) / 64, n_maps), dim3(64), 0, c->stream, maps, c->fb_scratch.as<float>(), w, h, 1, s0_ems_prob,
                       no_change_prob, n_dev);
    VK_CHECK_LAST();
    return 0;
}

// ---- sums in the reference's tree order ----------------------------------------------------------------------------------
constexpr int ST_THREADS = 256;  // = the reference's reduction block (reduce_vector_sum.h: 256 threads x 2 rows)
constexpr int ST_MAXBLK = 64;    // level-1 blocks: up to 32768 elements (the pipeline draws 8192 hypotheses)
template <int NV> struct TreeBuf { float s[NV][ST_THREADS]; float lvl[NV][ST_MAXBLK]; float out[NV]; };
// out[v] = sum over i < n of elem(i)[v], summed like the reference: per level, block b covers rows [512 b, 512 b + 512), thread t
// starts from x[t] + x[t+256] (rows beyond n contribute nothing), then a binary tree over strides 128..1; the block sums are the
// rows of the next level.  Called by all ST_THREADS threads; results in tb.out after the call (synchronised).
template <int NV, typename ElemFn>
__device__ __forceinline__ void tree_sum(int n, ElemFn elem, TreeBuf<NV>& tb) {
#pragma clang fp contract(off)
    const int t = threadIdx.x;
    const int nb = (n + 2 * ST_THREADS - 1) / (2 * ST_THREADS);
    if (n == 1) {  // the reference's loop `while (n > 1)` does not run: the single row is the result
        if (t == 0) { float v[NV]; elem(0, v); for (int k = 0; k < NV; k++) tb.out[k] = v[k]; }
        __syncthreads();
        return;
    }
    for (int b = 0; b < nb; b++) {
        const int idx = b * 2 * ST_THREADS + t;
        float v0[NV], v1[NV];
        const bool h0 = idx < n, h1 = idx + ST_THREADS < n;
        if (h0) elem(idx, v0);
        if (h1) elem(idx + ST_THREADS, v1);
#pragma unroll
        for (int k = 0; k < NV; k++) tb.s[k][t] = h0 ? (h1 ? v0[k] + v1[k] : v0[k]) : 0.f;
        __syncthreads();
        for (int stride = ST_THREADS / 2; stride >= 1; stride >>= 1) {
            if (t < stride) {
#pragma unroll
                for (int k = 0; k < NV; k++) tb.s[k][t] += tb.s[k][t + stride];
            }
            __syncthreads();
        }
        if (t < NV) tb.lvl[t][b] = tb.s[t][0];
        __syncthreads();
    }
    if (nb == 1) {
        if (t < NV) tb.out[t] = tb.lvl[t][0];
        __syncthreads();
        return;
    }
    // second level: nb (<= 64) rows, one block
#pragma unroll
    for (int k = 0; k < NV; k++) tb.s[k][t] = t < nb ? tb.lvl[k][t] : 0.f;  // t + 256 < nb never holds
    __syncthreads();
    for (int stride = ST_THREADS / 2; stride >= 1; stride >>= 1) {
        if (t < stride) {
#pragma unroll
            for (int k = 0; k < NV; k++) tb.s[k][t] += tb.s[k][t + stride];
        }
        __syncthreads();
    }
    if (t < NV) tb.out[t] = tb.s[t][0];
    __syncthreads();
}

// serial n x n inverse + determinant, double, LU with partial pivoting: aux_funs.cpp:101-118 (cv::determinant / Matx::inv);
// same operation order as the oracle's lu_inverse.  One lane.  Returns det; Ainv valid iff det > 0.
__device__ static double lu_inverse_serial(const double* A, double* Ainv, int n) {
#pragma clang fp contract(off)
    double a[36], b[36];
    for (int i = 0; i < n * n; i++) a[i] = A[i];
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) b[i * n + j] = (i == j) ? 1.0 : 0.0;
    double det = 1;
    for (int i = 0; i < n; i++) {
        int k = i;
        for (int j = i + 1; j < n; j++) if (fabs(a[j * n + i]) > fabs(a[k * n + i])) k = j;
        if (fabs(a[k * n + i]) < 2.220446049250313e-16) return 0;
        if (k != i) {
            for (int j = 0; j < n; j++) {
                double t = a[i * n + j]; a[i * n + j] = a[k * n + j]; a[k * n + j] = t;
                t = b[i * n + j]; b[i * n + j] = b[k * n + j]; b[k * n + j] = t;
            }
            det = -det;
        }
        det *= a[i * n + i];
        const double d = -1 / a[i * n + i];
        for (int j = i + 1; j < n; j++) {
            const double alpha = a[j * n + i] * d;
            for (int c = i + 1; c < n; c++) a[j * n + c] += alpha * a[i * n + c];
            for (int c = 0; c < n; c++) b[j * n + c] += alpha * b[i * n + c];
        }
    }
    if (det > 0) {
        for (int i = n - 1; i >= 0; i--)
            for (int c = 0; c < n; c++) {
                double s = b[i * n + c];
                for (int k = i + 1; k < n; k++) s -= a[i * n + k] * b[k * n + c];
                b[i * n + c] = s / a[i * n + i];
            }
        for (int i = 0; i < n * n; i++) Ainv[i] = b[i];
    }
    return det;
}

// (float)(1. / (double)b): what cv::Mat `/= b` multiplies by (OpenCV core/mat.inl.hpp) -- geometry.cpp:238,249, py_export.cpp:74
__host__ __device__ __forceinline__ float cv_div_scale(float b) { return (float)(1.0 / (double)b); }

// ---- per-camera mode finding in the reference's arithmetic (voldor/geometry.cpp:156-263) -------------------------------------
// ONE workgroup of 256 threads: ordered compaction of the finite hypotheses into `pool` (geometry.cpp:156-165), mean-shift,
// optionally the robust-Gaussian refit, unscaling, the pose into CamState / PoseBlock, the truncation decision.
struct StrictShared {
    TreeBuf<28> tb;
    int wcnt[4];
    int base;
    int flag;
    double full[36], inv[36];
    float cov[21], cinv[21], mean[6];
};
__global__ __launch_bounds__(ST_THREADS) static void k_pose_strict(const float* __restrict__ rvecs, const float* __restrict__ tvecs, int n_poses, ModeParams mp,
                                                                    CamState* cam, PoseBlock* P, int cam_idx, const int* __restrict__ n_points_dev,
                                                                    float* __restrict__ pool) {
#pragma clang fp contract(off)
    __shared__ StrictShared S;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (*n_points_dev < 4) {  // geometry.cpp:84-88
        if (t == 0) { cam->success = 0; maybe_decide(mp, P, cam, cam_idx); }
        return;
    }
    // ---- pool of finite hypotheses, in index order; rvec scaled for the mean-shift metric (:156-165, :191)
    if (t == 0) S.base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n_poses; i0 += ST_THREADS) {
        const int i = i0 + t;
        float v[6] = { 0, 0, 0, 0, 0, 0 };
        bool fin = false;
        if (i < n_poses) {
            // k_solve<.., FROM_MAP> leaves the pool as coordinate planes [3][n_poses] (vk_pose.hip)
            v[0] = rvecs[i]; v[1] = rvecs[(size_t)n_poses + i]; v[2] = rvecs[(size_t)2 * n_poses + i];
            v[3] = tvecs[i]; v[4] = tvecs[(size_t)n_poses + i]; v[5] = tvecs[(size_t)2 * n_poses + i];
            fin = isfinite(v[0] + v[1] + v[2] + v[3] + v[4] + v[5]);
        }
        const unsigned long long m = __ballot(fin);
        if (lane == 0) S.wcnt[wv] = __popcll(m);
        __syncthreads();
        int off = S.base;
        for (int k = 0; k < wv; k++) off += S.wcnt[k];
        if (fin) {
            const int r = off + __popcll(m & ((1ull << lane) - 1ull));
            pool[(size_t)r * 6] = v[0] * mp.rvec_scale; pool[(size_t)r * 6 + 1] = v[1] * mp.rvec_scale; pool[(size_t)r * 6 + 2] = v[2] * mp.rvec_scale;
            pool[(size_t)r * 6 + 3] = v[3]; pool[(size_t)r * 6 + 4] = v[4]; pool[(size_t)r * 6 + 5] = v[5];
        }
        __syncthreads();
        if (t == 0) S.base += S.wcnt[0] + S.wcnt[1] + S.wcnt[2] + S.wcnt[3];
        __syncthreads();
    }
    const int used = S.base;
    if (used == 0) {
        if (t == 0) { cam->success = 0; maybe_decide(mp, P, cam, cam_idx); }
        return;
    }
    __threadfence_block();
    __syncthreads();
    // ---- mean-shift, meanshift.cu:34-150 (host rand() of :76 -> rng3, as everywhere)
    const bool external_init = mp.use_external_init_mean < 0 ? (cam->pose_sample_count != 0) : (mp.use_external_init_mean != 0);
    float io_mean[6], c_mean[6];
    for (int d = 0; d < 3; d++) { io_mean[d] = cam->rvec[d] * mp.rvec_scale; io_mean[3 + d] = cam->t[d]; }
    const float two_var = 2 * mp.kernel_var;
    TreeBuf<28>& tb = S.tb;
    if (external_init) {
        for (int d = 0; d < 6; d++) c_mean[d] = io_mean[d];
    } else {
        float best = 0;
        int best_idx = -1;
        for (int trial = 0; trial < mp.ms_max_init_trials; trial++) {  // :75-95
            const int idx_rand = (int)(rng3(RAND_SEED, (uint32_t)trial, 0x4D53u) % (uint32_t)used);
            float c[6];
            for (int d = 0; d < 6; d++) c[d] = pool[(size_t)idx_rand * 6 + d];
            auto elem = [&](int i, float* v) {
                float l2 = 0;
                for (int d = 0; d < 6; d++) { const float df = pool[(size_t)i * 6 + d] - c[d]; l2 += df * df; }
                v[0] = vsm_expf(-l2 / two_var);
                for (int k = 1; k < 28; k++) v[k] = 0.f;
            };
            tree_sum<28>(used, elem, tb);
            const float wsum = tb.out[0];
            __syncthreads();
            if (wsum > best) { best = wsum; best_idx = idx_rand; }
            if (best > mp.ms_good_init_confidence * (float)used) break;
        }
        if (best_idx < 0) best_idx = 0;
        for (int d = 0; d < 6; d++) c_mean[d] = pool[(size_t)best_idx * 6 + d];
    }
    int ms_iters = 0;
    float conf = 0.f;
    for (int iter = 0; iter < mp.ms_max_iters; iter++) {  // :103-134
        auto elem = [&](int i, float* v) {
            float x[6], l2 = 0;
            for (int d = 0; d < 6; d++) { x[d] = pool[(size_t)i * 6 + d]; const float df = x[d] - c_mean[d]; l2 += df * df; }
            const float wgt = vsm_expf(-l2 / two_var);
            v[0] = wgt;
            for (int d = 0; d < 6; d++) v[1 + d] = x[d] * wgt;
            for (int k = 7; k < 28; k++) v[k] = 0.f;
        };
        tree_sum<28>(used, elem, tb);
        const float wsum = tb.out[0];
        float m[6];
        for (int d = 0; d < 6; d++) m[d] = tb.out[1 + d] / wsum;
        __syncthreads();
        conf = wsum / (float)used;
        ms_iters = iter + 1;
        float disp = 0;
        for (int d = 0; d < 6; d++) disp += (io_mean[d] - m[d]) * (io_mean[d] - m[d]);  // vs. the stale io mean on the 1st pass (SURVEY B-6)
        disp = sqrtf(disp);
        for (int d = 0; d < 6; d++) io_mean[d] = m[d];
        if (disp < mp.ms_epsilon) break;
        for (int d = 0; d < 6; d++) c_mean[d] = io_mean[d];
    }
    float pose_opm[6];
    for (int d = 0; d < 6; d++) pose_opm[d] = io_mean[d];
    float density = conf;
    int gu_iters = cam->last_used_gu_iters;
    float covar_out[36];
    bool write_covar = false;
    if (mp.do_rg) {  // geometry.cpp:201-246, fit_robust_gaussian.cu:101-286
        const float sc = mp.rg_pose_scaling;
        const int N = used;
        if (t < 21) S.cov[t] = 0.f;
        __syncthreads();
        if (t < 6) S.cov[(t * t + t) / 2 + t] = mp.kernel_var * (sc * sc);  // diag(var) *= sc*sc (:203-208)
        if (t < 6) S.mean[t] = pose_opm[t] * sc;
        __syncthreads();
        float weight = 0;
        int iter = 0;
        bool reliable = true;
        gu_iters = 0;  // fit_robust_gaussian.cu:158-159 resets *used_iters on entry
        for (iter = 0; iter < mp.rg_max_iters; iter++) {
            if (t == 0) {  // covar_half_to_full (:17-25), Ledoit-Wolf shrinkage (aux_funs.cpp:124-141), inverse (:101-118)
                for (int d1 = 0; d1 < 6; d1++)
                    for (int d2 = 0; d2 <= d1; d2++) {
                        S.full[d1 * 6 + d2] = (double)S.cov[(d1 * d1 + d1) / 2 + d2];
                        if (d1 != d2) S.full[d2 * 6 + d1] = S.full[d1 * 6 + d2];
                    }
                if (iter > 0 && mp.rg_covar_reg_lambda > 0) {
                    double tr = 0;
                    for (int d = 0; d < 6; d++) tr += S.full[d * 6 + d];
                    const double m = tr / (double)6, lam = (double)mp.rg_covar_reg_lambda;
                    for (int i = 0; i < 6; i++)
                        for (int j = 0; j < 6; j++) S.full[i * 6 + j] = lam * m * (i == j ? 1.0 : 0.0) + (1 - lam) * S.full[i * 6 + j];
                }
                const double det = lu_inverse_serial(S.full, S.inv, 6);
                S.flag = det <= 0 ? 2 : 0;
                if (det > 0)
                    for (int d1 = 0; d1 < 6; d1++)
                        for (int d2 = 0; d2 <= d1; d2++) {
                            S.cov[(d1 * d1 + d1) / 2 + d2] = (float)S.full[d1 * 6 + d2];
                            S.cinv[(d1 * d1 + d1) / 2 + d2] = (float)S.inv[d1 * 6 + d2];
                        }
            }
            __syncthreads();
            if (S.flag == 2) { reliable = false; break; }
            const float prev_density = weight / (float)N;
            float mean[6], cinv[21];
            for (int d = 0; d < 6; d++) mean[d] = S.mean[d];
            for (int k = 0; k < 21; k++) cinv[k] = S.cinv[k];
            auto elem = [&](int i, float* v) {  // e_step (:56-97)
                float x[6], diff[6];
                for (int d = 0; d < 6; d++) { x[d] = pool[(size_t)i * 6 + d] * sc; diff[d] = x[d] - mean[d]; }
                float z = 0;
                for (int d1 = 0; d1 < 6; d1++) {
                    float tmp = 0;
                    for (int d2 = 0; d2 < 6; d2++) {
                        const int hi = d1 >= d2 ? d1 : d2, lo = d1 >= d2 ? d2 : d1;
                        tmp += cinv[(hi * hi + hi) / 2 + lo] * diff[d2];
                    }
                    z += tmp * diff[d1];
                }
                z = sqrtf(z);
                const float wgt = z < mp.rg_trunc_sigma ? 1.f : 0.f;
                v[0] = wgt;
                for (int d = 0; d < 6; d++) v[1 + d] = wgt * x[d];
                for (int d1 = 0; d1 < 6; d1++)
                    for (int d2 = 0; d2 <= d1; d2++) v[7 + (d1 * d1 + d1) / 2 + d2] = wgt * diff[d1] * diff[d2];
            };
            tree_sum<28>(N, elem, tb);
            weight = tb.out[0];
            float nm[6], nc[21];
            for (int d = 0; d < 6; d++) nm[d] = tb.out[1 + d] / weight;
            for (int k = 0; k < 21; k++) nc[k] = tb.out[7 + k] / weight;
            __syncthreads();
            if (!isfinite(weight)) { reliable = false; break; }
            if (fabsf(weight / (float)N - prev_density) < mp.rg_epsilon) { reliable = true; break; }
            if (t < 6) S.mean[t] = nm[t];  // m step (:213-243)
            if (t < 21) S.cov[t] = nc[t];
            __syncthreads();
        }
        __syncthreads();
        if (reliable) {
            density = weight / (float)N; gu_iters = iter;
            const float isc2 = cv_div_scale(sc * sc);  // pose_covar /= sc*sc (cv::Mat, :224)
            for (int i1 = 0; i1 < 6; i1++)
                for (int i2 = 0; i2 < 6; i2++) {
                    const int hi = i1 >= i2 ? i1 : i2, lo = i1 >= i2 ? i2 : i1;
                    float cv = S.cov[(hi * hi + hi) / 2 + lo] * isc2;
                    if (i1 < 3 || i2 < 3) cv /= mp.rvec_scale;  // element-wise at<float>() /= (:226-233)
                    if (i1 < 3 && i2 < 3) cv /= mp.rvec_scale;
                    covar_out[i1 * 6 + i2] = cv;
                }
            for (int d = 0; d < 6; d++) pose_opm[d] = S.mean[d];
        } else {
            for (int k = 0; k < 36; k++) covar_out[k] = 0.f;
            for (int d = 0; d < 6; d++) pose_opm[d] = pose_opm[d] * sc;  // pose_opm *= sc (:210) stays as it went in
        }
        write_covar = true;
        const float isc = cv_div_scale(sc);  // pose_opm /= sc (:238)
        for (int d = 0; d < 6; d++) pose_opm[d] *= isc;
    }
    {
        const float irs = cv_div_scale(mp.rvec_scale);  // :249
        for (int d = 0; d < 3; d++) pose_opm[d] *= irs;
    }
    if (t == 0) {
        bool ok = true;
        for (int d = 0; d < 6; d++) ok = ok && isfinite(pose_opm[d]);  // checkRange :256
        cam->pose_sample_count = used;
        cam->pose_density = density;
        cam->last_used_ms_iters = ms_iters;
        cam->last_used_gu_iters = gu_iters;
        if (write_covar) for (int k = 0; k < 36; k++) cam->covar[k] = covar_out[k];
        cam->success = ok ? 1 : 0;
        if (ok) {
            for (int d = 0; d < 3; d++) { cam->rvec[d] = pose_opm[d]; cam->t[d] = pose_opm[3 + d]; P->ts[cam_idx][d] = pose_opm[3 + d]; }
            float R[9];
            angle_axis_to_rotmat(pose_opm, R, true);
            for (int k = 0; k < 9; k++) P->Rs[cam_idx][k] = R[k];
        }
        maybe_decide(mp, P, cam, cam_idx);
    }
}

// ---- B-inner, strict: meanshift_gpu / fit_robust_gaussian on a host-supplied sample matrix space[N][dims] ------------------------
// Same arithmetic as k_pose_strict (the reference's tree-order sums, serial fp64 LU), generic in the dimension like the reference's
// entry points (meanshift.cu:34-150: dims <= 16; fit_robust_gaussian.cu:101-286: dims <= 6).  io layout as k_meanshift_only /
// k_robust_gaussian_only in vk_pose.hip.
__global__ __launch_bounds__(ST_THREADS) static void k_meanshift_strict(const float* __restrict__ space, int N, ModeParams mp, float* __restrict__ io,
                                                                         int* __restrict__ ioi) {
#pragma clang fp contract(off)
    __shared__ TreeBuf<17> tb;
    const int dims = mp.dims;
    float io_mean[16], c_mean[16];
    for (int d = 0; d < 16; d++) { io_mean[d] = d < dims ? io[d] : 0.f; c_mean[d] = io_mean[d]; }
    const float two_var = 2 * mp.kernel_var;
    if (!mp.use_external_init_mean) {  // meanshift.cu:72-95
        float best = 0;
        int best_idx = -1;
        for (int trial = 0; trial < mp.ms_max_init_trials; trial++) {
            const int idx_rand = (int)(rng3(RAND_SEED, (uint32_t)trial, 0x4D53u) % (uint32_t)N);
            auto elem = [&](int i, float* v) {
                float l2 = 0;
                for (int d = 0; d < dims; d++) { const float df = space[(size_t)i * dims + d] - space[(size_t)idx_rand * dims + d]; l2 += df * df; }
                v[0] = vsm_expf(-l2 / two_var);
                for (int k = 1; k < 17; k++) v[k] = 0.f;
            };
            tree_sum<17>(N, elem, tb);
            const float wsum = tb.out[0];
            __syncthreads();
            if (wsum > best) { best = wsum; best_idx = idx_rand; }
            if (best > mp.ms_good_init_confidence * (float)N) break;
        }
        if (best_idx < 0) best_idx = 0;
        for (int d = 0; d < dims; d++) c_mean[d] = space[(size_t)best_idx * dims + d];
    }
    int iters = 0;
    float conf = 0.f;
    for (int iter = 0; iter < mp.ms_max_iters; iter++) {  // :103-134
        auto elem = [&](int i, float* v) {
            float l2 = 0;
            for (int d = 0; d < dims; d++) { const float df = space[(size_t)i * dims + d] - c_mean[d]; l2 += df * df; }
            const float wgt = vsm_expf(-l2 / two_var);
            v[0] = wgt;
            for (int d = 0; d < 16; d++) v[1 + d] = d < dims ? space[(size_t)i * dims + d] * wgt : 0.f;
        };
        tree_sum<17>(N, elem, tb);
        const float wsum = tb.out[0];
        float m[16];
        for (int d = 0; d < 16; d++) m[d] = d < dims ? tb.out[1 + d] / wsum : 0.f;
        __syncthreads();
        conf = wsum / (float)N;
        iters = iter + 1;
        float disp = 0;
        for (int d = 0; d < dims; d++) disp += (io_mean[d] - m[d]) * (io_mean[d] - m[d]);
        disp = sqrtf(disp);
        for (int d = 0; d < dims; d++) io_mean[d] = m[d];
        if (disp < mp.ms_epsilon) break;
        for (int d = 0; d < dims; d++) c_mean[d] = io_mean[d];
    }
    if (threadIdx.x == 0) {
        for (int d = 0; d < dims; d++) io[d] = io_mean[d];
        io[16] = conf; ioi[0] = iters; ioi[1] = 0;
    }
}
// io: [0..5] mean in/out, [6..41] covar full (dims x dims) in/out, [42] density out ; ioi: [0] iters, [1] 0 reliable / 1 not
__global__ __launch_bounds__(ST_THREADS) static void k_rg_strict(const float* __restrict__ space, int N, ModeParams mp, float* __restrict__ io,
                                                                  int* __restrict__ ioi) {
#pragma clang fp contract(off)
    __shared__ TreeBuf<28> tb;
    __shared__ double s_full[36], s_inv[36];
    __shared__ float s_cov[21], s_cinv[21], s_mean[6];
    __shared__ int s_flag;
    const int dims = mp.dims, t = threadIdx.x, dc = (dims * dims + dims) / 2;
    if (t < dims) s_mean[t] = io[t];
    if (t == 0)
        for (int d1 = 0; d1 < dims; d1++)
            for (int d2 = 0; d2 <= d1; d2++) s_cov[(d1 * d1 + d1) / 2 + d2] = io[6 + d1 * dims + d2];
    __syncthreads();
    float weight = 0;
    int iter = 0;
    bool reliable = true;
    for (iter = 0; iter < mp.rg_max_iters; iter++) {
        if (t == 0) {
            for (int d1 = 0; d1 < dims; d1++)
                for (int d2 = 0; d2 <= d1; d2++) {
                    s_full[d1 * dims + d2] = (double)s_cov[(d1 * d1 + d1) / 2 + d2];
                    if (d1 != d2) s_full[d2 * dims + d1] = s_full[d1 * dims + d2];
                }
            if (iter > 0 && mp.rg_covar_reg_lambda > 0) {
                double tr = 0;
                for (int d = 0; d < dims; d++) tr += s_full[d * dims + d];
                const double m = tr / (double)dims, lam = (double)mp.rg_covar_reg_lambda;
                for (int i = 0; i < dims; i++)
                    for (int j = 0; j < dims; j++) s_full[i * dims + j] = lam * m * (i == j ? 1.0 : 0.0) + (1 - lam) * s_full[i * dims + j];
            }
            const double det = lu_inverse_serial(s_full, s_inv, dims);
            s_flag = det <= 0 ? 2 : 0;
            if (det > 0)
                for (int d1 = 0; d1 < dims; d1++)
                    for (int d2 = 0; d2 <= d1; d2++) {
                        s_cov[(d1 * d1 + d1) / 2 + d2] = (float)s_full[d1 * dims + d2];
                        s_cinv[(d1 * d1 + d1) / 2 + d2] = (float)s_inv[d1 * dims + d2];
                    }
        }
        __syncthreads();
        if (s_flag == 2) { reliable = false; break; }
        const float prev_density = weight / (float)N;
        auto elem = [&](int i, float* v) {  // e_step (:56-97)
            float x[6], diff[6];
            for (int d = 0; d < 6; d++) { x[d] = d < dims ? space[(size_t)i * dims + d] : 0.f; diff[d] = d < dims ? x[d] - s_mean[d] : 0.f; }
            float z = 0;
            for (int d1 = 0; d1 < dims; d1++) {
                float tmp = 0;
                for (int d2 = 0; d2 < dims; d2++) {
                    const int hi = d1 >= d2 ? d1 : d2, lo = d1 >= d2 ? d2 : d1;
                    tmp += s_cinv[(hi * hi + hi) / 2 + lo] * diff[d2];
                }
                z += tmp * diff[d1];
            }
            z = sqrtf(z);
            const float wgt = z < mp.rg_trunc_sigma ? 1.f : 0.f;
            for (int k = 0; k < 28; k++) v[k] = 0.f;
            v[0] = wgt;
            for (int d = 0; d < dims; d++) v[1 + d] = wgt * x[d];
            for (int d1 = 0; d1 < dims; d1++)
                for (int d2 = 0; d2 <= d1; d2++) v[7 + (d1 * d1 + d1) / 2 + d2] = wgt * diff[d1] * diff[d2];
        };
        tree_sum<28>(N, elem, tb);
        weight = tb.out[0];
        float nm[6], nc[21];
        for (int d = 0; d < 6; d++) nm[d] = tb.out[1 + d] / weight;
        for (int k = 0; k < 21; k++) nc[k] = tb.out[7 + k] / weight;
        __syncthreads();
        if (!isfinite(weight)) { reliable = false; break; }
        if (fabsf(weight / (float)N - prev_density) < mp.rg_epsilon) { reliable = true; break; }
        if (t < dims) s_mean[t] = nm[t];
        if (t < dc) s_cov[t] = nc[t];
        __syncthreads();
    }
    __syncthreads();
    if (t == 0) {
        ioi[1] = reliable ? 0 : 1;
        if (reliable) {
            ioi[0] = iter;
            io[42] = weight / (float)N;
            for (int d = 0; d < dims; d++) io[d] = s_mean[d];
            for (int d1 = 0; d1 < dims; d1++)
                for (int d2 = 0; d2 <= d1; d2++) {
                    io[6 + d1 * dims + d2] = s_cov[(d1 * d1 + d1) / 2 + d2];
                    io[6 + d2 * dims + d1] = s_cov[(d1 * d1 + d1) / 2 + d2];
                }
        }
    }
}
int meanshift_strict_device(Context* c, const float* space_dev, int N, const ModeParams& mp, float* io_dev, int* ioi_dev) {
    if (N > 2 * ST_THREADS * ST_MAXBLK) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_meanshift_strict, dim3(1), dim3(ST_THREADS), 0, c->stream, space_dev, N, mp, io_dev, ioi_dev);
    VK_CHECK_LAST();
    return 0;
}
int robust_gaussian_strict_device(Context* c, const float* space_dev, int N, const ModeParams& mp, float* io_dev, int* ioi_dev) {
    if (N > 2 * ST_THREADS * ST_MAXBLK) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_rg_strict, dim3(1), dim3(ST_THREADS), 0, c->stream, space_dev, N, mp, io_dev, ioi_dev);
    VK_CHECK_LAST();
    return 0;
}

int pose_mode_strict_device(Context* c, int n_poses, const ModeParams& mp, CamState* cam_dev, PoseBlock* P, int cam_idx) {
    if (n_poses > 2 * ST_THREADS * ST_MAXBLK) {
        fprintf(stderr, "voldor_hip: strict mode supports up to %d pose hypotheses\n", 2 * ST_THREADS * ST_MAXBLK);
        return (int)hipErrorInvalidValue;
    }
    if (int e = c->pool.reserve(sizeof(float) * 6 * (size_t)n_poses)) return e;
    hipLaunchKernelGGL(k_pose_strict, dim3(1), dim3(ST_THREADS), 0, c->stream, c->rvecs.as<float>(), c->tvecs.as<float>(), n_poses, mp, cam_dev, P, cam_idx,
                       c->n_points.as<int>(), c->pool.as<float>());
    VK_CHECK_LAST();
    return 0;
}

}  // namespace vk
