// oracle/_ref builder, part 2: the reference's CUDA kernel files -- kernels, device functions AND host entry points --
// compiled for the CPU and executed thread by thread.
//
// oracle/ref_prep.pl rewrites the <<< >>> launch statements of gpu-kernels/{optimize_depth,collect_p3p_instances,
// meanshift,fit_robust_gaussian,solve_batch_ap3p,solve_batch_lambdatwist,align_frame,gblur}.cu and fb_smooth.h into a temp directory
// outside the repo (REF_PREP_DIR, deleted after the build); each file is included below in its own namespace (they all
// define file-static __constant__ symbols with the same names).  ref_stubs/emul/ supplies threadIdx & co, a sequential
// launcher, cudaMalloc/cudaMemcpy on host memory, a host-backed GMat, and stand-ins for the three pieces that cannot be
// run as they are: the RNG (D1), the texture filter (D2), the __syncthreads tree reduction (summation order restated)
// and aux_funs.cpp (OpenCV Matx66d: LU in double, defined at the bottom of this file).
//
// TEST INFRASTRUCTURE ONLY: pins oracle/orc_model.c + oracle/orc_pose.c (tests/golden/ref_kernels.npz).
#ifdef REF_PREP_DIR
#include <vector>
#include "ref_stubs/emul/cuda_emul.h"
#include "gpu-kernels/utils.h"
#include "gpu-kernels/gpu_kernels.h"
#include "gpu-kernels/residual_model.h"
#include "gpu-kernels/rodrigues.h"
#include "lambdatwist/lambdatwist_p4p.h"
#include "ref_stubs/emul/gmat.h"
#include "ref_stubs/emul/reduce_vector_sum.h"
#include "ref_stubs/emul/aux_funs.h"
#include "ref_stubs/emul/gblur.h"
#include "gpu-kernels/vops.h"

#define EMUL_STR2(x) #x
#define EMUL_STR(x) EMUL_STR2(x)
namespace ref_od {
#include EMUL_STR(REF_PREP_DIR/optimize_depth.cu)
}
#undef BLOCK_WIDTH
#undef MAX_FRAMES
#undef RAND_SEED
namespace ref_collect {
#include EMUL_STR(REF_PREP_DIR/collect_p3p_instances.cu)
}
#undef BLOCK_WIDTH
#undef MAX_FRAMES
#undef RAND_SEED
#define RAND_SEED 233  /* utils.h:18, dropped by the #undef above */
#define rand emul_host_rand
namespace ref_ms {
#include EMUL_STR(REF_PREP_DIR/meanshift.cu)
}
#undef rand
#undef MAX_DIMS
#undef N_THREADS
namespace ref_rg {
#include EMUL_STR(REF_PREP_DIR/fit_robust_gaussian.cu)
}
#undef MAX_DIMS
#undef N_THREADS
namespace ref_ap3p {
#include EMUL_STR(REF_PREP_DIR/solve_batch_ap3p.cu)
}
#undef N_THREADS
namespace ref_lt {
#include EMUL_STR(REF_PREP_DIR/solve_batch_lambdatwist.cu)
}
#undef N_THREADS
#undef MAX_FRAMES
#undef BLOCK_WIDTH
namespace ref_align {
#include EMUL_STR(REF_PREP_DIR/align_frame.cu)
}
namespace ref_gb {
using ::GMatf;
#include EMUL_STR(REF_PREP_DIR/gblur.cu)
}

extern "C" {

// fb_smooth_batch_inplace (fb_smooth.h:72-108) on caller memory: maps [N][h][w], smoothed in place
void ref_fb_smooth(float* maps, int N, int w, int h, float s0_ems_prob, float no_change_prob) {
    GMatf m; m.bind(maps, w, h, N);
    ref_od::fb_smooth_batch_inplace(m, s0_ems_prob, no_change_prob, N, w, h);
}

// optimize_depth_gpu (optimize_depth.cu:293-520), the reference's own host function.  flows [N][h][w][2], rig [N][h][w]
// (in/out), priors/pconfs [N_dp][h][w], confs [N_dp][h][w] (in/out), depth [h][w] (in/out), cost [h][w] (out: the
// reference keeps it on the device), K = 3x3 row-major, Rs [N][9], ts [N][3].  rand_epoch: counter value the cuRAND
// states start from (the reference keeps its states across calls; here they are re-created for every call).
int ref_optimize_depth(float* flows, float* rig, float* priors, float* pconfs, float* confs, float* depth, float* cost, float* K, float* Rs,
                       float* ts, float* dpRs, float* dpts, float abs_resize_factor, int N, int N_dp, int w, int h, float basefocal,
                       int n_rand_samples, int global_prop_step, int local_prop_width, float lambda, float omega, float disp_delta, float delta,
                       int fb_smooth, float s0_ems_prob, float no_change_prob, float range_factor, int update_rigidness_only,
                       unsigned rand_epoch) {
    const size_t px = (size_t)w * h;
    std::vector<float*> f(N), r(N), R(N), t(N), p(N_dp), pc(N_dp), c(N_dp), dR(N_dp), dt(N_dp);
    for (int i = 0; i < N; i++) { f[i] = flows + i * px * 2; r[i] = rig + i * px; R[i] = Rs + i * 9; t[i] = ts + i * 3; }
    for (int i = 0; i < N_dp; i++) { p[i] = priors + i * px; pc[i] = pconfs + i * px; c[i] = confs + i * px; dR[i] = dpRs + i * 9; dt[i] = dpts + i * 3; }
    ref_od::d_rand_states.create(1, 1, 1);  // force optimize_depth.cu:357-362 to re-create and re-seed the states
    emul_curand_epoch = rand_epoch;
    const int rc = ref_od::optimize_depth_gpu(f.data(), r.data(), r.data(), N_dp ? p.data() : nullptr, N_dp ? pc.data() : nullptr,
                                              N_dp ? c.data() : nullptr, N_dp ? c.data() : nullptr, depth, depth, K, R.data(), t.data(),
                                              N_dp ? dR.data() : nullptr, N_dp ? dt.data() : nullptr, abs_resize_factor, N, N_dp, w, h, basefocal,
                                              n_rand_samples, global_prop_step, local_prop_width, lambda, omega, disp_delta, delta, fb_smooth != 0,
                                              s0_ems_prob, no_change_prob, range_factor, update_rigidness_only != 0);
    emul_curand_epoch = 0;
    if (cost) ref_od::d_cost_map.copy_to_host(cost, make_cudaPos(0, 0, 0), w, h, 1);
    return rc;
}

// collect_p3p_instances (collect_p3p_instances.cu:147-250): p2_map [h][w][2], p3_map [h][w][3], NaN where rejected
int ref_collect_p3p(float* flows, float* rig, float* depth, float* K, float* Rs, float* ts, float* o_p2_map, float* o_p3_map, int N, int w, int h,
                    int active_idx, float rigidness_thresh, float rigidness_sum_thresh, float sample_min_depth, float sample_max_depth,
                    int max_trace_on_flow) {
    const size_t px = (size_t)w * h;
    std::vector<float*> f(N), r(N), R(N), t(N);
    for (int i = 0; i < N; i++) { f[i] = flows + i * px * 2; r[i] = rig + i * px; R[i] = Rs + i * 9; t[i] = ts + i * 3; }
    return ref_collect::collect_p3p_instances(f.data(), r.data(), depth, K, R.data(), t.data(), o_p2_map, o_p3_map, N, w, h, active_idx,
                                              rigidness_thresh, rigidness_sum_thresh, sample_min_depth, sample_max_depth, max_trace_on_flow);
}

// solve_batch_p3p_{lambdatwist,ap3p}_gpu (solve_batch_lambdatwist.cu:51-102, solve_batch_ap3p.cu:387-437)
int ref_solve_batch_p3p(float* p3s, float* p2s, float* o_rvecs, float* o_tvecs, float* K, int N_pts, int N_poses, int use_ap3p) {
    return use_ap3p ? ref_ap3p::solve_batch_p3p_ap3p_gpu(p3s, p2s, o_rvecs, o_tvecs, K, N_pts, N_poses)
                    : ref_lt::solve_batch_p3p_lambdatwist_gpu(p3s, p2s, o_rvecs, o_tvecs, K, N_pts, N_poses);
}

// meanshift_gpu (meanshift.cu:34-150)
int ref_meanshift(float* space, float kernel_var, float* io_mean, float* o_confidence, int* used_iters, int use_external_init_mean, int N, int dims,
                  float epsilon, int max_iters, int max_init_trials, float good_init_confidence) {
    emul_rand_trial = 0; emul_rand_mod = (uint32_t)N;
    return ref_ms::meanshift_gpu(space, kernel_var, io_mean, o_confidence, used_iters, use_external_init_mean != 0, N, dims, epsilon, max_iters,
                                 max_init_trials, good_init_confidence);
}

// fit_robust_gaussian (fit_robust_gaussian.cu:101-286)
int ref_fit_robust_gaussian(float* space, float* io_mean, float* io_covar, float trunc_sigma, float covar_reg_lambda, float* o_density,
                            int* used_iters, int N, int dims, float epsilon, int max_iters) {
    return ref_rg::fit_robust_gaussian(space, io_mean, io_covar, trunc_sigma, covar_reg_lambda, o_density, used_iters, N, dims, epsilon, max_iters);
}

// align_frame_init_gpu / align_frame_eval_gpu (align_frame.cu:414-554): images/depths/weights [N][h][w], K 3x3 row-major
int ref_align_init(float* images, float* depths, float* weights, float* K, float vbf, float crw, int N, int w, int h) {
    const size_t px = (size_t)w * h;
    std::vector<float*> im(N), d(N), wt(N);
    for (int i = 0; i < N; i++) { im[i] = images ? images + i * px : nullptr; d[i] = depths + i * px; wt[i] = weights + i * px; }
    return ref_align::align_frame_init_gpu(images ? im.data() : nullptr, d.data(), wt.data(), K, vbf, crw, N, w, h);
}
int ref_align_eval(int ref_fid, int tar_fid, const float* params_ref, const float* params_tar, float* o_residual, float* o_jacobian,
                   int apply_weights) {
    return ref_align::align_frame_eval_gpu(ref_fid, tar_fid, params_ref, params_tar, o_residual, o_jacobian, apply_weights != 0);
}

// gblur_gpu (gblur.cu:47-72): src / dst [d][h][w]; returns the reference's error code (26 = kernel too wide)
int ref_gblur(float* src, float* dst, int w, int h, int d, float sigma, int ksize) {
    GMatf s, o;
    s.bind(src, w, h, d);
    const int rc = ref_gb::gblur_gpu(s, o, sigma, ksize);
    if (rc == 0) { o.copy_to_host(dst, make_cudaPos(0, 0, 0), w, h, d); o.free(); }
    return rc;
}

// libm / strict / ulp-jitter switch of ref_stubs/emul/cuda_emul.h (and of minicv's Rodrigues)
int ref_math_mode = 0;
void ref_set_math_mode(int mode) { ref_math_mode = mode; }
unsigned int ref_rand_salt = 0;
void ref_set_rand_salt(unsigned int salt) { ref_rand_salt = salt; }
// round 4: cuRAND XORWOW streams / CUDA linear texture filtering instead of the stand-ins D1 / D2 (cuda_emul.h, vk_ref_cuda.h)
int ref_reference_rng = 0, ref_reference_tex = 0;
void ref_set_reference_rng(int on) { ref_reference_rng = on ? 1 : 0; }
void ref_set_reference_tex(int on) { ref_reference_tex = on ? 1 : 0; }
unsigned int ref_jitter_salt = 0;
void ref_set_jitter_salt(unsigned int salt) { ref_jitter_salt = salt; }
// a new window: the next optimize_depth_gpu call re-creates its cuRAND states, starting from counter value `rand_epoch`
void ref_reset_window_state(unsigned rand_epoch) {
    ref_od::d_rand_states.create(1, 1, 1);
    emul_curand_epoch = rand_epoch;
}
}  // extern "C"

// ---- the entry points of gpu_kernels.h at global scope, for the reference's host pipeline (voldor/*.cpp compiled in place,
// ref_wrap_host.cpp): plain forwards into the namespaces above
int meanshift_gpu(float* h_space, float kernel_var, float* h_io_mean, float* h_o_confidence, int* used_iters, bool use_external_init_mean, int N,
                  int dims, float epsilon, int max_iters, int max_init_trials, float good_init_confidence) {
    emul_rand_trial = 0; emul_rand_mod = (uint32_t)N;
    {  // test aid: REF_DUMP_POOL="<call index>,<path>" writes the pool and the start of that mean-shift call (same layout as ORC_DUMP_POOL, oracle/orc_voldor.c)
        static int call = 0;
        const char* dp = getenv("REF_DUMP_POOL");
        int want; char path[512];
        if (dp && sscanf(dp, "%d,%511s", &want, path) == 2 && want == call && dims == 6) {
            FILE* f = fopen(path, "wb");
            if (f) { fwrite(&N, sizeof N, 1, f); fwrite(h_space, sizeof(float), (size_t)N * 6, f); fwrite(h_io_mean, sizeof(float), 6, f); fclose(f); }
        }
        call++;
    }
    return ref_ms::meanshift_gpu(h_space, kernel_var, h_io_mean, h_o_confidence, used_iters, use_external_init_mean, N, dims, epsilon, max_iters,
                                 max_init_trials, good_init_confidence);
}
int fit_robust_gaussian(float* h_space, float* h_io_mean, float* h_io_covar, float trunc_sigma, float covar_reg_lambda, float* h_o_density,
                        int* used_iters, int N, int dims, float epsilon, int max_iters) {
    return ref_rg::fit_robust_gaussian(h_space, h_io_mean, h_io_covar, trunc_sigma, covar_reg_lambda, h_o_density, used_iters, N, dims, epsilon, max_iters);
}
int collect_p3p_instances(float* h_flows[], float* h_rigidnesses[], float* h_depth, float* h_K, float* h_Rs[], float* h_ts[], float* h_o_p2_map,
                          float* h_o_p3_map, int N, int w, int h, int active_idx, float rigidness_thresh, float rigidness_sum_thresh,
                          float sample_min_depth, float sample_max_depth, int max_trace_on_flow) {
    return ref_collect::collect_p3p_instances(h_flows, h_rigidnesses, h_depth, h_K, h_Rs, h_ts, h_o_p2_map, h_o_p3_map, N, w, h, active_idx,
                                              rigidness_thresh, rigidness_sum_thresh, sample_min_depth, sample_max_depth, max_trace_on_flow);
}
int solve_batch_p3p_ap3p_gpu(float* h_p3s, float* h_p2s, float* h_o_rvecs, float* h_o_tvecs, float* h_K, int N_pts, int N_poses) {
    return ref_ap3p::solve_batch_p3p_ap3p_gpu(h_p3s, h_p2s, h_o_rvecs, h_o_tvecs, h_K, N_pts, N_poses);
}
int solve_batch_p3p_lambdatwist_gpu(float* h_p3s, float* h_p2s, float* h_o_rvecs, float* h_o_tvecs, float* h_K, int N_pts, int N_poses) {
    return ref_lt::solve_batch_p3p_lambdatwist_gpu(h_p3s, h_p2s, h_o_rvecs, h_o_tvecs, h_K, N_pts, N_poses);
}
int optimize_depth_gpu(float* h_flows[], float* h_rigidnesses[], float* h_o_rigidnesses[], float* h_depth_priors[], float* h_depth_prior_pconfs[],
                       float* h_depth_prior_confs[], float* h_o_depth_prior_confs[], float* h_depth, float* h_o_depth, float* h_K, float* h_Rs[],
                       float* h_ts[], float* h_dp_Rs[], float* h_dp_ts[], float abs_resize_factor, int N, int N_dp, int w, int h, float basefocal,
                       int n_rand_samples, int global_prop_step, int local_prop_width, float lambda, float omega, float disp_delta, float delta,
                       bool fb_smooth, float s0_ems_prob, float no_change_prob, float range_factor, bool update_rigidness_only) {
    return ref_od::optimize_depth_gpu(h_flows, h_rigidnesses, h_o_rigidnesses, h_depth_priors, h_depth_prior_pconfs, h_depth_prior_confs,
                                      h_o_depth_prior_confs, h_depth, h_o_depth, h_K, h_Rs, h_ts, h_dp_Rs, h_dp_ts, abs_resize_factor, N, N_dp, w, h,
                                      basefocal, n_rand_samples, global_prop_step, local_prop_width, lambda, omega, disp_delta, delta, fb_smooth,
                                      s0_ems_prob, no_change_prob, range_factor, update_rigidness_only);
}

// ---- stand-in for gpu-kernels/aux_funs.cpp:97-141 (cv::Matx66d): partial-pivot Gauss-Jordan in double, N <= 6
static double emul_lu(const double* A, double* Ainv, int n) {  // returns det(A); Ainv (may be null) only valid when det != 0
    double a[36], b[36];
    for (int i = 0; i < n * n; i++) { a[i] = A[i]; b[i] = (i / n == i % n) ? 1.0 : 0.0; }
    double det = 1.0;
    for (int i = 0; i < n; i++) {
        int k = i;
        for (int j = i + 1; j < n; j++) if (fabs(a[j * n + i]) > fabs(a[k * n + i])) k = j;
        if (fabs(a[k * n + i]) < DBL_EPSILON) return 0.0;
        if (k != i) { for (int j = 0; j < n; j++) { std::swap(a[i * n + j], a[k * n + j]); std::swap(b[i * n + j], b[k * n + j]); } det = -det; }
        const double d = -1.0 / a[i * n + i];
        for (int j = i + 1; j < n; j++) {
            const double alpha = a[j * n + i] * d;
            for (int c = i + 1; c < n; c++) a[j * n + c] += alpha * a[i * n + c];
            for (int c = 0; c < n; c++) b[j * n + c] += alpha * b[i * n + c];
        }
        det *= a[i * n + i];
    }
    if (Ainv) {
        for (int i = n - 1; i >= 0; i--) for (int c = 0; c < n; c++) {
            double s = b[i * n + c];
            for (int k = i + 1; k < n; k++) s -= a[i * n + k] * b[k * n + c];
            b[i * n + c] = s / a[i * n + i];
        }
        for (int i = 0; i < n * n; i++) Ainv[i] = b[i];
    }
    return det;
}
double determinant(double* mat, int N) { return emul_lu(mat, nullptr, N); }
double inverse(double* mat, double* mat_inv, int N) {
    double inv[36];
    const double det = emul_lu(mat, inv, N);
    if (det > 0) for (int i = 0; i < N * N; i++) mat_inv[i] = inv[i];
    return det;
}
double regularize_covar_LW_given_lambda(double* mat, double* mat_ret, double lambda, int dims) {
    double m = 0;
    for (int i = 0; i < dims; i++) m += mat[i * dims + i];
    m /= (double)dims;
    double S[36];
    for (int i = 0; i < dims; i++) for (int j = 0; j < dims; j++) S[i * dims + j] = lambda * m * (i == j ? 1.0 : 0.0) + (1 - lambda) * mat[i * dims + j];
    for (int i = 0; i < dims * dims; i++) mat_ret[i] = S[i];
    return emul_lu(S, nullptr, dims);
}
#endif
