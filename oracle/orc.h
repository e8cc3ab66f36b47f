/* oracle/orc.h -- CPU ORACLE for the VOLDOR per-frame VO inner loop.
 *
 * TEST INFRASTRUCTURE ONLY.  A plain-C restatement of the reference algorithm
 * (htkseason/VOLDOR, /root/reference) used as the parity checker for the HIP
 * product path in voldor_amd/.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library.  The product never does.
 *
 * PARITY PINNING: the reference ships no tests / golden vectors for this path
 *  and its CUDA+OpenCV build cannot run here.  Two levels of pinning
 * against the reference's own code stand in for them:
 *  1. the headers that compile on the host (the lambdatwist headers, residual_model.h,
 *     rodrigues.h+svd3_cuda.h) are built in place into oracle/_ref (ref_wrap.cpp;
 *     tests/test_oracle_vs_golden.py, tests/golden/ref_{lambdatwist,residual,rodrigues}.npz);
 *  2. the CUDA kernel files themselves (optimize_depth.cu, fb_smooth.h, collect_p3p_instances.cu,
 *     solve_batch_{lambdatwist,ap3p}.cu, meanshift.cu, fit_robust_gaussian.cu), host entry points
 *     included, are compiled for the CPU on a small launch-emulation layer (ref_prep.pl,
 *     ref_stubs/emul/, ref_wrap_kernels.cpp) and run thread by thread with the D1/D2
 *     substitutions below; this oracle reproduces their outputs bit for bit (pixel passes,
 *     fb_smooth, correspondence maps, solver translations) or to float summation order
 *     (mean-shift, robust Gaussian) -- tests/test_oracle_vs_ref_kernels.py,
 *     tests/golden/ref_kernels.npz.
 *  3. the host pipeline (voldor/py_export.cpp, voldor.cpp, geometry.cpp, utils.cpp) is compiled in
 *     place against ref_stubs/minicv (stand-in for the OpenCV calls) and linked to 2.
 *     (ref_wrap_host.cpp): the reference's own py_voldor_wrapper runs end to end.  With the
 *     reference's draw (the default; ORC_REFERENCE_DRAW=0 selects D3b), the reference's rodrigues()
 *     -- restated to the bit in voldor_amd/csrc/vk_ref_svd.h (orc_set_reference_svd) or installed as
 *     the reference's own code via orc_set_rodrigues_hook -- and, for the reference's default
 *     exclusive mode, ORC_EMULATE_B1=1, orc_voldor reproduces every output of the windows bit for
 *     bit -- tests/test_oracle_vs_ref_window.py, tests/golden/ref_window.npz, ref_window_strict.npz.
 * Still unpinned: OpenCV's own numerics (5-point bootstrap replaced by D5; Rodrigues / inv / gemm
 * restated in minicv from documented behaviour).
 *
 * Deliberate, documented deviations from the reference:
 *  D1 random numbers: counter-based hash (orc_rng) instead of cuRAND XORWOW.  orc_set_reference_rng(1) (round 4): XORWOW
 *     streams as the reference seeds them, restated from the published algorithm (voldor_amd/csrc/vk_ref_cuda.h); pinned against an
 *     independent implementation and rocRAND's jump table, NOT against a cuRAND run (the seed-scramble constants are unverifiable here).
 *  D2 bilinear sampling: exact fp32 weights, clamp per layer (no 8-bit weights,
 *     no bleeding between stacked layers) -- gmat.h:175-179.  orc_set_reference_tex(1) (round 4): the texture unit's linear filter
 *     (8 fractional bits, one texture over the stacked layers) from the CUDA programming guide's formula; rounding of the fraction and
 *     the order of the four products are that guide's open ends, NOT pinned against a CUDA run.
 *  D3 random sample index clamped to N_pts-1 (reference can read one past the end,
 *     solve_batch_lambdatwist.cu:16-19).
 *  D3b (optional since round 3: ORC_REFERENCE_DRAW=0; the default is the reference's index draw) inside the
 *     window pipeline the 4 correspondences of a hypothesis are drawn uniformly from the valid set by
 *     rejection over the map instead of by index into the compacted list (same distribution; a 1-pixel
 *     change of the set no longer re-draws all 8192 hypotheses -- but an independent sample of them:
 *     tests/test_gpu_ensemble.py).
 *  D4 one depth buffer shared by the depth and the pose half (the reference keeps a
 *     stale un-normalised copy in optimize_depth.cu).  ORC_EMULATE_B1=1 reproduces the reference's
 *     default exclusive_gpu_context mode (product: --reference_stale_depth 1).
 *  A camera's rotation vector is Camera::rvec() of its float matrix (cv::Rodrigues round trip, utils.h:44-53; round 4,
 *     voldor_amd/csrc/vk_ref_cv.h): what the next mean shift starts from and what the window returns -- not a deviation, a correction.
 *  D5 8-point LMedS two-view bootstrap instead of OpenCV's 5-point findEssentialMat (not in the tree).
 *     The product's five-point option (--bootstrap_points 5) is held against orc_fivept.py: the same constraints solved by another algorithm.
 *  D6 world-scale normalisation skipped when the window is lost (reference: 0/0).
 *  D8 rodrigues(): exact polar factor instead of the reference's approximate fp32 SVD (svd3_cuda.h);
 *     orc_set_reference_svd(1) switches to that SVD, bit for bit (product: --reference_svd 1).
 *  (D7 is product-only: fb_smooth arithmetic, see DESIGN.md.)
 *  Switches for the whole-window comparison with the reference pipeline: ORC_REFERENCE_DRAW (default 1),
 *  orc_set_reference_svd() / orc_set_rodrigues_hook() (D8), ORC_EMULATE_B1=1 (D4, test only).
 *
 * All citations are file:line relative to /root/reference.
 */
#ifndef ORC_H
#define ORC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_FRAMES 16

/* ---- RNG (deviation D1) ---- */
uint32_t orc_rng(uint32_t seed, uint32_t stream, uint32_t counter);
float orc_u01(uint32_t r); /* (0,1] like curand_uniform */

/* ---- residual model: gpu-kernels/residual_model.h:6-68 ---- */
float orc_fun_fmag_c(float fmag);
float orc_fun_fmag_scale(float fmag);
float orc_fisk_dist_pdf(float x, float c, float scale);
float orc_fun_rigidness(float dx1, float dy1, float dx2, float dy2, float lambda, float abs_rf);
float orc_fun_depth_rigidness(float d1, float d2, float basefocal, float omega, float abs_rf);

/* ---- depth half: gpu-kernels/optimize_depth.cu:54-520, fb_smooth.h:17-109 ---- */
typedef struct {
    int N, N_dp, w, h;
    float K[9];
    float Rs[ORC_MAX_FRAMES][9], ts[ORC_MAX_FRAMES][3];
    float dp_Rs[ORC_MAX_FRAMES][9], dp_ts[ORC_MAX_FRAMES][3];
    float abs_resize_factor, basefocal;
    int n_rand_samples, global_prop_step, local_prop_width;
    float lambda, omega, disp_delta, delta;
    int fb_smooth;
    float s0_ems_prob, no_change_prob, range_factor;
    int update_rigidness_only;
} orc_od_params;

/* flows [N][h][w][2]; rig [N][h][w] in/out; priors/pconfs [N_dp][h][w]; confs in/out;
 * depth [h][w] in/out; cost [h][w] out (scratch); rand_epoch in/out (persistent RNG counter). */
void orc_optimize_depth(const orc_od_params* p, const float* flows, float* rig,
                        const float* priors, const float* pconfs, float* confs,
                        float* depth, float* cost, uint32_t* rand_epoch);
void orc_fb_smooth(float* maps, int n_maps, int w, int h, float s0_ems_prob, float no_change_prob);
void orc_compute_cost_map(const orc_od_params* p, const float* flows, const float* rig,
                          const float* priors, const float* pconfs, const float* confs,
                          const float* depth, float* cost);
void orc_update_rigidnesses(const orc_od_params* p, const float* flows, float* rig,
                            const float* priors, float* confs, const float* depth);
/* gpu-kernels/gblur.cu:12-72 */
int orc_gblur(const float* src, float* dst, int w, int h, int d, float sigma, int ksize);

/* ---- pose half ---- */
/* gpu-kernels/collect_p3p_instances.cu:70-145; maps are NaN where invalid */
void orc_collect_p3p(const float* flows, const float* rig, const float* depth, const float* K,
                     const float (*Rs)[9], const float (*ts)[3], float* p2_map, float* p3_map,
                     int N, int w, int h, int active_idx, float rigidness_thresh,
                     float rigidness_sum_thresh, float sample_min_depth, float sample_max_depth,
                     int max_trace_on_flow);
/* voldor/geometry.cpp:68-80 row-major compaction; returns n_points */
int orc_compact_p3p(const float* p2_map, const float* p3_map, int npx, float* pts2, float* pts3);
/* 4 random indices of pose sample `idx` (deviation D1,D3) */
void orc_pose_sample_indices(int idx, int n_pts, int out4[4]);
int orc_pose_sample_pixels(int idx, int npx, const float* p2_map, int out4[4]);
void orc_solve_batch_p3p_maps(const float* p2_map, const float* p3_map, int npx, int n_valid, float* rvecs,
                              float* tvecs, const float* K, int n_poses, int use_ap3p, int use_double);
/* lambdatwist/lambdatwist_p4p.h:5-62; use_double selects _T (0: GPU path float, 1: CPU path) */
int orc_lambdatwist_p4p(const float* y8, const float* x12, float fx, float fy, float cx, float cy,
                        int use_double, float* R9, float* t3);
/* gpu-kernels/solve_batch_ap3p.cu:294-378 (4 points -> best of <=4 AP3P solutions) */
int orc_ap3p_p4p(const float* y8, const float* x12, float fx, float fy, float cx, float cy,
                 float* R9, float* t3);
/* gpu-kernels/rodrigues.h:82-114 (project to SO(3), then angle-axis :5-79) */
void orc_rodrigues(const float* R9, float* rvec3);
void orc_rotmat_to_angle_axis(const float* R9, float* rvec3);
void orc_rvec_to_rotmat(const float* rvec3, float* R9); /* cv::Rodrigues(vec->mat), OpenCV 3.4 */
/* gpu-kernels/solve_batch_lambdatwist.cu:51-102 / solve_batch_ap3p.cu:387-437 */
void orc_last_two_view_translation(float* t3); /* test hook: recoverPose-style t of the last orc_estimate_pose_epipolar call */
void orc_set_rodrigues_hook(void (*fn)(const float* R9, float* rvec3, float* Rproj9)); /* test hook, see orc_pose.c */
/* --reference_svd 1 of the product: orc_rodrigues projects with the reference's approximate SVD (svd3_cuda.h:36-1044 restated to the bit
 * in voldor_amd/csrc/vk_ref_svd.h, shared with the HIP kernels like vk_strict_math.h) instead of the exact polar factor (D8) */
/* reference mode, round 4 (voldor_amd/csrc/vk_ref_cuda.h): cuRAND XORWOW streams as the reference seeds them (D1 off) / CUDA's
 * 8-bit-fraction linear filter over the stacked layers (D2 off) */
void orc_set_reference_rng(int on);
void orc_set_reference_tex(int on);
int orc_get_reference_rng(void);
int orc_get_reference_tex(void);
const uint32_t* orc_xorwow_jumps(void); /* [32][800]: T^(2^67 * 2^k) of the XORWOW transition */
void orc_xorwow_stream(unsigned long long seed, uint32_t sub, int n, uint32_t* raw, float* uni, uint32_t* state6);
float orc_fetch1(const float* stack, int f, int n_layers, int w, int h, float x, float y);
void orc_fetch2(const float* stack, int f, int n_layers, int w, int h, float x, float y, float* ox, float* oy);
void orc_set_reference_svd(int on);
int orc_get_reference_svd(void);
void orc_reference_project_rotation(const float* R9, float* Q9); /* rodrigues.h:82-108 alone */
void orc_solve_batch_p3p(const float* pts3, const float* pts2, float* rvecs, float* tvecs,
                         const float* K, int n_pts, int n_poses, int use_ap3p, int use_double);
/* gpu-kernels/meanshift.cu:34-150 */
void orc_meanshift(const float* space, float kernel_var, float* io_mean, float* o_confidence,
                   int* used_iters, int use_external_init_mean, int N, int dims, float epsilon,
                   int max_iters, int max_init_trials, float good_init_confidence);
/* gpu-kernels/fit_robust_gaussian.cu:101-286 ; returns 0 iff reliable */
int orc_fit_robust_gaussian(const float* space, float* io_mean, float* io_covar, float trunc_sigma,
                            float covar_reg_lambda, float* o_density, int* used_iters, int N,
                            int dims, float epsilon, int max_iters);

/* ---- bootstrap: voldor/geometry.cpp:267-332 ---- */
void orc_estimate_depth_closed_form(const float* flow, float* depth, const float* K,
                                    const float* R9, const float* t3, int w, int h,
                                    float min_depth, float max_depth);
int orc_estimate_pose_epipolar(const float* flow, const float* K, int w, int h, int step,
                               float* R9, float* t3);

/* ---- whole window: voldor/py_export.cpp:5-79 + voldor/voldor.cpp:4-317 ---- */
int orc_voldor(const float* flows, const float* disparity, const float* disparity_pconf,
               const float* depth_priors, const float* depth_prior_poses,
               const float* depth_prior_pconfs, float fx, float fy, float cx, float cy,
               float basefocal, int N, int N_dp, int w, int h, const char* config,
               int* n_registered, float* poses, float* poses_covar, float* depth,
               float* depth_conf);

/* strict math (test switch): transcendental calls go through voldor_amd/csrc/vk_strict_math.h, the header the HIP kernels
 * compile in strict mode -> one rounding sequence on both sides (oracle/orc_math.h) */
void orc_set_strict_math(int on);
int orc_get_strict_math(void);
void orc_set_threads(int n);
int orc_get_max_threads(void);

#ifdef __cplusplus
}
#endif
/* ---- frame alignment maps (gpu-kernels/align_frame.cu), orc_align.c ---- */
typedef struct orc_align orc_align;
void orc_rot_with_rvec(const float* p3, const float* rvec, float* out3, float* J_rvec9, float* J_p39);
orc_align* orc_align_init(const float* images, const float* depths, const float* weights, const float* K9, float vbf, float crw, int N, int w, int h);
void orc_align_eval(const orc_align* A, int ref_fid, int tar_fid, const float* params_ref9, const float* params_tar9, float* residual,
                    float* jacobian, int apply_weights);
void orc_align_free(orc_align* A);

#endif
