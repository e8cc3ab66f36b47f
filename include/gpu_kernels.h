// include/gpu_kernels.h -- B-inner drop-in boundary of the MI355X VOLDOR kernel library.
//
// Declares, with identical names / C++ linkage / argument order / default arguments, the six
// visual-odometry entry points that the reference declares in
//   /root/reference/gpu-kernels/gpu_kernels.h:11-58
// and that voldor/voldor.cpp:254,274 and voldor/geometry.cpp:36,44,52,150,153,193,221 call.
// A build of the reference host code links against libvoldor_hip.so instead of its
// libgpu-kernels archive with no source change.
//
// Contract (same as the reference):
//  * every pointer is caller-owned HOST memory, dense row-major float32, no pitch;
//    `float* x[]` are caller-allocated tables of per-frame pointers; outputs may alias inputs;
//  * a NULL input means "reuse what is already on the device", a NULL output "do not download";
//  * return value: 0 on success, otherwise the HIP runtime error code as int (message on stderr);
//    fit_robust_gaussian additionally returns non-zero for an unreliable fit;
//  * N <= 16 flows, N_dp <= 16 depth priors; not thread-safe (one caller thread per device).
// The two frame-alignment symbols of the reference header (gpu_kernels.h:60-74, the mapping
// back-end's residual / Jacobian maps) are declared at the end of this file.
#pragma once

#define DLL_EXPORT __attribute__((visibility("default")))

// replaces gpu-kernels/meanshift.cu:34-150 (decl. gpu_kernels.h:11-15)
DLL_EXPORT int meanshift_gpu(float* h_space, float kernel_var,
	float* h_io_mean, float* h_o_confidence, int* used_iters,
	bool use_external_init_mean, int N, int dims,
	float epsilon = 1e-5f, int max_iters = 100,
	int max_init_trials = 20, float good_init_confidence = 0.5f);

// replaces gpu-kernels/fit_robust_gaussian.cu:101-286 (decl. gpu_kernels.h:17-22)
DLL_EXPORT int fit_robust_gaussian(
	float* h_space, float* h_io_mean, float* h_io_covar,
	float trunc_sigma, float covar_reg_lambda,
	float* h_o_density, int* used_iters,
	int N, int dims,
	float epsilon, int max_iters);

// replaces gpu-kernels/collect_p3p_instances.cu:147-250 (decl. gpu_kernels.h:24-35)
DLL_EXPORT int collect_p3p_instances(
	float* h_flows[], float* h_rigidnesses[],
	float* h_depth,
	float* h_K, float* h_Rs[], float* h_ts[],
	float* h_o_p2_map, float* h_o_p3_map,
	int N, int w, int h,
	int active_idx,
	float rigidness_thresh,
	float rigidness_sum_thresh,
	float sample_min_depth,
	float sample_max_depth,
	int max_trace_on_flow);

// replaces gpu-kernels/solve_batch_ap3p.cu:387-437 (decl. gpu_kernels.h:37-39)
DLL_EXPORT int solve_batch_p3p_ap3p_gpu(float* h_p3s, float* h_p2s,
	float* h_o_rvecs, float* h_o_tvecs,
	float* h_K, int N_pts, int N_poses);
// replaces gpu-kernels/solve_batch_lambdatwist.cu:51-102 (decl. gpu_kernels.h:40-42)
DLL_EXPORT int solve_batch_p3p_lambdatwist_gpu(float* h_p3s, float* h_p2s,
	float* h_o_rvecs, float* h_o_tvecs,
	float* h_K, int N_pts, int N_poses);

// replaces gpu-kernels/optimize_depth.cu:293-520 (decl. gpu_kernels.h:44-58)
DLL_EXPORT int optimize_depth_gpu(
	float* h_flows[],
	float* h_rigidnesses[], float* h_o_rigidnesses[],
	float* h_depth_priors[], float* h_depth_prior_pconfs[],
	float* h_depth_prior_confs[], float* h_o_depth_prior_confs[],
	float* h_depth, float* h_o_depth,
	float* h_K, float* h_Rs[], float* h_ts[],
	float* h_dp_Rs[], float* h_dp_ts[],
	float abs_resize_factor,
	int N, int N_dp, int w, int h, float basefocal,
	int n_rand_samples, int global_prop_step, int local_prop_width,
	float lambda, float omega, float disp_delta, float delta,
	bool fb_smooth, float s0_ems_prob, float no_change_prob,
	float range_factor,
	bool update_rigidness_only);

// replaces gpu-kernels/align_frame.cu:443-554 (decl. gpu_kernels.h:60-66): uploads N key-frames (images optional:
// NULL or crw <= 0 = geometry only), derives normals and image gradients on the device
DLL_EXPORT int align_frame_init_gpu(
	float* h_images[],
	float* h_depths[],
	float* h_weights[],
	float* h_K,
	float vbf, float crw,
	int N, int w, int h);

// replaces gpu-kernels/align_frame.cu:414-441 (decl. gpu_kernels.h:68-74): residual [h][w] and Jacobian [h][w*9]
// (rvec, tvec, depth scale, colour scale, colour offset of the reference frame) of frame ref_fid against tar_fid
DLL_EXPORT int align_frame_eval_gpu(
	int ref_fid,
	int tar_fid,
	const float* h_params_ref,
	const float* h_params_tar,
	float* h_o_residual, float* h_o_jacobian,
	const bool apply_weights = true);

