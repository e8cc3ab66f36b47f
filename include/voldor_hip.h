/* include/voldor_hip.h -- C-ABI (extern "C") of libvoldor_hip.so: plain pointers and sizes only.
 *
 * Section A re-exports the six B-inner entry points of include/gpu_kernels.h
 * (= /root/reference/gpu-kernels/gpu_kernels.h:11-58) with `int` in place of `bool`, for FFI
 * callers that cannot bind mangled C++ names (ctypes / cgo / JNI).
 * Section B is the B-outer window call, = /root/reference/voldor/py_export.h:3-11 with
 * `int* n_registered` in place of `int&`; vk_voldor_device() is the same call for inputs that are
 * already resident in HBM (any pointer may be a device pointer; direction is inferred).
 * Section C: inspection / control helpers used by tests, bench.py and multi-GPU launchers.
 * Section D: the multi-GPU exchange (RCCL) below the C-ABI.
 * All functions return 0 on success, else a HIP error code (message on stderr) unless noted.
 */
#ifndef VOLDOR_HIP_H
#define VOLDOR_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* ---- A. B-inner (host pointers, NULL = keep device copy / skip download) ---- */
int vk_meanshift_gpu(float* h_space, float kernel_var, float* h_io_mean, float* h_o_confidence,
                     int* used_iters, int use_external_init_mean, int N, int dims, float epsilon,
                     int max_iters, int max_init_trials, float good_init_confidence);
int vk_fit_robust_gaussian(float* h_space, float* h_io_mean, float* h_io_covar, float trunc_sigma,
                           float covar_reg_lambda, float* h_o_density, int* used_iters, int N,
                           int dims, float epsilon, int max_iters);
int vk_collect_p3p_instances(float** h_flows, float** h_rigidnesses, float* h_depth, float* h_K,
                             float** h_Rs, float** h_ts, float* h_o_p2_map, float* h_o_p3_map,
                             int N, int w, int h, int active_idx, float rigidness_thresh,
                             float rigidness_sum_thresh, float sample_min_depth,
                             float sample_max_depth, int max_trace_on_flow);
int vk_solve_batch_p3p_ap3p_gpu(float* h_p3s, float* h_p2s, float* h_o_rvecs, float* h_o_tvecs,
                                float* h_K, int N_pts, int N_poses);
int vk_solve_batch_p3p_lambdatwist_gpu(float* h_p3s, float* h_p2s, float* h_o_rvecs,
                                       float* h_o_tvecs, float* h_K, int N_pts, int N_poses);
/* same sampling, solver scalar = double (the reference's CPU instantiation, geometry.cpp:112) */
int vk_solve_batch_p3p_lambdatwist_f64_gpu(float* h_p3s, float* h_p2s, float* h_o_rvecs,
                                           float* h_o_tvecs, float* h_K, int N_pts, int N_poses);
int vk_optimize_depth_gpu(float** h_flows, float** h_rigidnesses, float** h_o_rigidnesses,
                          float** h_depth_priors, float** h_depth_prior_pconfs,
                          float** h_depth_prior_confs, float** h_o_depth_prior_confs,
                          float* h_depth, float* h_o_depth, float* h_K, float** h_Rs, float** h_ts,
                          float** h_dp_Rs, float** h_dp_ts, float abs_resize_factor, int N, int N_dp,
                          int w, int h, float basefocal, int n_rand_samples, int global_prop_step,
                          int local_prop_width, float lambda, float omega, float disp_delta,
                          float delta, int fb_smooth, float s0_ems_prob, float no_change_prob,
                          float range_factor, int update_rigidness_only);
/* gblur_gpu (gpu-kernels/gblur.cu:47-72) on host arrays [d][h][w]; ksize 0 = max(ceil(6 sigma),3) */
int vk_gblur(const float* h_src, float* h_dst, int w, int h, int d, float sigma, int ksize);
/* fb_smooth (gpu-kernels/fb_smooth.h:72-108) alone: rows then columns, in place on host maps [n_maps][h][w] */
int vk_fb_smooth(float* h_maps, int n_maps, int w, int h, float s0_ems_prob, float no_change_prob);

/* ---- B. B-outer ---- */
int vk_py_voldor_wrapper(const float* flows, const float* disparity, const float* disparity_pconf,
                         const float* depth_priors, const float* depth_prior_poses,
                         const float* depth_prior_pconfs, float fx, float fy, float cx, float cy,
                         float basefocal, int N, int N_dp, int w, int h, const char* config,
                         int* n_registered, float* poses, float* poses_covar, float* depth,
                         float* depth_conf);
/* identical semantics; image-sized arguments (flows, disparity*, depth_priors, depth_prior_pconfs,
 * depth, depth_conf) may be DEVICE pointers (e.g. torch tensors); depth_prior_poses, poses,
 * poses_covar, n_registered are host memory. */
int vk_voldor_device(const float* flows, const float* disparity, const float* disparity_pconf,
                     const float* depth_priors, const float* depth_prior_poses,
                     const float* depth_prior_pconfs, float fx, float fy, float cx, float cy,
                     float basefocal, int N, int N_dp, int w, int h, const char* config,
                     int* n_registered, float* poses, float* poses_covar, float* depth,
                     float* depth_conf);
/* vk_voldor_device plus the result as one DEVICE record pose_block_dev[1 + 6N + 36N] = { n_registered, poses[N][6],
 * poses_covar[N][36] } (unregistered slots zero): the send buffer of the multi-GPU pose exchange (one ncclAllGather per batch
 * step), packed on the device.  The host outputs may be NULL. */
int vk_voldor_device_block(const float* flows, const float* disparity, const float* disparity_pconf,
                           const float* depth_priors, const float* depth_prior_poses,
                           const float* depth_prior_pconfs, float fx, float fy, float cx, float cy,
                           float basefocal, int N, int N_dp, int w, int h, const char* config,
                           int* n_registered, float* poses, float* poses_covar, float* depth,
                           float* depth_conf, float* pose_block_dev);
/* n_windows independent windows of the same geometry and config IN FLIGHT TOGETHER on the current device (no reference
 * counterpart: voldor/py_export.cpp processes one window per call).  Argument b of every pointer array belongs to window b
 * (arrays may be NULL where the single call takes NULL); images may be host or device pointers; n_registered[n_windows],
 * poses[n_windows][N][6], poses_covar[n_windows][N][36] are host arrays.  Each window runs on its own stream and buffers
 * and gives the result of the one-at-a-time call.  At most VOLDOR_HIP_INFLIGHT (environment, default 4) windows are in
 * flight at a time; a longer batch queues behind them (a worker that finishes a window takes the next one). */
int vk_voldor_device_batch(int n_windows, const float* const* flows, const float* const* disparity,
                           const float* const* disparity_pconf, const float* const* depth_priors,
                           const float* const* depth_prior_poses, const float* const* depth_prior_pconfs,
                           float fx, float fy, float cx, float cy, float basefocal, int N, int N_dp, int w, int h,
                           const char* config, int* n_registered, float* poses, float* poses_covar,
                           float* const* depth, float* const* depth_conf);
/* align_frame_init_gpu / align_frame_eval_gpu (gpu-kernels/gpu_kernels.h:60-74; the mapping back-end's residual and
 * Jacobian maps), `bool` -> `int`. */
int vk_align_frame_init_gpu(float** h_images, float** h_depths, float** h_weights, float* h_K, float vbf, float crw,
                            int N, int w, int h);
int vk_align_frame_eval_gpu(int ref_fid, int tar_fid, const float* h_params_ref, const float* h_params_tar,
                            float* h_o_residual, float* h_o_jacobian, int apply_weights);
/* eval_covisibility (slam_py/slam_utils.py:18-53; the step after every VO call, voldor_slam.py:496-504) on the
 * device: depth[h][w] and the optional 0/1 byte mask[h][w] (depth_conf > thresh) may be host or device pointers, T44 is
 * the row-major 4x4 Tc1c2, K9 the row-major intrinsics.  o_counts (optional) receives {visible samples, occupied cells}. */
int vk_eval_covisibility(const float* depth, const unsigned char* mask, const float* T44, const float* K9,
                         int w, int h, int stride, float* o_score, int* o_counts);
/* Middlebury .flo files (voldor/utils.cpp:23-41 load_flow; slam_py/flow_utils.py:10-34): out == NULL only reports
 * the size.  0 ok, 1 cannot open, 2 bad magic / arguments, 3 buffer too small, 4 truncated. */
int vk_read_flo(const char* path, int* w, int* h, float* out, size_t cap_floats);
int vk_write_flo(const char* path, const float* flow, int w, int h);
/* per-camera statistics of the last window (voldor/utils.h:41-45): arrays of length >= N */
int vk_last_camera_stats(int* pose_sample_count, float* pose_density, float* pose_rigidness_density,
                         int* ms_iters, int* gu_iters, int n);
/* bootstrap pieces (voldor/geometry.cpp:267-332) exposed for parity tests; host pointers */
int vk_estimate_pose_epipolar(const float* h_flow, const float* h_K, int w, int h, float* h_o_R9, float* h_o_t3);
/* the GPU bootstrap of the window pipeline on one host flow [h][w][2] (pose + closed-form depth) */
int vk_bootstrap_gpu(const float* h_flow, const float* h_K, int w, int h, float* h_o_R9, float* h_o_t3, float* h_o_depth);
/* the same procedure with the FIVE-point minimal solver (Nister 2004; the solver behind cv::findEssentialMat, which geometry.cpp:316-326 calls):
 * host path, GPU path (points = 8 | 5; config key --bootstrap_points), and the solver alone (host code: five normalised correspondences
 * q1[5][2], q2[5][2] -> up to ten essential matrices Es[10][9], returns their number) */
int vk_estimate_pose_epipolar5(const float* h_flow, const float* h_K, int w, int h, float* h_o_R9, float* h_o_t3);
int vk_bootstrap_gpu_points(const float* h_flow, const float* h_K, int w, int h, int points, float* h_o_R9, float* h_o_t3, float* h_o_depth);
int vk_fivept_solve(const double* q1, const double* q2, double* Es);
int vk_estimate_depth_closed_form(const float* h_flow, const float* h_K, const float* h_R9, const float* h_t3,
                                  int w, int h, float* h_o_depth);

/* ---- C. helpers ---- */
int vk_get_compacted_points(float* h_o_pts2, float* h_o_pts3, int max_points); /* returns n_points (<0 error) */
int vk_set_rand_epoch(unsigned epoch);   /* depth-sampling RNG counter of the current context */
unsigned vk_get_rand_epoch(void);
/* Strict-math mode (process-wide default; the window call also takes the config key --strict_math 0|1): every stage runs in
 * the reference's operation order on software transcendentals (voldor_amd/csrc/vk_strict_math.h), a few times slower, so that
 * results can be compared bit for bit with the CPU oracle in the same mode (parity pinning).  Unset, the
 * default comes from the environment variable VOLDOR_HIP_STRICT_MATH. */
int vk_set_strict_math(int on);
int vk_get_strict_math(void);
/* rodrigues() of the P3P batch (gpu-kernels/rodrigues.h:82-114) through the reference's approximate fp32 SVD
 * (gpu-kernels/svd3_cuda.h:36-1044, restated to the bit in voldor_amd/csrc/vk_ref_svd.h) instead of the exact polar factor.
 * Process-wide default (also VOLDOR_HIP_REFERENCE_SVD=1); the window call takes the config key --reference_svd 0|1.  With
 * --strict_math 1 --reference_draw 1 a window then equals the reference pipeline's strict-math window in every output bit. */
int vk_set_reference_svd(int on);
int vk_get_reference_svd(void);
/* Reference mode, the last two stand-ins (process-wide defaults, also VOLDOR_HIP_REFERENCE_RNG / _TEX = 1; config keys --reference_rng /
 * --reference_tex 0|1; effective with strict math only): the depth samples and the hypothesis draws from cuRAND's XORWOW streams as the
 * reference seeds and advances them (gpu-kernels/optimize_depth.cu:273,286-291; solve_batch_lambdatwist.cu:16-19,44-48) instead of the
 * counter generator, and every at_tex of the reference (gpu-kernels/gmat.h:49-62,175-179) through CUDA's linear texture filter -- 8-bit
 * fractions, one texture over the stacked layers -- instead of the exact per-layer bilinear; both restated from their published
 * definitions in voldor_amd/csrc/vk_ref_cuda.h. */
int vk_set_reference_rng(int on);
int vk_get_reference_rng(void);
int vk_set_reference_tex(int on);
int vk_get_reference_tex(void);
int vk_profile_enable(int on);           /* HIP-event timing of kernel groups on the library's stream */
int vk_profile_get(const char* name, double* total_ms, long* count);
int vk_device_count(void);
int vk_set_device(int dev);
const char* vk_version(void);

/* ---- D. multi-GPU (voldor_amd/csrc/vk_dist.hip).  The reference has no multi-GPU path (file-static device
 * buffers, gpu-kernels/optimize_depth.cu:45-52): nothing to cite but the partition section 8e prescribes -- one process per GPU,
 * independent sequences sharded across the ranks, no data-path collective, ONE ncclAllGather of the pose records
 * { n_registered | poses[N][6] | poses_covar[N][36] } (1 + 42 N floats per rank) per batch step over RCCL / xGMI.
 * The communicator, its stream and the device records are owned by the library (C++ host code); the launcher only carries
 * the 128-byte ncclUniqueId of rank 0 to the other ranks (or names a file all ranks can see).  RCCL (librccl.so.1, or the path in
 * VOLDOR_HIP_RCCL) is bound at the first call of this section.  Return codes: 0, a HIP error code, or 1000 + ncclResult_t.
 * Like the rest of the library (and the reference: file-static state, one caller thread) this section is driven by ONE thread per process;
 * vk_dist_init_file needs a path that is new for every job. */
#define VK_DIST_ID_BYTES 128
int vk_dist_get_unique_id(void* id_out);                      /* rank 0: ncclGetUniqueId */
int vk_dist_init(int rank, int world, const void* id);        /* ncclCommInitRank on the current device (vk_set_device first) */
int vk_dist_init_file(int rank, int world, const char* path, int timeout_s); /* rendezvous through a file written by rank 0 */
int vk_dist_rank(void);                                       /* -1 before vk_dist_init */
int vk_dist_world(void);                                      /* 0 before vk_dist_init */
int vk_dist_rccl_version(void);
int vk_dist_allgather(const float* send_dev, float* recv_dev, int count); /* ncclAllGather of `count` floats per rank, device buffers; returns when done */
int vk_dist_allreduce_max(double* io_host);                   /* max over the ranks (step timing); acts as a barrier */
int vk_dist_barrier(void);
/* Latency of the all-gathers issued so far (vk_voldor_sharded, vk_dist_allgather), in call order, microseconds: dev_us = HIP events around
 * ncclAllGather on the communicator's stream, host_us = the host's wall clock from issue to completion.  Either array may be NULL; at most
 * `cap` samples (the library keeps the first VK_DIST_STATS_MAX since the last reset); *n = samples written; reset != 0 clears the record. */
#define VK_DIST_STATS_MAX 65536
int vk_dist_allgather_stats(float* dev_us, float* host_us, int cap, int* n, int reset);
/* One batch step of the sharded job: this rank's window through vk_voldor_device_block (arguments as vk_voldor_device; flows ==
 * NULL: no sequence for this rank in this step), then the all-gather.  all_blocks_host[world][1 + 42 N] (host) receives every
 * rank's record; n_registered = -1 marks an empty slot.  A rank whose own window fails STILL takes part in the all-gather (the others
 * would wait for it forever): its record is { VK_DIST_FAILED, its error code, 0 ... }, its own call returns that error code, and the
 * other ranks return 0 and find the failure in the gathered records. */
#define VK_DIST_FAILED (-2)
int vk_voldor_sharded(const float* flows, const float* disparity, const float* disparity_pconf,
                      const float* depth_priors, const float* depth_prior_poses,
                      const float* depth_prior_pconfs, float fx, float fy, float cx, float cy,
                      float basefocal, int N, int N_dp, int w, int h, const char* config,
                      int* n_registered, float* poses, float* poses_covar, float* depth,
                      float* depth_conf, float* all_blocks_host);
int vk_dist_finalize(void);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
