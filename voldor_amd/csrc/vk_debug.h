/* voldor_amd/csrc/vk_debug.h -- verification entry points of libvoldor_hip.so.  NOT part of the product C-ABI (include/voldor_hip.h):
 * nothing a caller of the library needs, only what the test suite uses to hold each launch structure of the pipeline against its plain
 * form (tests/hooks.py binds them).  The switches are one plain struct inside the library (vk_internal.hpp DebugSwitches), written
 * between calls by the one thread that drives the library. */
#ifndef VK_DEBUG_H
#define VK_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif
/* name / values (default first):
 *   "local_serial"    0 | 1   fast mode: local propagation step by step, one lane per chain (optimize_depth.cu:320-396 order), instead of table + planned runs
 *   "cost_rand_plain" 0 | 1   the sample pass evaluates every random depth in full, one after the other (optimize_depth.cu:269-284), instead of exact
 *                             early rejection + survivor queue -- fast AND strict arithmetic
 *   "fb_segment"      0 | 12 | 20 | 40   steps per lane of the segmented fb_smooth of the fast mode (0: by image size)
 *   "global_split"    1 | 0   0: global propagation with one lane per site instead of a group of lanes
 *   "refit_partition" 1 | 0   0: every gate pass of the refit walks the whole pool in arrival order
 *   "split_trials"    1 | 0   0: the mode kernel evaluates the initial-mode trials itself instead of one workgroup per trial
 *   "strict_pose_coop" 1 | 0  strict mode kernel: 16 single-wave workgroups (one per 512-row block of the pool, block sums through global memory, one grid
 *                             barrier per sum) or ONE 512-thread workgroup; the same bits
 *   "strict_plain"    0 | 1   1: strict mode on the plain launch structures (one lane per chain / line, one 256-thread workgroup walking the
 *                             reference's sum tree block by block) instead of the parallel structures -- both give the same bits
 *   "strict_coop_max_polls"  0 | n > 0   polls after which a workgroup of the cooperative strict mode kernel gives up a meeting (0: 2^20); tests set 1 to
 *                             force the hand-over to the single-workgroup kernel
 *   "defer_reduce"    1 | 0   0: every optimize_depth call of a window launches its own density reduction instead of leaving it to the extra workgroups
 *                             of the next correspondence trace
 *   "fb_ride"         1 | 0   0: fb_smooth of a window's depth half always runs as its own launches instead of riding, block by block, in the launches of the
 *                             pose half's mode kernels (FbRide, vk_common.hpp)
 *   "estep_pairs"     1 | 0 | 2   the fast E-step with two pixels per lane on packed fp32 (same bits): from 1.5 M pixels | never | at every size
 *   "fb_side"         1 | 0   0: strict mode: fb_smooth of a window's depth half runs inside the depth half instead of on a second stream next to the pose half
 *   "bootstrap_default" 0 | 5 | 8   what the default of --bootstrap_points stands for: five-point in the fast mode and 8-point in strict mode | always that one.  The
 *                             oracle and the reference goldens start from the 8-point pose: tests that hold a fast window against them set 8
 *   "strict_filter"   1 | 0 | 2   2: as 1, and the pass counts what the filter saw and kept (vk_debug_counter "sf_*").  0: the strict sample pass evaluates every random depth in
 *                             strict arithmetic (exact progressive rejection, k_cost_rand_q_strict) instead of first discarding, on hardware fp32 transcendentals with a
 *                             margin, the ones that cannot win (k_cost_rand_f_strict, vk_depth_impl.hpp); the same bits
 *   "strict_table_filter" 1 | 0 | 2   the table pass of the strict local propagation behind the same filter, the entries it cannot discard queued and evaluated in strict arithmetic by a
 *                             second launch (k_local_table_filter / k_local_table_exact): from 1 M pixels | never | at every size; the same bits
 * Returns the previous value, -1 for an unknown name / value. */
int vk_debug_switch(const char* name, int value);
/* Counters, read and cleared: "strict_coop_fallbacks" = cameras (default context) whose cooperative strict mode kernel gave up a meeting and were
 * computed by the single-workgroup kernel launched behind it; "fb_blocks_rode" = 256-thread fb_smooth blocks the window pipeline (default context) put
 * into mode-kernel launches instead of launches of their own (FbRide); "reduces_rode" = density reductions it attached to a correspondence trace
 * (OdParams::defer_reduce); "fb_side_passes" = fb_smooth passes it ran on the side stream next to a pose half -- counted on the host where the launch is built;
 * "sf_samples" / "sf_sample_survivors" = random depths the strict sample pass put through its fp32 filter / the ones left for strict arithmetic, "sf_table_tiles" /
 * "sf_table_queued" = 64x4 tiles of the strict table pass that went through the filter / entries it queued for strict arithmetic (counted on the device while "strict_filter" is 2; saturating).  -1: unknown name / device error. */
int vk_debug_counter(const char* name);
/* How the 256-thread blocks of a riding fb_smooth (FbRide, vk_common.hpp) would be dealt over the mode kernels of a window with this geometry -- host
 * arithmetic only, no device.  out: [stacks that ride: 0 | 1 (the rigidness maps) | 2 (and the prior confidences), 100 x steps per lane of the rigidness maps' row pass + those of
 * their column pass, row slots R, column slots C (a stack's blocks padded to an even number), launches that carry rows, then per camera: kind
 * (0 nothing, 1 rows, 2 columns), first block, blocks].  Returns the ints written (5 + 3 n_flows), -1 when out is too short. */
int vk_debug_fb_ride_plan(int w, int h, int n_flows, int n_dp, int* out, int n_out);
/* The mode kernel of the window pipeline (k_pose_mode: packed-pair mean shift; with do_rg the robust-Gaussian refit on the same registers --
 * geometry.cpp:156-263, meanshift.cu:34-150, fit_robust_gaussian.cu:131-263) on a caller-supplied pool of pose hypotheses, so that the kernels
 * of the timed path can be held against the oracle stage by stage.  h_rvecs / h_tvecs: [n_poses][3], n_poses <= 8192, a non-finite
 * hypothesis is dropped; io_pose6: rvec and t of the starting pose in (used when use_external_init_mean), the estimate out; o_covar36: the
 * camera record's covariance (zeros unless the refit ran and was reliable).  With the process-wide strict mode on (vk_set_strict_math) the
 * strict-math mode kernel runs instead.  Nonzero on a device error. */
int vk_pose_mode_pool(const float* h_rvecs, const float* h_tvecs, int n_poses, int use_external_init_mean, float* io_pose6,
                      float kernel_var, float rvec_scale, float ms_epsilon, int ms_max_iters, int ms_max_init_trials, float ms_good_init_confidence,
                      int do_rg, float rg_trunc_sigma, float rg_covar_reg_lambda, float rg_epsilon, int rg_max_iters, float rg_pose_scaling,
                      float* o_covar36, float* o_density, int* o_sample_count, int* o_ms_iters, int* o_gu_iters, int* o_success);
#ifdef __cplusplus
}
#endif
#endif
