// voldor_amd/csrc/vk_internal.hpp -- launchers implemented in the .hip translation units.
#pragma once
#include "vk_common.hpp"

namespace vk {

struct ModeParams {
    int dims;
    float kernel_var, ms_epsilon; int ms_max_iters, ms_max_init_trials; float ms_good_init_confidence;
    int use_external_init_mean;  // -1: derive from CamState.pose_sample_count on the device
    float rvec_scale, rg_pose_scaling;
    int do_rg; float rg_trunc_sigma, rg_covar_reg_lambda, rg_epsilon; int rg_max_iters;
    int rg_partition = 1;  // refit: re-deal the pool by distance and stop a pass outside the gate's ball (filled in by pose_mode_device: vk_set_refit_partition)
    // > 0 on the LAST camera of an EM iteration: the kernel that finishes it also takes the truncation decision for the
    // decide_n cameras (voldor.cpp:171-194 -> PoseBlock::n_active) instead of a separate one-thread launch
    int decide_n = 0, decide_allow_trunc = 0; float decide_trunc_rigidness_density = 0.f, decide_trunc_sample_density = 0.f;
    CamBrief* host_brief = nullptr;  // with decide_n: pinned host records the same thread fills for the host's copy of the decision
};

// Verification switches (NOT part of the product C-ABI: declared in vk_debug.h, set through the one entry vk_debug_switch; the test
// suite uses them to hold every launch structure of the fast / strict pipelines against its plain form).  One plain struct, written
// by the test thread between calls, read by the launchers.
struct DebugSwitches {
    int local_serial = 0;      // 1: local propagation walks every chain step by step (k_local_serial) instead of table + runs
    int cost_rand_plain = 0;   // 1: the sample pass evaluates every random depth in full, one after the other
    int fb_segment = 0;        // 0: by size; 12 / 20 / 40: steps per lane of the segmented fb_smooth
    int global_split = 1;      // 0: global propagation with one lane per site
    int refit_partition = 1;   // 0: every gate pass of the refit walks the whole pool in its arrival order
    int split_trials = 1;      // 0: the mode kernel runs the initial-mode trials itself
    int strict_pose_coop = 1;  // strict mode kernel on one single-wave workgroup per 512-row block of the pool (16 compute units) instead of one 512-thread workgroup; same bits
    // round 5
    int strict_coop_max_polls = 0;  // > 0: the cooperative strict mode kernel gives up a meeting after this many polls (tests force the give-up path); 0: 2^22
    int defer_reduce = 1;      // 0: every optimize_depth call of a window launches its own density reduction
    int fb_ride = 1;           // 0: fb_smooth of a window's depth half always runs as its own launches (not in the launches of the pose half's mode kernels)
    int estep_pairs = 1;       // the E-step with two pixels per lane on packed fp32 (same bits): 0 never, 1 from 1.5 M pixels, 2 always
    // round 6
    int fb_side = 1;           // 0: strict mode's fb_smooth runs inside the depth half instead of on the side stream next to the pose half
    int bootstrap_default = 0; // what --bootstrap_points -1 (the default) means: 0 = five-point in the fast mode, 8-point in strict mode; 5 | 8 = that one.  The test suite sets 8 where a
                               // window is held against the oracle or the reference goldens, whose two-view pose is the 8-point one
    int strict_filter = 1;     // 2: as 1, and the pass counts what the filter saw and kept (vk_debug_counter "sf_*").  0: the strict sample pass evaluates every random depth in strict arithmetic (no fp32 pre-filter, vk_depth_impl.hpp); same bits
    int strict_table_filter = 1;  // the table pass of the strict local propagation as filter + queued entries in strict arithmetic (two launches) instead of every entry (one): 1 from 1 M pixels, 2 always, 0 never; same bits
    int strict_plain = 0;      // 1: strict mode on the plain launch structures of rounds 1-3 (one lane per chain / line, one 256-thread workgroup walking the sum tree)
};
DebugSwitches& debug_switches();  // vk_abi.hip

#ifdef __HIPCC__
// voldor.cpp:171-194 on the device: the first camera that failed, was not allowed to run (rigidness density) or is not
// confident enough truncates the window at its index.  The host applies the same rule to its copy of the records.
// `cams` = record of camera 0.  Called by ONE thread, after it has written the last camera's record.
__device__ __forceinline__ void decide_active(PoseBlock* P, const CamState* cams, int n_flows, int allow_trunc, float trunc_rigidness_density,
                                              float trunc_sample_density) {
    int n = n_flows;
    for (int i = 0; i < n_flows; i++) {
        int ok = 0;
        if (!allow_trunc || cams[i].pose_rigidness_density > trunc_rigidness_density) ok = cams[i].success;
        if (!ok || (allow_trunc && cams[i].pose_density < trunc_sample_density)) { n = i; break; }
    }
    P->n_active = n;
}
__device__ __forceinline__ void maybe_decide(const ModeParams& mp, PoseBlock* P, const CamState* cam, int cam_idx) {
    if (mp.decide_n > 0) {
        const CamState* cams = cam - cam_idx;
        decide_active(P, cams, mp.decide_n, mp.decide_allow_trunc, mp.decide_trunc_rigidness_density, mp.decide_trunc_sample_density);
        if (mp.host_brief) {
            for (int i = 0; i < mp.decide_n; i++)
                mp.host_brief[i] = { cams[i].success, cams[i].pose_sample_count, cams[i].last_used_ms_iters, cams[i].last_used_gu_iters,
                                     cams[i].pose_density, cams[i].pose_rigidness_density };
            __threadfence_system();
        }
    }
}
#endif

// vk_depth.hip
int optimize_depth_device(Context* c, ImageSet& S, const OdParams& p);
int cost_map_device(Context* c, ImageSet& S, const OdParams& p);
// dst: the row pass writes there (the column pass then works on it in place); NULL: in place.  st: the stream to launch on; NULL: the context's
int fb_smooth_device(Context* c, float* maps, int n_maps, int w, int h, float s0_ems_prob, float no_change_prob, const int* n_dev = nullptr,
                     PoseBlock* cumP = nullptr, int cumN = 0, int cumNdp = 0, float* world_scale = nullptr, float* dst = nullptr, hipStream_t st = nullptr);
// vk_strict.hip
int fb_smooth_strict_device(Context* c, float* maps, int n_maps, int w, int h, float s0_ems_prob, float no_change_prob, const int* n_dev = nullptr, float* dst = nullptr, hipStream_t st = nullptr);
int pose_mode_strict_device(Context* c, int n_poses, const ModeParams& mp, CamState* cam_dev, PoseBlock* P, int cam_idx);
int strict_filter_stat(Context* c, int k);  // counter k of the strict passes' fp32 filter (vk_debug.h "sf_*"; counted with vk_debug_switch "strict_filter" 2), read and cleared, saturating; -1: device error
int strict_coop_fallbacks(Context* c);  // cameras the single-workgroup strict mode kernel took over from the cooperative one (read and cleared); -1: device error
int meanshift_strict_device(Context* c, const float* space_dev, int N, const ModeParams& mp, float* io_dev, int* ioi_dev);
int robust_gaussian_strict_device(Context* c, const float* space_dev, int N, const ModeParams& mp, float* io_dev, int* ioi_dev);
int fill_device(Context* c, float* p, float v, size_t n);
int scale_device(Context* c, float* p, const float* s_dev, size_t n);
int disp_to_depth_device(Context* c, const float* disp, float* out, float bf, size_t n);
int depth_conf_device(Context* c, const float* rig, const float* confs, float* out, int n_flows, int n_dp, size_t npx);
int gblur_device(Context* c, const float* src, float* dst, float* tmp, float* gk_dev, int w, int h, int d, float sigma, int ksize);

// vk_pose.hip
int flush_pending_reduce(Context* c);  // vk_depth.hip
int collect_device(Context* c, const ImageSet& S, int N, int w, int h, int active_idx, float rig_thresh, float rig_sum_thresh,
                   float min_depth, float max_depth, int max_trace, CamState* cam_dev, bool compact, bool block_compact = false, bool ref_tex = false);
int solve_device(Context* c, const float* pts2, const float* pts3, int* n_pts_dev, float fx, float fy, float cx, float cy,
                 int n_poses, int solver, bool strict = false, CamState* cam_dev = nullptr, bool ref_svd = false, bool ref_rng = false);
// draw = 0: rejection over the map (D3b), falling back to the compacted list below DRAW_LIST_DENSITY; 1: always the reference's
// index draw over the compacted list (geometry.cpp:68-88 + solve_batch_lambdatwist.cu:16-19); -1: rejection only (tests)
int solve_from_maps_device(Context* c, int npx, float fx, float fy, float cx, float cy, int n_poses, int solver, CamState* cam_dev,
                           int draw = 0, bool strict = false, bool ref_svd = false, bool ref_rng = false);
// vk_ref_cuda.h on the device (vk_pose.hip): jump matrices, XORWOW state tables
int xorwow_jumps_device(Context* c);                              // c->xw_jumps ready
int xorwow_pixel_states_device(Context* c, int npx, uint32_t epoch);  // c->xw_px_states = states `epoch` draws after curand_init(RAND_SEED, pixel, 0)
int xorwow_pose_states_device(Context* c, int n_poses);           // c->xw_pose_states = states after curand_init(RAND_SEED, idx, 0)
int pose_mode_device(Context* c, int n_poses, const ModeParams& mp, CamState* cam_dev, PoseBlock* P, int cam_idx, bool trials_first = false, const FbRide* ride = nullptr);
void fb_smooth_plan_segments(int w, int h, int n_maps, int* rows_seg, int* cols_seg, bool* segmented);  // vk_depth.hip: steps per lane of the two passes
int meanshift_device(Context* c, const float* space_dev, int N, const ModeParams& mp, float* io_dev, int* ioi_dev);
int robust_gaussian_device(Context* c, const float* space_dev, int N, const ModeParams& mp, float* io_dev, int* ioi_dev);

// vk_bootstrap.hip
int bootstrap_device(Context* c, ImageSet& S, int w, int h, float fx, float fy, float cx, float cy, CamState* cam0_dev, bool strict = false, int points = 8);

// vk_abi.hip: process-wide default of the strict-math mode (vk_set_strict_math / VOLDOR_HIP_STRICT_MATH)
bool strict_math_default();
bool reference_rng_default();  // vk_set_reference_rng / VOLDOR_HIP_REFERENCE_RNG
bool reference_tex_default();  // vk_set_reference_tex / VOLDOR_HIP_REFERENCE_TEX
bool reference_svd_default();  // vk_set_reference_svd / VOLDOR_HIP_REFERENCE_SVD: rodrigues() through the reference's approximate SVD (vk_ref_svd.h)

}  // namespace vk
