// voldor_amd/csrc/vk_voldor.hip -- the drop-in boundary "B-outer": py_voldor_wrapper
// (voldor/py_export.h:3-11, py_export.cpp:5-79) = one visual-odometry window.
//
// Host orchestration of the EM schedule of voldor/voldor.cpp:4-317 (init / solve / bootstrap /
// optimize_cameras / optimize_depth / normalize_world_scale) and the config grammar of
// voldor/config.h:4-253, re-designed so that every image and every pose stays resident in HBM
// for the whole window: the reference downloads depth + N rigidness maps after every
// optimize_depth call, downloads 6 MB of correspondence maps per camera, compacts on the CPU
// and re-uploads.  Here the host only reads one CamState record per camera.
#include "vk_common.hpp"
#include "vk_internal.hpp"
#include "vk_p3p.hpp"
#include "../../include/py_export.h"
#include "../../include/voldor_hip.h"
#include <string>
#include <sstream>
#include <iostream>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace vk {

// ---- Config: voldor/config.h:4-82 (defaults), :110-253 (grammar) ------------------------------
struct Config {
    float omega = 0.15f, disp_delta = 1.f, delta = 0.5f, basefocal = 0;
    int rg_refine = 1, rg_refine_last_only = 1; float rg_trunc_sigma = 3.f, rg_covar_reg_lambda = 0.001f, rg_pose_scaling = 100.f;
    int rg_max_iters = 100; float rg_epsilon = 1e-5f;
    float resize_factor = 1.0f, abs_resize_factor = 1.0f, fx = 0, fy = 0, cx = 0, cy = 0; int exclusive_gpu_context = 1;
    bool debug = false, silent = false, save_everything = false; int viz_img_per_row = 2; float viz_depth_scale = 5;
    float lambda = 0.15f, meanshift_kernel_var = 0.1f, meanshift_rvec_scale = 25.0f; int norm_world_scale = 1;
    int cpu_p3p = 0, lambdatwist = 1, n_poses_to_sample = 8192; float pose_sample_min_depth = 0.1f, pose_sample_max_depth = 1000.0f;
    int max_trace_on_flow = 3; float rigidness_threshold = 0.5f, rigidness_sum_threshold = 1.f;
    float trunc_rigidness_density = 0.05f, trunc_sample_density = 0.001f, no_trunc_iters = 2; int max_iters = 5, min_iters_after_trunc = 3;
    int fb_smooth = 1; float fb_emm = 0.5f, fb_no_change_prob = 0.9f;
    int optimize_depth = 1, depth_rand_samples = 10, depth_global_prop_step = 8, depth_local_prop_width = 32; float depth_range_factor = 1.f;
    int meanshift_max_iters = 100, meanshift_max_init_trials = 20; float meanshift_good_init_confidence = 0.5f, meanshift_epsilon = 1e-5f;
    int kitti_estimate_ground = 0, kitti_ground_holo_width = 5; float kitti_ground_roi = 0.4f, kitti_ground_meanshift_kernel_var = 0.01f;
    // Extensions (not in config.h; they select how results are COMPUTED, not what is computed; parity pinning):
    //  strict_math     1: every stage in the reference's operation order on software transcendentals (vk_strict.hip) -- the window then
    //                  reproduces the CPU oracle in strict mode bit for bit; -1 (default) = the process-wide setting of vk_set_strict_math
    //  reference_draw  1 (default): pose hypotheses index the row-major list of valid correspondences like geometry.cpp:68-88 +
    //                  solve_batch_lambdatwist.cu:16-19 -- rank select over k_collect's validity bitmask, the list is never built;
    //                  0: rejection draw over the map (D3b: same distribution, but an independent sample of the hypotheses) with that
    //                  draw as the low-density fallback.  Default 1 since the ensemble test (tests/test_gpu_ensemble.py): with the
    //                  reference's draw the fast path is statistically indistinguishable from the reference under 1-ulp jitter
    //                  in every metric; with D3b its depth maps sit 1.4x further out
    //  reference_svd   1: rodrigues() of every pose hypothesis through the reference's approximate fp32 SVD (svd3_cuda.h restated to the bit
    //                  in vk_ref_svd.h) instead of the exact polar factor (D8); with strict_math and reference_draw a window then equals the
    //                  REFERENCE pipeline's strict window bit for bit; -1 (default) = the process-wide setting of vk_set_reference_svd
    //  reference_rng   1 (with strict_math): the depth samples and the hypothesis draws come from cuRAND's XORWOW streams as the reference
    //                  seeds and advances them (curand_init(233, pixel | idx, 0); vk_ref_cuda.h) instead of the counter generator (D1)
    //  reference_tex   1 (with strict_math): every at_tex of the reference through CUDA's linear filter -- 8-bit fractions, one texture over
    //                  the stacked layers -- instead of the exact per-layer bilinear (D2).  -1 (default) = the process-wide settings
    //  reference_stale_depth  1 (with strict_math, monocular windows, exclusive_gpu_context 1 = the reference's default): reproduce SURVEY
    //                  Appendix B-1 -- optimize_depth.cu keeps its own device copy of the depth map, which the reference refreshes for the first
    //                  call only and which never sees normalize_world_scale() (voldor.cpp:250-291, :309-317) -- instead of the one
    //                  normalised map (D4).  0 (default) = what the reference does with --exclusive_gpu_context 0
    int strict_math = -1, reference_draw = 1, reference_svd = -1, reference_rng = -1, reference_tex = -1, reference_stale_depth = 0;
    //  bootstrap_points  the monocular two-view bootstrap (voldor/geometry.cpp:316-330: cv::findEssentialMat(pts1, pts2, K, LMEDS, 0.999, 1.0) + recoverPose).
    //                  5: the reference's estimator -- the five-point minimal solver inside LMedS over the 134 subsets OpenCV draws at that confidence
    //                  (vk_fivept.hpp, vk_bootstrap.hip; OpenCV's own numerics are not in the tree: deviation D5 = "not OpenCV's rounding");
    //                  8: the normalised 8-point LMedS of rounds 1-5 (256 subsets);
    //                  -1 (default): 5 in the fast mode (round 6: a sample over a workgroup made it +0.09 ms per window instead of +2.0), 8 in strict mode --
    //                  the reference-mode goldens (tests/golden/ref_window*.npz) were generated with the oracle's 8-point pose injected where the
    //                  reference calls OpenCV, and a strict window reproduces them bit for bit only from that pose
    int bootstrap_points = -1;

    // Returns 0, or non-zero where the reference prints and calls exit(1) (config.h:101-108,245-248):
    // a library must not exit its host process, so the error is reported to the caller instead.
    int read_config(const std::string& s) {
        struct Key { const char* name; int kind; void* p; };  // kind 0 float, 1 int
#define KF(n) { "--" #n, 0, &this->n }
#define KI(n) { "--" #n, 1, &this->n }
        const Key keys[] = {
            KF(basefocal), KF(omega), KF(disp_delta), KF(delta), KI(rg_refine), KI(rg_refine_last_only), KF(rg_trunc_sigma),
            KF(rg_covar_reg_lambda), KF(rg_epsilon), KI(rg_max_iters), KF(rg_pose_scaling), KF(resize_factor), KF(abs_resize_factor),
            KF(fx), KF(fy), KF(cx), KF(cy), KI(viz_img_per_row), KF(viz_depth_scale), KI(exclusive_gpu_context), KF(lambda),
            KF(meanshift_kernel_var), KF(meanshift_rvec_scale), KI(norm_world_scale), KI(cpu_p3p), KI(lambdatwist), KI(max_trace_on_flow),
            KI(n_poses_to_sample), KF(pose_sample_min_depth), KF(pose_sample_max_depth), KF(rigidness_threshold),
            KF(rigidness_sum_threshold), KF(trunc_rigidness_density), KF(trunc_sample_density), KI(max_iters), KF(no_trunc_iters),
            KI(min_iters_after_trunc), KI(fb_smooth), KF(fb_emm), KF(fb_no_change_prob), KI(optimize_depth), KI(depth_rand_samples),
            KI(depth_global_prop_step), KI(depth_local_prop_width), KF(depth_range_factor), KI(meanshift_max_iters),
            KI(meanshift_max_init_trials), KF(meanshift_good_init_confidence), KF(meanshift_epsilon), KI(kitti_estimate_ground),
            KI(kitti_ground_holo_width), KF(kitti_ground_roi), KF(kitti_ground_meanshift_kernel_var),
            KI(strict_math), KI(reference_draw), KI(reference_svd), KI(reference_rng), KI(reference_tex), KI(reference_stale_depth), KI(bootstrap_points),
        };
#undef KF
#undef KI
        std::istringstream iss(s);
        std::string tok;
        while (iss >> tok) {
            if (tok == "--debug") { debug = true; continue; }
            if (tok == "--silent") { silent = true; continue; }
            if (tok == "--save_everything") { save_everything = true; continue; }
            const Key* k = nullptr;
            for (const Key& c : keys) if (tok == c.name) { k = &c; break; }
            if (!k) { std::cout << "Invalid input config : " << tok << std::endl; return 1; }
            std::string val;
            if (!(iss >> val)) { std::cout << "Config array index out of bound." << std::endl; return 2; }
            // str_to_arg falls through i->l->f->d (config.h:85-99): the stored value is stod(str)
            // converted to the field type.
            double v;
            try { v = std::stod(val); } catch (...) { std::cout << "Invalid value for " << tok << " : " << val << std::endl; return 3; }
            if (k->kind == 1) *static_cast<int*>(k->p) = (int)v;
            else *static_cast<float*>(k->p) = (float)v;
        }
        return 0;
    }
};

enum OdFlag { OD_DEFAULT = 0, OD_ONLY_USE_DEPTH_PRIOR = 1, OD_UPDATE_RIGIDNESS_ONLY = 2 };  // voldor.h:7-11

// false in a worker of a batch that keeps several windows in flight (WindowPool): the compute units a mode kernel leaves idle are then busy with the other
// windows' kernels, and fb_smooth as riders (one 512-thread workgroup per compute unit at the mode kernel's 256 registers) costs the batch more than
// its own launches do (four cfg2 windows in flight: 656 -> 627 windows/s with riders)
static thread_local bool g_window_alone = true;

// How the blocks of a riding fb_smooth fall onto the mode kernels of an EM iteration's cameras (FbRide, vk_common.hpp; Voldor::plan_fb_ride has the
// story).  Pure arithmetic on the window's geometry -- held to its invariants on the CPU through vk_debug_fb_ride_plan (tests/test_fb_ride_plan.py):
// every block of the row pass in exactly one launch, then every block of the column pass, no launch with more than 480 blocks.
struct FbRidePlan { bool on = false; int n_stacks = 0, R = 0, C = 0, k_rows = 0, rows_per = 0, cols_per = 0, split_r = 0, split_c = 0; FbStack rows[2], cols[2]; };
static void fb_ride_plan(int w, int h, int n_flows, int n_dp, const float* rig, float* rig2, float* confs, FbRidePlan* out) {
    *out = FbRidePlan();
    if (n_flows < 2) return;
    struct { const float* src; float* dst; int n; } stacks[2] = { { rig, rig2, n_flows }, { confs, confs, n_dp } };
    FbRidePlan q;
    bool can[2] = { false, false };
    for (int j = 0; j < 2; j++) {
        if (stacks[j].n <= 0) continue;
        int rs = 0, cs = 0;
        bool segmented = false;
        fb_smooth_plan_segments(w, h, stacks[j].n, &rs, &cs, &segmented);  // (the segments the pass's own launches use: riding must not change the arithmetic)
        if (!segmented) continue;
        // 40-step segments (from 8 M map pixels: 1080p) do not ride: measured (round 6, profiles/r06n_*) the mode kernels that carry them take 25.6 us
        // instead of 12.7 -- 1.4 ms more per 1080p window for 1.2 ms less fb_smooth in the depth half (28.65 against 27.83 ms).  A 40-step block is ~50 us of
        // strided memory traffic at any occupancy; 12- and 20-step blocks are short dependent chains, which is what hides in a 13 us launch
        if (rs == 40 || cs == 40) continue;
        const int Sr = (w + rs - 1) / rs, Sc = (h + cs - 1) / cs;
        if (Sr > 256 || Sc > 256) continue;
        const int lpb = 256 / Sr, CW = std::min(16, 256 / Sc);
        FbStack& r = q.rows[j];
        r.src = stacks[j].src; r.dst = stacks[j].dst; r.n_maps = stacks[j].n; r.S = Sr; r.seg = rs; r.blocks_x = (h + lpb - 1) / lpb;
        r.vec4 = ((w % 4) == 0 && (reinterpret_cast<uintptr_t>(r.src) % 16) == 0 && (reinterpret_cast<uintptr_t>(r.dst) % 16) == 0) ? 1 : 0;
        r.n_blocks = r.blocks_x * r.n_maps;
        FbStack& k = q.cols[j];
        k.src = stacks[j].dst; k.dst = stacks[j].dst; k.n_maps = stacks[j].n; k.S = Sc; k.seg = cs; k.CW = CW; k.blocks_x = (w + CW - 1) / CW;
        k.n_blocks = k.blocks_x * k.n_maps;
        can[j] = true;
    }
    if (!can[0]) return;  // (the rigidness maps are what is worth moving; the prior confidences alone stay in the depth half)
    // Both stacks if their blocks fit the launches, else the rigidness maps alone (round 6: every stack in the segments its own launches use -- rows and
    // columns may differ, e.g. 12-step rows and 20-step columns on a tall image).
    // The first k_rows mode kernels carry the row slots, the others the column slots: the split with the lightest heaviest launch.
    const int cap = 480;
    for (int ns = (can[1] && stacks[1].n > 0) ? 2 : 1; ns >= 1; ns--) {
        q.split_r = (q.rows[0].n_blocks + 1) & ~1; q.split_c = (q.cols[0].n_blocks + 1) & ~1;
        q.R = q.split_r + (ns == 2 ? q.rows[1].n_blocks : 0); q.C = q.split_c + (ns == 2 ? q.cols[1].n_blocks : 0);
        int best = -1, best_load = 1 << 30;
        for (int k = 1; k < n_flows; k++) {
            const int load = std::max(((q.R + k - 1) / k + 1) & ~1, ((q.C + (n_flows - k) - 1) / (n_flows - k) + 1) & ~1);
            if (load < best_load) { best_load = load; best = k; }
        }
        if (best < 0 || best_load > cap) continue;
        q.k_rows = best; q.rows_per = ((q.R + best - 1) / best + 1) & ~1; q.cols_per = ((q.C + (n_flows - best) - 1) / (n_flows - best) + 1) & ~1;
        q.n_stacks = ns; q.on = true;
        if (ns == 1) { q.rows[1] = FbStack(); q.cols[1] = FbStack(); }
        *out = q;
        return;
    }
}
static FbRide fb_ride_of_camera(const FbRidePlan& P, int i, int n_flows, int w, int h, float e0, float p) {
    FbRide r;
    if (!P.on) return r;
    r.w = w; r.h = h; r.e0 = e0; r.p = p;
    const bool rows = i < P.k_rows;
    const int per = rows ? P.rows_per : P.cols_per, total = rows ? P.R : P.C;
    r.first = (rows ? i : i - P.k_rows) * per;
    r.count = std::max(0, std::min(per, total - r.first));
    r.split = rows ? P.split_r : P.split_c;
    r.kind = r.count > 0 ? (rows ? 1 : 2) : 0;
    for (int j = 0; j < 2; j++) r.st[j] = rows ? P.rows[j] : P.cols[j];
    return r;
}

struct Voldor {
    Context* c = nullptr;
    Config cfg;
    int n_flows = 0, n_flows_init = 0, n_dp = 0, w = 0, h = 0, iters_cur = 0, iters_remain = 0;
    bool has_disparity = false, strict = false, ref_rng = false, ref_tex = false;
    CamState hcams[MAX_FRAMES];

    CamState* dcams() { return c->cams.as<CamState>(); }
    const float* host_flows = nullptr;  // flows still (partly) in host memory: frames [frames_up, n_flows_init) are not on the device yet
    int frames_up = 0;
    // frames 0 .. upto - 1 of host-resident flows are on their way when this returns: copied on the context's own non-blocking copy stream (the
    // kernels already enqueued on the window's stream run meanwhile), and the window's stream waits for "frame f is up" before whatever is
    // enqueued next.  No legacy null stream (a blocking hipMemcpy there synchronises with every blocking stream of the process and
    // relies on the copy being complete at return).  The caller's buffer is read until the events have fired: every exit of voldor_run_on
    // drains the copy stream (WindowGuard).
    bool copies_in_flight = false;
    int upload_frames_up_to(int upto) {
        if (!host_flows) return 0;
        if (int e = c->ensure_copy_stream()) return e;
        const size_t fb = sizeof(float) * 2 * (size_t)w * h;
        for (; frames_up < upto && frames_up < n_flows_init; frames_up++) {
            VK_CHECK(hipMemcpyAsync(c->od.flows.as<char>() + (size_t)frames_up * fb, reinterpret_cast<const char*>(host_flows) + (size_t)frames_up * fb, fb, hipMemcpyHostToDevice, c->copy_stream));
            VK_CHECK(hipEventRecord(c->ev_frame[frames_up], c->copy_stream));
            VK_CHECK(hipStreamWaitEvent(c->stream, c->ev_frame[frames_up], 0));
            copies_in_flight = true;
        }
        if (frames_up >= n_flows_init) host_flows = nullptr;
        return 0;
    }

    // voldor.cpp:4-128
    int init(const float* flows, const float* disparity, const float* disparity_pconf, const float* depth_priors,
             const float* depth_prior_poses, const float* depth_prior_pconfs, int N, int N_dp_in, int w_, int h_) {
        w = w_; h = h_;
        c->pending_reduce = ReduceArgs();  // (a window that failed half way may have left one behind)
        strict = cfg.strict_math < 0 ? strict_math_default() : cfg.strict_math != 0;
        ref_rng = strict && (cfg.reference_rng < 0 ? reference_rng_default() : cfg.reference_rng != 0);  // (the fast kernels keep D1 / D2: their arithmetic is not the reference's anyway)
        ref_tex = strict && (cfg.reference_tex < 0 ? reference_tex_default() : cfg.reference_tex != 0);
        if (ref_rng && !cfg.reference_draw) { std::cout << "--reference_rng 1 needs --reference_draw 1" << std::endl; return (int)hipErrorInvalidValue; }
        if (cfg.bootstrap_points != 5 && cfg.bootstrap_points != 8 && cfg.bootstrap_points != -1) { std::cout << "--bootstrap_points takes 5, 8 or -1 (got " << cfg.bootstrap_points << ")" << std::endl; return (int)hipErrorInvalidValue; }
        if (cfg.reference_stale_depth && !(strict && cfg.exclusive_gpu_context && cfg.norm_world_scale && N_dp_in == 0 && !disparity) && !cfg.silent)
            std::cout << "--reference_stale_depth 1 has no effect here (it needs --strict_math 1, --exclusive_gpu_context 1, --norm_world_scale 1 and a window without depth priors)" << std::endl;
        n_flows = n_flows_init = N;
        iters_cur = 0; iters_remain = cfg.max_iters;
        n_dp = N_dp_in + (disparity ? 1 : 0);
        has_disparity = disparity != nullptr;
        if (N < 1 || N > MAX_FRAMES || n_dp > MAX_DISP_FRAMES || w <= 0 || h <= 0) return (int)hipErrorInvalidValue;
        if (cfg.kitti_estimate_ground && !cfg.silent)  // (parsed and silently ignored)
            std::cout << "--kitti_estimate_ground 1: ground-plane scale estimation (voldor/voldor.cpp:146-147, :320-) is outside this library's scope (SURVEY section 2: OUT OF SCOPE); the window is computed without it" << std::endl;
        if (cfg.resize_factor != 1.f) {
            std::cout << "resize_factor != 1 is deprecated in the reference (config.h:23) and not supported" << std::endl;
            return (int)hipErrorInvalidValue;
        }
        ImageSet& S = c->od;
        hipStream_t st = c->stream;
        const size_t npx = (size_t)w * h;
        S.w = w; S.h = h;
        if (int e = S.ensure_pose()) return e;
        if (int e = S.flows.reserve(sizeof(float) * 2 * npx * N)) return e;
        if (int e = S.rig.reserve(sizeof(float) * npx * N)) return e;
        if (cfg.fb_smooth) { if (int e = S.rig2.reserve(sizeof(float) * npx * N)) return e; }  // (the row pass of an fb_smooth that runs during the pose half -- riding, or on the side stream -- writes here)
        if (int e = S.depth.reserve(sizeof(float) * npx)) return e;
        if (int e = S.cost.reserve(sizeof(float) * npx)) return e;
        if (int e = c->cams.reserve(sizeof(CamState) * MAX_FRAMES)) return e;
        if (int e = c->ms_io.reserve(sizeof(float) * (64 + 8 * 64) + sizeof(int) * 4)) return e;  // at its full size once: world_scale_ptr() stays valid whatever the mode kernels reserve later
        // Flows in HOST memory (py_voldor_wrapper: what the Cython binding passes) go up frame by frame, each right before the first kernel that reads it
        // (upload_frames_up_to): camera i of the first EM iteration traces through the flows of frames <= i only (collect_p3p_instances.cu:100-125) and the
        // bootstrap reads frame 0, so the transfer of frame i + 1 runs while the GPU is busy with camera i instead of in front of everything (round 5:
        // 12 MB at 640x480 x 5, ~0.25 ms of a 4.1 ms host-inclusive window).  Device-resident flows (vk_voldor_device): one device-to-device copy.
        {
            hipPointerAttribute_t at;
            const bool on_device = hipPointerGetAttributes(&at, flows) == hipSuccess && at.type == hipMemoryTypeDevice;
            if (!on_device) (void)hipGetLastError();  // (an unregistered host pointer is reported as an error by some runtimes)
            host_flows = on_device ? nullptr : flows;
            frames_up = on_device ? N : 0;
            if (on_device) VK_CHECK(hipMemcpyAsync(S.flows.p, flows, sizeof(float) * 2 * npx * N, hipMemcpyDeviceToDevice, st));
        }
        if (int e = fill_device(c, S.rig.as<float>(), 1.f, npx * N)) return e;
        PoseBlock& pb = *c->h_pb;  // pinned staging: the previous window of this context ended with a stream synchronize, so it is free
        memset(&pb, 0, sizeof pb);
        pb.n_active = N;
        pb.K4[0] = cfg.fx; pb.K4[1] = cfg.cx; pb.K4[2] = cfg.fy; pb.K4[3] = cfg.cy;
        pb.K4i[0] = 1.f / cfg.fx; pb.K4i[1] = -cfg.cx / cfg.fx; pb.K4i[2] = 1.f / cfg.fy; pb.K4i[3] = -cfg.cy / cfg.fy;
        for (int i = 0; i < MAX_FRAMES; i++) { pb.Rs[i][0] = pb.Rs[i][4] = pb.Rs[i][8] = 1.f; pb.dpRs[i][0] = pb.dpRs[i][4] = pb.dpRs[i][8] = 1.f; }
        if (n_dp > 0) {
            if (int e = S.priors.reserve(sizeof(float) * npx * n_dp)) return e;
            if (int e = S.pconfs.reserve(sizeof(float) * npx * n_dp)) return e;
            if (int e = S.confs.reserve(sizeof(float) * npx * n_dp)) return e;
            int o = 0;
            if (disparity) {  // :31-49
                if (int e = c->tmp.reserve(sizeof(float) * npx)) return e;
                VK_CHECK(hipMemcpyAsync(c->tmp.p, disparity, sizeof(float) * npx, hipMemcpyDefault, st));
                if (int e = disp_to_depth_device(c, c->tmp.as<float>(), S.priors.as<float>(), cfg.basefocal, npx)) return e;
                if (disparity_pconf) VK_CHECK(hipMemcpyAsync(S.pconfs.p, disparity_pconf, sizeof(float) * npx, hipMemcpyDefault, st));
                else if (int e = fill_device(c, S.pconfs.as<float>(), 1.f, npx)) return e;
                o = 1;
            }
            for (int i = 0; i < N_dp_in; i++) {  // :51-67
                VK_CHECK(hipMemcpyAsync(S.priors.as<float>() + (size_t)(o + i) * npx, depth_priors + (size_t)i * npx, sizeof(float) * npx, hipMemcpyDefault, st));
                if (depth_prior_pconfs)
                    VK_CHECK(hipMemcpyAsync(S.pconfs.as<float>() + (size_t)(o + i) * npx, depth_prior_pconfs + (size_t)i * npx, sizeof(float) * npx, hipMemcpyDefault, st));
                else if (int e = fill_device(c, S.pconfs.as<float>() + (size_t)(o + i) * npx, 1.f, npx)) return e;
                angle_axis_to_rotmat(depth_prior_poses + i * 6, pb.dpRs[o + i], strict);
                for (int d = 0; d < 3; d++) pb.dpts[o + i][d] = depth_prior_poses[i * 6 + 3 + d];
            }
            if (int e = fill_device(c, S.confs.as<float>(), 1.f, npx * n_dp)) return e;
        }
        VK_CHECK(hipMemcpyAsync(S.pose.p, &pb, sizeof pb, hipMemcpyHostToDevice, st));
        memset(hcams, 0, sizeof hcams);
        for (int i = 0; i < MAX_FRAMES; i++) hcams[i].pose_rigidness_density = 1.f;  // rigidness maps start at 1 (:96-98)
        memcpy(c->h_cams_up, hcams, sizeof hcams);
        VK_CHECK(hipMemcpyAsync(c->cams.p, c->h_cams_up, sizeof hcams, hipMemcpyHostToDevice, st));  // from pinned staging: no host wait here
        if (n_dp > 0) {  // :106-117
            VK_CHECK(hipMemcpyAsync(S.depth.p, S.priors.p, sizeof(float) * npx, hipMemcpyDeviceToDevice, st));
            if (!disparity) { if (int e = optimize_depth(OD_ONLY_USE_DEPTH_PRIOR)) return e; }
        } else if (int e = fill_device(c, S.depth.as<float>(), 1.f, npx)) return e;
        return 0;
    }

    // voldor.cpp:203-307 (the upload / "minimal cache" branches collapse: everything is resident)
    int optimize_depth(OdFlag flag, bool with_world_scale = false, bool defer_reduce = false) {
        if (n_flows == 0 && n_dp == 0) return 0;
        OdParams p;
        p.abs_resize_factor = cfg.abs_resize_factor;
        p.N = (flag == OD_ONLY_USE_DEPTH_PRIOR) ? 0 : n_flows; p.N_dp = n_dp; p.w = w; p.h = h; p.basefocal = cfg.basefocal;
        p.n_rand_samples = cfg.depth_rand_samples; p.global_prop_step = cfg.depth_global_prop_step; p.local_prop_width = cfg.depth_local_prop_width;
        p.lambda = cfg.lambda; p.omega = cfg.omega; p.disp_delta = has_disparity ? cfg.disp_delta : -1.f; p.delta = cfg.delta;
        p.fb_smooth = cfg.fb_smooth != 0; p.s0_ems_prob = cfg.fb_emm; p.no_change_prob = cfg.fb_no_change_prob;
        p.range_factor = cfg.depth_range_factor; p.update_rigidness_only = (flag == OD_UPDATE_RIGIDNESS_ONLY);
        p.strict = strict; p.ref_rng = ref_rng; p.ref_tex = ref_tex;
        if (strict && cfg.reference_stale_depth && cfg.exclusive_gpu_context && cfg.norm_world_scale && n_dp == 0) {
            if (int e = c->stale_depth.reserve(sizeof(float) * (size_t)w * h)) return e;
            p.stale_depth = c->stale_depth.as<float>();
            p.stale_refresh = iters_cur < 2;  // voldor.cpp:250: "iters_cur == 0 || iters_cur == 1": the calls that upload the map
        }
        p.defer_reduce = defer_reduce && !strict && debug_switches().defer_reduce != 0;
        p.fb_done = flag != OD_ONLY_USE_DEPTH_PRIOR ? fb_rode : 0;  // (enqueue_cameras: fb_smooth ran during the pose half: in the mode kernels' launches, or on the side stream)
        p.cum_done = p.fb_done && cum_rode;                         // (and the last mode kernel prepared the projective maps)
        fb_rode = 0; cum_rode = false;
        if (with_world_scale) p.world_scale_out = world_scale_ptr();  // voldor.cpp:309-317: the pose half rides on the density launch, the depth half follows
        return optimize_depth_device(c, c->od, p);  // with world_scale_out: depth and poses leave normalised (voldor.cpp:309-317)
    }

    // voldor/geometry.cpp:5-265, all on the device; success / density come back in CamState
    float* world_scale_ptr() const { return c->ms_io.as<float>() + 48; }  // (ms_io is reserved at its full size in init(): the pointer holds for the window)
    int optimize_camera_pose(int i, bool rg_refine, bool last = false, const FbRide* ride = nullptr) {
        ImageSet& S = c->od;
        if (c->prof) prof_begin(c);
        const int solver = cfg.lambdatwist ? (cfg.cpu_p3p ? 2 : 0) : 1;  // cpu_p3p=1 selects the reference's CPU instantiation lambdatwist_p4p<double,...> (geometry.cpp:112)
        const bool ref_svd = cfg.reference_svd < 0 ? reference_svd_default() : cfg.reference_svd != 0;
        if (int e = collect_device(c, S, n_flows, w, h, i, cfg.rigidness_threshold, cfg.rigidness_sum_threshold,
                                   cfg.pose_sample_min_depth, cfg.pose_sample_max_depth, cfg.max_trace_on_flow, dcams() + i, false,
                                   /*block_compact=*/cfg.reference_draw != 0, ref_tex))  // the index draw reads (block, rank in block) directly
            return e;
        ModeParams mp{};
        mp.dims = 6; mp.kernel_var = cfg.meanshift_kernel_var; mp.ms_epsilon = cfg.meanshift_epsilon;
        mp.ms_max_iters = cfg.meanshift_max_iters; mp.ms_max_init_trials = cfg.meanshift_max_init_trials;
        mp.ms_good_init_confidence = cfg.meanshift_good_init_confidence; mp.use_external_init_mean = -1;
        mp.rvec_scale = cfg.meanshift_rvec_scale; mp.rg_pose_scaling = cfg.rg_pose_scaling; mp.do_rg = rg_refine ? 1 : 0;
        mp.rg_trunc_sigma = cfg.rg_trunc_sigma; mp.rg_covar_reg_lambda = cfg.rg_covar_reg_lambda; mp.rg_epsilon = cfg.rg_epsilon;
        mp.rg_max_iters = cfg.rg_max_iters;
        if (last) {  // the kernel that finishes the last camera also takes the truncation decision (PoseBlock::n_active)
            mp.decide_n = n_flows; mp.decide_allow_trunc = iters_cur > cfg.no_trunc_iters ? 1 : 0;
            mp.decide_trunc_rigidness_density = cfg.trunc_rigidness_density; mp.decide_trunc_sample_density = cfg.trunc_sample_density;
            mp.host_brief = c->h_brief_dev;
        }
        if (int e = solve_from_maps_device(c, w * h, cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.n_poses_to_sample, solver, dcams() + i,
                                           cfg.reference_draw ? 1 : 0, strict, ref_svd, ref_rng))
            return e;
        if (strict) { if (int e = pose_mode_strict_device(c, cfg.n_poses_to_sample, mp, dcams() + i, S.pb(), i)) return e; }
        else if (int e = pose_mode_device(c, cfg.n_poses_to_sample, mp, dcams() + i, S.pb(), i, hcams[i].pose_sample_count == 0, ride)) return e;
        if (c->prof) prof_end(c, "optimize_camera_pose");
        return 0;
    }

    // voldor.cpp:164-201.  The reference decides after every camera (on the host) whether to go on; here all cameras of
    // the iteration are enqueued back to back, the truncation rule runs ON THE DEVICE (decide_active, at the end of the last camera's pose kernel -> PoseBlock::
    // n_active, which the depth kernels clamp to), and the host applies the same rule to its copy of the records only
    // after it has already enqueued this iteration's depth half: the GPU never waits for the host decision.
    // Equivalent to the reference order: a camera that fails or is skipped truncates the window at its index, so whatever
    // the speculatively executed later cameras wrote (their own pose slots only) is never read again.
    // fb_smooth of the depth half riding in the pose half (FbRide, vk_common.hpp).  The mode kernel of a camera keeps ONE compute unit busy for ~13 us;
    // fb_smooth -- the first two (with depth priors: four) launches of the depth half -- depends on nothing the pose half computes: the rigidness maps
    // are only READ there, by the traces.  So its 256-thread blocks are dealt over the mode kernels of the iteration's cameras as extra workgroups: the
    // row blocks first (rigidness maps out of place, rig -> rig2: the traces of the later cameras still read rig; prior confidences in place), then, in
    // later launches, the column blocks (in place on rig2); the last mode kernel also prepares the projective maps of the depth half.  rig and rig2
    // then change names and optimize_depth starts at its cost kernel.  Same arithmetic on the same values: every output bit of a window unchanged.
    // Not in strict mode, not in an iteration with the refit (that kernel's LDS leaves no room), not where a pass needs 40-step segments (1080p: there
    // fb_smooth is 110 us of memory pass, not launch latency), not when the blocks do not fit (at most 480 per launch: riders and the mode kernel's own
    // workgroup should not have to share a compute unit).
    FbRidePlan fbp;
    int fb_rode = 0;  // fb_smooth of the coming depth half ran during the pose half: bit 0 the rigidness maps, bit 1 the prior confidences
    void plan_fb_ride(bool rg) {
        fbp = FbRidePlan();
        if (strict || rg || !cfg.fb_smooth || !cfg.optimize_depth || !debug_switches().fb_ride || !g_window_alone) return;
        ImageSet& S = c->od;
        if (!S.rig2.p) return;
        fb_ride_plan(w, h, n_flows, n_dp, S.rig.as<float>(), S.rig2.as<float>(), S.confs.as<float>(), &fbp);
    }
    FbRide ride_of_camera(int i, bool cum_ok) const {
        FbRide r = fb_ride_of_camera(fbp, i, n_flows, w, h, cfg.fb_emm, cfg.fb_no_change_prob);
        if (cum_ok && i == n_flows - 1) { r.cum_N = n_flows; r.cum_Ndp = n_dp; r.world_scale = (cfg.norm_world_scale && n_dp == 0) ? world_scale_ptr() : nullptr; }
        return r;
    }
    // fb_smooth of the coming depth half NEXT TO the pose half on a second stream (round 6), in strict mode: the reference's recurrence step by step is
    // ~100 waves walking w or h dependent steps with two IEEE divisions each -- 0.2 ms per EM iteration at 640x480, 0.3 ms at 1241x376 -- on a chip the
    // strict pose half (16 waves of mode kernel, 128 of P3P) leaves just as idle.  Same dependencies as the riders: the smoothing reads what the last
    // E-step left and the pose half only READS the rigidness maps (the traces), so the row pass writes rig -> rig2, the column pass works on rig2, the
    // prior confidences in place (no trace reads them); the streams fork after the last launch before the cameras and join before the depth half; rig
    // and rig2 then change names.  The kernels are the depth half's own: every output bit of a window unchanged (tests/test_gpu_riders.py).  Measured,
    // reference mode: cfg2 13.5 -> 12.2 ms, cfg3 29.2 -> 24.7 ms per window.  NOT in the fast mode: there the one pass that cannot ride (40-step segments,
    // 1080p) is 110 us of memory traffic that then competes with the traces of the pose half -- 28.2 -> 28.4 ms per cfg5 window, measured and left out
    // (profiles/r06_summary.md).
    bool cum_rode = false, side_used = false;
    bool plan_fb_side() const {
        return strict && !fbp.on && cfg.fb_smooth && cfg.optimize_depth && debug_switches().fb_side && g_window_alone && c->od.rig2.p && n_flows >= 1;
    }
    int launch_fb_side() {
        ImageSet& S = c->od;
        if (int e = c->ensure_side_stream()) return e;
        VK_CHECK(hipEventRecord(c->ev_fork, c->stream));
        VK_CHECK(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
        if (int e = fb_smooth_strict_device(c, S.rig.as<float>(), n_flows, w, h, cfg.fb_emm, cfg.fb_no_change_prob, nullptr, S.rig2.as<float>(), c->side_stream)) return e;
        if (int e = fb_smooth_strict_device(c, S.confs.as<float>(), n_dp, w, h, cfg.fb_emm, cfg.fb_no_change_prob, nullptr, nullptr, c->side_stream)) return e;
        VK_CHECK(hipEventRecord(c->ev_join, c->side_stream));
        return 0;
    }
    int enqueue_cameras() {
        const bool rg = cfg.rg_refine && (!cfg.rg_refine_last_only || iters_remain == 0);
        plan_fb_ride(rg);
        const bool side = plan_fb_side();
        if (side) { side_used = true; if (int e = launch_fb_side()) return e; c->dbg_fb_side_passes++; }
        const bool during = fbp.on || side;      // fb_smooth of the coming depth half happens during this pose half
        const bool cum_ok = during && !strict && !rg;  // ... and the last mode kernel (not the refit kernel: no LDS left) can prepare the projective maps
        for (int i = 0; i < n_flows; i++) {
            // (first EM iteration of a host-memory call: frame i arrives while camera i - 1 runs.  CUDA's linear filter over the STACK of layers
            // -- --reference_tex 1 -- blends the bottom row of layer i with the top row of layer i + 1 (vk_ref_cuda.h:133-149): one frame more)
            if (int e = upload_frames_up_to(ref_tex ? i + 2 : i + 1)) return e;
            const FbRide ride = ride_of_camera(i, cum_ok);
            if (int e = optimize_camera_pose(i, rg, i == n_flows - 1, (fbp.on || (cum_ok && i == n_flows - 1)) ? &ride : nullptr)) return e;
        }
        if (side) VK_CHECK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
        if (during) { std::swap(c->od.rig, c->od.rig2); fb_rode = (side || fbp.n_stacks == 2 || n_dp == 0) ? 3 : 1; cum_rode = cum_ok; }  // the smoothed maps are `rig` from here on
        if (int e = upload_frames_up_to(n_flows_init)) return e;
        // rigidness densities (reduced on the device by the last optimize_depth, voldor.cpp:171) and results: the last camera's kernel
        // has stored what the host needs into pinned memory (CamBrief); the event marks it complete
        VK_CHECK(hipEventRecord(c->ev_cams, c->stream));
        return 0;
    }
    int finish_cameras() {
        const bool allow_trunc = iters_cur > cfg.no_trunc_iters;
        VK_CHECK(hipEventSynchronize(c->ev_cams));
        for (int i = 0; i < n_flows; i++) {
            const CamBrief& b = c->h_brief[i];
            hcams[i].success = b.success; hcams[i].pose_sample_count = b.pose_sample_count; hcams[i].last_used_ms_iters = b.last_used_ms_iters;
            hcams[i].last_used_gu_iters = b.last_used_gu_iters; hcams[i].pose_density = b.pose_density; hcams[i].pose_rigidness_density = b.pose_rigidness_density;
        }
        for (int i = 0; i < n_flows; i++) {
            int ok = 0;
            if (!allow_trunc || hcams[i].pose_rigidness_density > cfg.trunc_rigidness_density) ok = hcams[i].success;
            if (!cfg.silent) print_cam(i);
            if (!ok || (allow_trunc && hcams[i].pose_density < cfg.trunc_sample_density)) {
                if (!cfg.silent) std::cout << "truncated at camera " << i << std::endl;
                iters_remain = std::max(iters_remain, cfg.min_iters_after_trunc);
                n_flows = i;
                break;
            }
        }
        return 0;
    }
    void print_cam(int i) {  // Camera::print_info (utils.h:66-76), without the OpenCV-derived lines
        const CamState& s = hcams[i];
        std::cout << "pose pool size = " << s.pose_sample_count << std::endl
                  << "rigidness density = " << s.pose_rigidness_density << std::endl
                  << "pose density = " << s.pose_density << std::endl
                  << "last used meanshift iters = " << s.last_used_ms_iters << std::endl
                  << "last used gu iters = " << s.last_used_gu_iters << std::endl << std::endl;
    }

    int bootstrap_points_of() const {  // --bootstrap_points, -1 = the default: five-point in the fast mode, 8-point in strict mode (Config::bootstrap_points)
        if (cfg.bootstrap_points == 5 || cfg.bootstrap_points == 8) return cfg.bootstrap_points;
        const int d = debug_switches().bootstrap_default;
        return d ? d : (strict ? 8 : 5);
    }
    // voldor.cpp:130-149
    int solve() {
        if (int e = upload_frames_up_to(ref_tex ? 2 : 1)) return e;
        if (n_dp == 0) {  // bootstrap :151-162
            if (c->prof) prof_begin(c);
            if (int e = bootstrap_device(c, c->od, w, h, cfg.fx, cfg.fy, cfg.cx, cfg.cy, dcams(), strict, bootstrap_points_of())) return e;
            if (c->prof) prof_end(c, "bootstrap");
        }
        while (iters_remain > 0 && n_flows > 0) {
            iters_cur++; iters_remain--;
            if (int e = enqueue_cameras()) return e;
            // the depth half is enqueued with the pre-decision frame count; on the device it runs with n_active
            // (another EM iteration follows for certain: its first launch, the trace of camera 0, closes this E-step's density reduction)
            if (int e = optimize_depth(cfg.optimize_depth ? OD_DEFAULT : OD_UPDATE_RIGIDNESS_ONLY, cfg.norm_world_scale && n_dp == 0, iters_remain > 0)) return e;
            if (int e = finish_cameras()) return e;
        }
        return flush_pending_reduce(c);  // (the window was truncated to nothing in between)
    }
};

static thread_local Voldor g_last;  // stats of the last window (vk_last_camera_stats)

// The result of a window as ONE device-resident record, [n_registered | poses N x 6 | covariances N x 36] floats (unregistered
// slots zero): what a multi-GPU launcher all-gathers (sendcount 1 + 6N + 36N), packed where the data is
// instead of a D2H -> numpy -> H2D hop around an 844-byte collective.
__global__ static void k_pack_pose_block(const CamState* __restrict__ cams, int n_registered, int N, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, len = 1 + 42 * N;
    if (i >= len) return;
    float v = 0.f;
    if (i == 0) v = (float)n_registered;
    else if (i < 1 + 6 * N) { const int c = (i - 1) / 6, d = (i - 1) % 6; if (c < n_registered) v = d < 3 ? cams[c].rvec[d] : cams[c].t[d - 3]; }
    else { const int c = (i - 1 - 6 * N) / 36, k = (i - 1 - 6 * N) % 36; if (c < n_registered) v = cams[c].covar[k]; }
    out[i] = v;
}

static int voldor_run_on(Context* c, const float* flows, const float* disparity, const float* disparity_pconf, const float* depth_priors,
                         const float* depth_prior_poses, const float* depth_prior_pconfs, float fx, float fy, float cx, float cy,
                         float basefocal, int N, int N_dp, int w, int h, const char* config, int* n_registered, float* poses,
                         float* poses_covar, float* depth, float* depth_conf, float* pose_block_dev = nullptr) {
    if (!c) return (int)hipErrorNoDevice;
    Voldor& v = g_last;
    v = Voldor();
    v.c = c;
    // every exit, the failing ones included: a density reduction left for a trace that will never come must not survive the window (it
    // holds raw pointers into buffers a later call may re-reserve), and the copy stream must be done with the caller's flow buffer
    struct WindowGuard {
        Context* c; Voldor* v;
        ~WindowGuard() {
            c->pending_reduce = ReduceArgs();
            if (v->copies_in_flight && c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); v->copies_in_flight = false; }
            if (v->side_used && c->side_stream) (void)hipStreamSynchronize(c->side_stream);  // (a window that failed between fork and join: nothing of it may still run when the next one starts)
        }
    } guard{ c, &v };
    v.cfg.fx = fx; v.cfg.cx = cx; v.cfg.fy = fy; v.cfg.cy = cy; v.cfg.basefocal = basefocal;  // py_export.cpp:19-25
    if (int e = v.cfg.read_config(config ? config : "")) return 1000 + e;
    if (int e = v.init(flows, disparity, disparity_pconf, depth_priors, depth_prior_poses, depth_prior_pconfs, N, N_dp, w, h)) return e;
    if (int e = v.solve()) return e;
    // outputs: py_export.cpp:56-76
    const size_t npx = (size_t)w * h;
    VK_CHECK(hipMemcpyAsync(c->h_cams, c->cams.p, sizeof(CamState) * MAX_FRAMES, hipMemcpyDeviceToHost, c->stream));  // pinned: the host does not wait here
    if (depth) VK_CHECK(hipMemcpyAsync(depth, c->od.depth.p, sizeof(float) * npx, hipMemcpyDefault, c->stream));
    if (depth_conf) {
        if (int e = c->tmp.reserve(sizeof(float) * npx)) return e;
        if (int e = depth_conf_device(c, c->od.rig.as<float>(), c->od.confs.as<float>(), c->tmp.as<float>(), v.n_flows, v.n_dp, npx)) return e;
        VK_CHECK(hipMemcpyAsync(depth_conf, c->tmp.p, sizeof(float) * npx, hipMemcpyDefault, c->stream));
    }
    if (pose_block_dev) {
        hipLaunchKernelGGL(k_pack_pose_block, dim3((1 + 42 * N + 255) / 256), dim3(256), 0, c->stream, c->cams.as<CamState>(), v.n_flows, N, pose_block_dev);
        VK_CHECK_LAST();
    }
    VK_CHECK(hipStreamSynchronize(c->stream));
    memcpy(v.hcams, c->h_cams, sizeof(CamState) * MAX_FRAMES);
    if (n_registered) *n_registered = v.n_flows;
    for (int i = 0; i < v.n_flows; i++) {
        if (poses) { memcpy(poses + i * 6, v.hcams[i].rvec, 12); memcpy(poses + i * 6 + 3, v.hcams[i].t, 12); }
        if (poses_covar) memcpy(poses_covar + i * 36, v.hcams[i].covar, sizeof(float) * 36);
    }
    return 0;
}
static int voldor_run(const float* flows, const float* disparity, const float* disparity_pconf, const float* depth_priors,
                      const float* depth_prior_poses, const float* depth_prior_pconfs, float fx, float fy, float cx, float cy,
                      float basefocal, int N, int N_dp, int w, int h, const char* config, int* n_registered, float* poses,
                      float* poses_covar, float* depth, float* depth_conf) {
    return voldor_run_on(default_context(), flows, disparity, disparity_pconf, depth_priors, depth_prior_poses, depth_prior_pconfs, fx, fy,
                         cx, cy, basefocal, N, N_dp, w, h, config, n_registered, poses, poses_covar, depth, depth_conf);
}

// ---- several independent windows in flight on ONE device ------------------------------------------------------------
// Half of a window's GPU time is spent in single-workgroup kernels (mean-shift, robust-Gaussian refit) and 128-wave
// kernels (P3P) that cannot be batched across the cameras of a window (camera i+1 needs the pose of camera i,
// voldor.cpp:169-195): one CU of 256 is busy.  Independent windows (other sequences, or other windows of an offline
// sequence) have no such dependence, so a batch runs window b on its own context = own HIP stream + own buffers, driven
// by its own host thread (each window needs one host decision per EM iteration), and the hardware queues overlap the
// narrow kernels of one window with the wide per-pixel kernels of the others.  Results are those of the one-at-a-time
// call (same kernels, per-context depth-sampling counter).
struct BatchJob {
    const float *flows, *disparity, *disparity_pconf, *depth_priors, *depth_prior_poses, *depth_prior_pconfs;
    float fx, fy, cx, cy, basefocal; int N, N_dp, w, h; const char* config;
    int* n_registered; float *poses, *poses_covar, *depth, *depth_conf;
    int rc;
    uint32_t epoch0, epoch_end;  // depth-sampling counter the window starts from / ends at
};
class WindowPool {  // persistent workers: worker i owns pool context i of the device it was started on
    std::mutex mu, run_mu; std::condition_variable cv_work, cv_done;  // run_mu: one batch at a time per device (callers queue)
    std::vector<std::thread> workers; std::vector<BatchJob>* jobs = nullptr; size_t next = 0; int pending = 0, device = 0; bool stop = false;
    int cur_in_flight = 1;  // workers the current batch keeps busy
    void loop(int i) {
        (void)hipSetDevice(device);
        for (;;) {
            BatchJob* j;
            {   // windows are handed out one at a time: a worker that finishes early starts the next one, no static split
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || (jobs && next < jobs->size()); });
                if (stop) return;
                j = &(*jobs)[next++];
            }
            // every window of a batch starts from the batch's epoch, whichever worker / context picks it up: the result of a
            // window does not depend on how the batch was scheduled
            if (Context* pc = pool_context(i)) { pc->rand_epoch = j->epoch0; pc->rand_w = pc->rand_h = -1; }
            g_window_alone = cur_in_flight <= 1;
            j->rc = voldor_run_on(pool_context(i), j->flows, j->disparity, j->disparity_pconf, j->depth_priors, j->depth_prior_poses,
                                  j->depth_prior_pconfs, j->fx, j->fy, j->cx, j->cy, j->basefocal, j->N, j->N_dp, j->w, j->h, j->config,
                                  j->n_registered, j->poses, j->poses_covar, j->depth, j->depth_conf);
            if (Context* pc = pool_context(i)) j->epoch_end = pc->rand_epoch;
            { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) { jobs = nullptr; cv_done.notify_all(); } }
        }
    }
public:
    explicit WindowPool(int dev) : device(dev) {}
    ~WindowPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv_work.notify_all(); for (auto& t : workers) t.join(); }
    void run(std::vector<BatchJob>& js, int width) {
        std::lock_guard<std::mutex> serial(run_mu);  // cv_done.wait releases `mu`: without this a second caller could overwrite jobs / next / pending
        std::unique_lock<std::mutex> lk(mu);
        const size_t want = std::min(js.size(), (size_t)std::max(1, width));
        while (workers.size() < want) { const int i = (int)workers.size(); workers.emplace_back([this, i] { loop(i); }); }
        // more workers than `width` may exist from an earlier, wider call: they simply take windows too (each has its own context)
        jobs = &js; next = 0; pending = (int)js.size(); cur_in_flight = (int)std::min(js.size(), std::max(want, workers.size()));
        cv_work.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};
static std::mutex g_pool_mu;
static std::map<int, WindowPool*> g_pools;  // per device; intentionally never destroyed (worker threads outlive static teardown order)
static int voldor_run_batch(std::vector<BatchJob>& jobs, int width) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipErrorNoDevice;
    WindowPool* p;
    { std::lock_guard<std::mutex> lk(g_pool_mu); auto& q = g_pools[dev]; if (!q) q = new WindowPool(dev); p = q; }
    const uint32_t e0 = batch_rand_epoch(dev, nullptr);
    for (auto& j : jobs) { j.epoch0 = e0; j.epoch_end = e0; }
    p->run(jobs, width);
    batch_rand_epoch(dev, &jobs[0].epoch_end);  // the next batch continues where window 0 of this one stopped
    for (auto& j : jobs) if (j.rc) return j.rc;
    return 0;
}

}  // namespace vk

// voldor/py_export.h:3-11
int py_voldor_wrapper(const float* flows, const float* disparity, const float* disparity_pconf, const float* depth_priors,
                      const float* depth_prior_poses, const float* depth_prior_pconfs, const float fx, const float fy,
                      const float cx, const float cy, const float basefocal, const int N, const int N_dp, const int w, const int h,
                      const char* config, int& n_registered, float* poses, float* poses_covar, float* depth, float* depth_conf) {
    return vk::voldor_run(flows, disparity, disparity_pconf, depth_priors, depth_prior_poses, depth_prior_pconfs, fx, fy, cx, cy,
                          basefocal, N, N_dp, w, h, config, &n_registered, poses, poses_covar, depth, depth_conf);
}

extern "C" {
int vk_py_voldor_wrapper(const float* flows, const float* disparity, const float* disparity_pconf, const float* depth_priors,
                         const float* depth_prior_poses, const float* depth_prior_pconfs, float fx, float fy, float cx, float cy,
                         float basefocal, int N, int N_dp, int w, int h, const char* config, int* n_registered, float* poses,
                         float* poses_covar, float* depth, float* depth_conf) {
    return vk::voldor_run(flows, disparity, disparity_pconf, depth_priors, depth_prior_poses, depth_prior_pconfs, fx, fy, cx, cy,
                          basefocal, N, N_dp, w, h, config, n_registered, poses, poses_covar, depth, depth_conf);
}
int vk_voldor_device(const float* flows, const float* disparity, const float* disparity_pconf, const float* depth_priors,
                     const float* depth_prior_poses, const float* depth_prior_pconfs, float fx, float fy, float cx, float cy,
                     float basefocal, int N, int N_dp, int w, int h, const char* config, int* n_registered, float* poses,
                     float* poses_covar, float* depth, float* depth_conf) {
    return vk::voldor_run(flows, disparity, disparity_pconf, depth_priors, depth_prior_poses, depth_prior_pconfs, fx, fy, cx, cy,
                          basefocal, N, N_dp, w, h, config, n_registered, poses, poses_covar, depth, depth_conf);
}
int vk_voldor_device_block(const float* flows, const float* disparity, const float* disparity_pconf, const float* depth_priors,
                           const float* depth_prior_poses, const float* depth_prior_pconfs, float fx, float fy, float cx, float cy,
                           float basefocal, int N, int N_dp, int w, int h, const char* config, int* n_registered, float* poses,
                           float* poses_covar, float* depth, float* depth_conf, float* pose_block_dev) {
    return vk::voldor_run_on(vk::default_context(), flows, disparity, disparity_pconf, depth_priors, depth_prior_poses, depth_prior_pconfs, fx, fy,
                             cx, cy, basefocal, N, N_dp, w, h, config, n_registered, poses, poses_covar, depth, depth_conf, pose_block_dev);
}
int vk_voldor_device_batch(int n_windows, const float* const* flows, const float* const* disparity, const float* const* disparity_pconf,
                           const float* const* depth_priors, const float* const* depth_prior_poses,
                           const float* const* depth_prior_pconfs, float fx, float fy, float cx, float cy, float basefocal, int N, int N_dp,
                           int w, int h, const char* config, int* n_registered, float* poses, float* poses_covar, float* const* depth,
                           float* const* depth_conf) {
    if (n_windows <= 0 || !flows || !n_registered) return (int)hipErrorInvalidValue;
    std::vector<vk::BatchJob> jobs((size_t)n_windows);
    for (int b = 0; b < n_windows; b++) {
        vk::BatchJob& j = jobs[(size_t)b];
        j.flows = flows[b]; j.disparity = disparity ? disparity[b] : nullptr; j.disparity_pconf = disparity_pconf ? disparity_pconf[b] : nullptr;
        j.depth_priors = depth_priors ? depth_priors[b] : nullptr; j.depth_prior_poses = depth_prior_poses ? depth_prior_poses[b] : nullptr;
        j.depth_prior_pconfs = depth_prior_pconfs ? depth_prior_pconfs[b] : nullptr;
        j.fx = fx; j.fy = fy; j.cx = cx; j.cy = cy; j.basefocal = basefocal; j.N = N; j.N_dp = N_dp; j.w = w; j.h = h; j.config = config;
        j.n_registered = n_registered + b; j.poses = poses ? poses + (size_t)b * N * 6 : nullptr;
        j.poses_covar = poses_covar ? poses_covar + (size_t)b * N * 36 : nullptr;
        j.depth = depth ? depth[b] : nullptr; j.depth_conf = depth_conf ? depth_conf[b] : nullptr; j.rc = 0;
    }
    // at most VOLDOR_HIP_INFLIGHT (default 4) windows in flight at a time; the rest of the batch queues behind them
    int width = 4;
    if (const char* e = getenv("VOLDOR_HIP_INFLIGHT")) { const int v = atoi(e); if (v > 0) width = v; }
    return vk::voldor_run_batch(jobs, width);
}
int vk_last_camera_stats(int* pose_sample_count, float* pose_density, float* pose_rigidness_density, int* ms_iters, int* gu_iters, int n) {
    for (int i = 0; i < n && i < vk::MAX_FRAMES; i++) {
        const vk::CamState& s = vk::g_last.hcams[i];
        if (pose_sample_count) pose_sample_count[i] = s.pose_sample_count;
        if (pose_density) pose_density[i] = s.pose_density;
        if (pose_rigidness_density) pose_rigidness_density[i] = s.pose_rigidness_density;
        if (ms_iters) ms_iters[i] = s.last_used_ms_iters;
        if (gu_iters) gu_iters[i] = s.last_used_gu_iters;
    }
    return 0;
}
}

// vk_debug.h: the dealing of a riding fb_smooth for a window geometry, no device involved.  out: [on, seg, R, C, k_rows, then per camera kind, first,
// count]; returns the number of ints written (5 + 3 n_flows), -1 when `out` is too short
extern "C" __attribute__((visibility("default"))) int vk_debug_fb_ride_plan(int w, int h, int n_flows, int n_dp, int* out, int n_out) {
    if (n_flows < 0 || n_flows > vk::MAX_FRAMES || n_out < 5 + 3 * n_flows) return -1;
    vk::FbRidePlan P;
    alignas(16) static float dummy[4];
    vk::fb_ride_plan(w, h, n_flows, n_dp, dummy, dummy, dummy, &P);
    out[0] = P.on ? P.n_stacks : 0; out[1] = P.rows[0].seg * 100 + P.cols[0].seg; out[2] = P.R; out[3] = P.C; out[4] = P.k_rows;
    for (int i = 0; i < n_flows; i++) {
        const vk::FbRide r = vk::fb_ride_of_camera(P, i, n_flows, w, h, 0.5f, 0.9f);
        out[5 + 3 * i] = r.kind; out[6 + 3 * i] = r.first; out[7 + 3 * i] = r.count;
    }
    return 5 + 3 * n_flows;
}
