// voldor_amd/csrc/vk_common.hpp -- shared host-side plumbing for the MI355X (gfx950) VOLDOR
// kernel library: error handling, device buffers, the per-device context.
//
// The reference keeps file-static device buffers per translation unit
// (gpu-kernels/optimize_depth.cu:45-52, collect_p3p_instances.cu:29-34), which makes it
// single-device and non re-entrant.  Here every buffer hangs off a Context that is bound to
// one HIP device and one stream, so one process per GPU (or several contexts) is possible.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <string>
#include <map>

namespace vk {

constexpr int MAX_FRAMES = 16;       // gpu-kernels/optimize_depth.cu:20
constexpr int MAX_DISP_FRAMES = 16;  // gpu-kernels/optimize_depth.cu:21
constexpr unsigned RAND_SEED = 233;  // gpu-kernels/utils.h:18
constexpr int MAX_POSE_DIMS = 16;    // gpu-kernels/meanshift.cu:5

// gpuErrchk equivalent (gpu-kernels/utils.h:21-26): print, return the error code as int.
#define VK_CHECK(expr)                                                                         \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            fprintf(stderr, "GPUassert : %s\n%s at line %d\n", hipGetErrorString(_e), __FILE__, \
                    __LINE__);                                                                 \
            return (int)_e;                                                                    \
        }                                                                                      \
    } while (0)
#define VK_CHECK_LAST() VK_CHECK(hipGetLastError())

// Device-resident camera intrinsics + poses.  Kernels read it through a pointer (uniform
// address -> scalar loads); the pose kernels write new poses straight into it, so the EM
// loop never round-trips poses through the host.  Layout mirrors the reference's
// __constant__ block (optimize_depth.cu:24-29).
struct PoseBlock {
    float K4[4];   // fx, cx, fy, cy
    float K4i[4];  // 1/fx, -cx/fx, 1/fy, -cy/fy
    float Rs[MAX_FRAMES][9];
    float ts[MAX_FRAMES][3];
    float dpRs[MAX_DISP_FRAMES][9];
    float dpts[MAX_DISP_FRAMES][3];
    // number of flow frames still registered, decided ON THE DEVICE after the cameras of an EM iteration
    // (decide_active in vk_pose.hip, voldor.cpp:187-194): the depth kernels clamp their frame count to it, so the host can enqueue them
    // before it has seen the decision itself.  B-inner callers pass the count by argument: MAX_FRAMES here.
    int n_active;
    // bit f: depth prior f sits at the identity pose (R = I, t = 0 exactly: the disparity / RGB-D prior of the reference frame).  Written by
    // cum_poses_block (fast path); such a prior is sampled AT the pixel -- no projection, no bilinear arithmetic (prior_parts, vk_depth.hip)
    int dp_ident;
    int pad_[2];
    // Fast path only (k_cum_poses, vk_depth.hip): the rigid chain of optimize_depth.cu:54-81 folded into one projective map per
    // frame.  With (Rc_f, tc_f) = the pose that takes frame-0 coordinates to frame f+1 (Rc_f = R_f Rc_{f-1}, tc_f = R_f tc_{f-1} + t_f)
    // the homogeneous pixel of (x, y, depth d) in frame f+1 is  d * cumM[f] (x, y, 1)^T + cumT[f],  cumM = K Rc K^-1, cumT = K tc;
    // its third component is the camera-space depth.  dpM / dpT: the same for the depth-prior poses (each relative to frame 0).
    float cumM[MAX_FRAMES][9];
    float cumT[MAX_FRAMES][4];
    float dpM[MAX_DISP_FRAMES][9];
    float dpT[MAX_DISP_FRAMES][4];
};

// Per-camera state kept on the device (voldor/utils.h:31-45 Camera, minus OpenCV).
struct CamState {
    float rvec[3];
    float t[3];
    float covar[36];
    float pose_density;
    float pose_rigidness_density;
    int pose_sample_count;
    int last_used_ms_iters;
    int last_used_gu_iters;
    int success;   // result of the last optimize_camera_pose (geometry.cpp:5-265 return value)
    int n_points;  // correspondences that survived compaction (geometry.cpp:68-88)
    int pad;
};

// What the host needs from a camera record to take the truncation decision of voldor.cpp:171-194 (and to print it): the kernel that
// finishes the last camera of an EM iteration stores these straight into pinned host memory (no copy command between the pose
// half and the depth half).
struct CamBrief { int success, pose_sample_count, last_used_ms_iters, last_used_gu_iters; float pose_density, pose_rigidness_density; };

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // grow-only allocation, like GMat::create(lazy) (gpu-kernels/gmat.h:19-33)
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            fprintf(stderr, "GPUassert : %s (hipMalloc %zu bytes)\n", hipGetErrorString(e), bytes);
            return (int)e;
        }
        cap = bytes;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// One set of window images (flows, rigidness, depth, priors ...), row-major, x fastest,
// layer stride w*h (the reference's GMat contract, gmat.h:171-173, without the pitch).
struct ImageSet {
    int w = 0, h = 0;
    DevBuf flows;   // [N][h][w] float2
    DevBuf rig;     // [N][h][w]
    DevBuf rig2;    // [N][h][w]: where the row pass of a riding fb_smooth puts the maps (FbRide); swapped with `rig` when the pose half ends
    DevBuf depth;   // [h][w]
    DevBuf cost;    // [h][w]
    DevBuf priors, pconfs, confs;  // [N_dp][h][w]
    DevBuf pose;    // PoseBlock
    bool pose_init = false;
    int ensure_pose() {
        int e = pose.reserve(sizeof(PoseBlock));
        if (e) return e;
        if (!pose_init) {
            if (hipMemset(pose.p, 0, sizeof(PoseBlock)) != hipSuccess) return 1;
            const int all = MAX_FRAMES;
            if (hipMemcpy(&pb()->n_active, &all, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return 1;
            pose_init = true;
        }
        return 0;
    }
    PoseBlock* pb() const { return pose.as<PoseBlock>(); }
};

struct OdParams {  // scalar arguments of optimize_depth_gpu (gpu_kernels.h:44-58)
    float abs_resize_factor = 1.f;
    int N = 0, N_dp = 0, w = 0, h = 0;
    float basefocal = 0.f;
    int n_rand_samples = 10, global_prop_step = 8, local_prop_width = 32;
    float lambda = 0.15f, omega = 0.15f, disp_delta = 1.f, delta = 0.5f;
    bool fb_smooth = true;
    float s0_ems_prob = 0.5f, no_change_prob = 0.9f, range_factor = 1.f;
    bool update_rigidness_only = false;
    bool strict = false;  // strict-math mode: reference-order arithmetic on vk_strict_math.h (vk_strict.hip)
    // reference mode, round 4 (vk_ref_cuda.h; strict kernels only): cuRAND XORWOW streams for the depth samples / CUDA's 8-bit-fraction
    // linear filter over the stacked layers for every at_tex of the reference (D1 / D2 switched off)
    bool ref_rng = false, ref_tex = false;
    float* world_scale_out = nullptr;  // device float: also run normalize_world_scale's pose half (voldor.cpp:309-317) in the last launch
    // window pipeline, fast mode: the density reduction that closes the call (a launch of N + 1 workgroups) is not launched but left in
    // Context::pending_reduce: the correspondence trace of camera 0 -- the next launch of the stream, which reads none of its results -- carries it as
    // extra workgroups.  One dependent launch per EM iteration less (2.85 us boundary + 4.9 us kernel).
    bool defer_reduce = false;
    int fb_done = 0;       // window pipeline: fb_smooth of this call was done during the pose half (riding in the mode kernels' launches, FbRide; or on the side stream): bit 0 the rigidness maps, bit 1 the prior confidences
    bool cum_done = false; // fast mode: so were the projective maps of the chain and the world-scale factor (cum_poses_block in the last mode kernel)
    // --reference_stale_depth 1 (strict mode; SURVEY Appendix B-1, deviation D4 switched off): the depth map optimize_depth.cu keeps on the device.
    // With exclusive_gpu_context the reference uploads its depth map for the first call only (voldor.cpp:250-291), so from the second EM
    // iteration on the search starts from a copy that never saw normalize_world_scale().  Non-null: the kernels of this call work on that copy
    // (refreshed from the window's map first when stale_refresh), the window's map then receives the result (and, alone, the world scale).
    float* stale_depth = nullptr;
    bool stale_refresh = false;
};

// fb_smooth blocks riding in a mode kernel's launch (vk_pose.hip mode_fb_ride): a contiguous range of the 256-thread blocks of ONE pass over up to two
// stacks of maps (the rigidness maps, the prior confidences).  The window pipeline deals the row blocks (rigidness maps out of place: the traces of the
// later cameras still read them) and then the column blocks over the mode kernels of an EM iteration's cameras -- launches that keep one compute unit
// busy -- so that the depth half starts at its cost kernel: two to four dependent launches per EM iteration less (vk_voldor.hip plan_fb_ride).
struct FbStack { const float* src = nullptr; float* dst = nullptr; int n_maps = 0, S = 0, CW = 0, vec4 = 0, blocks_x = 1, n_blocks = 0, seg = 12; };  // seg: steps per lane of this pass over this stack (12 | 20 | 40)
struct FbRide {
    int kind = 0;   // 0: nothing rides; 1: row pass src -> dst; 2: column pass in place on dst
    // slots [first, first + count) of the pass: stack 0's blocks, padded to an even number (`split`), then stack 1's.  first, split and the share of a launch are even, a
    // riding workgroup takes two consecutive slots: its two halves always work on ONE stack -- one segment length, one access form, the same barriers
    int first = 0, count = 0, split = 0;
    int w = 0, h = 0;
    float e0 = 0.f, p = 0.f;
    FbStack st[2];
    // last mode kernel of the pose half: its own workgroup then prepares the projective maps of the depth half (cum_poses_block); -1: no
    int cum_N = -1, cum_Ndp = 0;
    float* world_scale = nullptr;
};
// arguments of the density reduction that closes an E-step (reduce_density_block, vk_cum_poses.hpp)
struct ReduceArgs {
    const float* partial = nullptr;  // [n_launch][nblk]; NULL: nothing to reduce
    int nblk = 0, npx = 0, n_launch = 0, scale_ready = 0;
    CamState* cams = nullptr; PoseBlock* P = nullptr; float* scale_out = nullptr;
    int n_px_blocks = 0;  // riding in k_collect: workgroups of the trace itself (the reduction blocks follow them)
};

struct ProfEntry { double ms = 0; long count = 0; };

struct Context {
    int device = 0;
    hipStream_t stream = nullptr;
    // B-inner keeps the reference's two independent caches (optimize_depth.cu vs
    // collect_p3p_instances.cu statics); the B-outer pipeline uses `od` for everything.
    ImageSet od, cp;
    DevBuf rig_partial;           // per-block rigidness sums -> pose_rigidness_density
    long dbg_fb_blocks_rode = 0, dbg_reduces_rode = 0, dbg_fb_side_passes = 0;  // verification counters (vk_debug_counter): fb_smooth blocks / density reductions that rode in another kernel's launch
    ReduceArgs pending_reduce;    // window pipeline, fast mode: the density reduction of the last E-step, left for the next correspondence trace (partial != NULL: pending)
    DevBuf local_tbl;             // [h][w] candidate-cost table of a local propagation pass
    DevBuf p2_map, p3_map;        // [h*w][2], [h*w][3] (collect_p3p_instances.cu:27-34)
    DevBuf blk_counts, blk_offsets, valid_mask;  // per 256-pixel block: valid correspondences, their exclusive scan; one validity bit per pixel
    DevBuf pts2, pts3;            // compacted correspondences (geometry.cpp:68-80)
    DevBuf n_points;              // int
    int n_map_blocks = 0;         // workgroups of the last k_collect launch (length of blk_counts)
    bool maps_block_compact = false;  // layout p2_map / p3_map were last written in: valid entries at the head of each block's segment (k_collect<true>) or NaN-marked per pixel
    DevBuf rvecs, tvecs;          // [n_poses][3]
    DevBuf pool;                  // [n_poses][dims]
    DevBuf ms_io;                 // small float/int scratch for B-inner meanshift / robust fit
    DevBuf cams;                  // CamState[MAX_FRAMES]
    DevBuf tmp;                   // misc scratch (gblur, depth_conf ...)
    DevBuf fb_scratch;            // strict fb_smooth: forward messages [n_maps][h][w]
    DevBuf stale_depth;           // --reference_stale_depth 1: optimize_depth.cu's own device copy of the depth map (OdParams::stale_depth)
    DevBuf sp_coop;               // strict mode kernel, cooperative form: block sums, pool size and the grid barrier's counter (vk_strict.hip CoopGlobal)
    bool strict = false;          // strict-math mode of the B-inner entry points that use this context (vk_set_strict_math)
    // --reference_rng 1: the jump matrices T^(2^67 2^k) (vk_ref_cuda.h), the per-pixel XORWOW states of the depth samples
    // (optimize_depth.cu:286-291; they stand xw_px_epoch draws after curand_init for a xw_px_n-pixel image) and the states right after
    // curand_init(RAND_SEED, idx, 0) of the solver's hypotheses (solve_batch_lambdatwist.cu:44-48: re-seeded per call, so a fixed table)
    DevBuf xw_jumps, xw_px_states, xw_pose_states;
    DevBuf sf_qcnt, sf_qlist;  // the queues of the strict table pass's filter (k_local_table_filter -> k_local_table_exact): 4 passes x SFQ counters, SFQ lists of pixel indices
    DevBuf sf_stats;  // 4 x u64 (two used): what the fp32 filter of the strict sample pass saw / kept (verification counters, vk_debug_switch "strict_filter" 2)
    int xw_px_n = 0, xw_pose_n = 0;
    uint32_t xw_px_epoch = 0;
    uint32_t rand_epoch = 0;      // persistent depth-sampling RNG counter (optimize_depth.cu:358-361)
    int rand_w = 0, rand_h = 0;
    // profiling (off by default): HIP events on ctx.stream around kernel groups
    bool prof = false;
    std::map<std::string, ProfEntry> prof_acc;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;  // outer / inner scope
    hipEvent_t ev_cams = nullptr;  // "camera records of this EM iteration are on the host"
    // host-resident flows of a window go up frame by frame on their own non-blocking stream (no legacy null stream: a blocking hipMemcpy there would
    // synchronise with every blocking stream of the process); ev_frame[f] = "frame f is on the device", waited for by `stream` before its first reader
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_frame[MAX_FRAMES] = {};
    // fb_smooth of the coming depth half next to the pose half (round 6): a second stream forked off `stream` after the E-step and joined before the cost kernel
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    PoseBlock* h_pb = nullptr;     // pinned staging of the per-window uploads (pose block, camera records): no host sync before the first launch
    CamState* h_cams_up = nullptr;
    CamBrief* h_brief = nullptr;   // pinned, written by the device (CamBrief above); h_brief_dev = its device address
    CamBrief* h_brief_dev = nullptr;
    CamState* h_cams = nullptr;    // pinned staging for that copy (a pageable destination would make the copy synchronous)
    int ensure_n_points() { return n_points.reserve(sizeof(int) * 4); }
    int init(int dev);
    int ensure_copy_stream();  // vk_abi.hip: the stream (and per-frame events) of the staggered flow upload, created at first use
    int ensure_side_stream();  // the second stream of strict mode's fb_smooth (and its fork / join events), created at first use
    void destroy();
};

Context* default_context();          // lazily created on the current HIP device
Context* pool_context(int idx);      // idx-th extra context of the current device (windows in flight next to each other)
int pool_set_rand_epoch(unsigned epoch);
uint32_t batch_rand_epoch(int dev, const uint32_t* set);  // epoch the windows of the next batch start from
int prof_begin(Context* c);
int prof_end(Context* c, const char* name);
int prof_begin_inner(Context* c);
int prof_end_inner(Context* c, const char* name, int launches);  // accumulates per-launch time

}  // namespace vk
