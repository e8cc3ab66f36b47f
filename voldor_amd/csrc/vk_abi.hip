// voldor_amd/csrc/vk_abi.hip -- the drop-in boundary "B-inner": the six VO entry points of
// gpu-kernels/gpu_kernels.h:11-58 with identical names, C++ linkage, argument order, default
// arguments (in include/gpu_kernels.h), NULL protocol and return convention (0 = success,
// otherwise the runtime error code as int), so voldor/*.cpp links against this library
// unchanged.  An extern "C" facade (include/voldor_hip.h) re-exports them for ctypes / FFI.
#include "vk_common.hpp"
#include "vk_internal.hpp"
#include "../../include/gpu_kernels.h"
#include "../../include/voldor_hip.h"
#include <mutex>
#include <cstdlib>

namespace vk {
void set_strict_math_default(int on);

int Context::init(int dev) {
    device = dev;
    VK_CHECK(hipSetDevice(dev));
    VK_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    VK_CHECK(hipEventCreate(&ev0));
    VK_CHECK(hipEventCreate(&ev1));
    VK_CHECK(hipEventCreate(&ev2));
    VK_CHECK(hipEventCreate(&ev3));
    VK_CHECK(hipEventCreateWithFlags(&ev_cams, hipEventDisableTiming));
    // (copy_stream / side_stream and their events: created when first needed -- ensure_copy_stream / ensure_side_stream.  ROCm maps streams onto a few
    // hardware queues and streams that share one serialise: a context that never uploads from the host and never runs strict mode keeps ONE stream, so
    // four windows in flight stay on four queues)
    VK_CHECK(hipHostMalloc((void**)&h_cams, sizeof(CamState) * MAX_FRAMES, hipHostMallocDefault));
    VK_CHECK(hipHostMalloc((void**)&h_pb, sizeof(PoseBlock), hipHostMallocDefault));
    VK_CHECK(hipHostMalloc((void**)&h_cams_up, sizeof(CamState) * MAX_FRAMES, hipHostMallocDefault));
    VK_CHECK(hipHostMalloc((void**)&h_brief, sizeof(CamBrief) * MAX_FRAMES, hipHostMallocMapped));
    VK_CHECK(hipHostGetDevicePointer((void**)&h_brief_dev, h_brief, 0));
    return 0;
}
int Context::ensure_copy_stream() {
    if (copy_stream) return 0;
    VK_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    for (int f = 0; f < MAX_FRAMES; f++) VK_CHECK(hipEventCreateWithFlags(&ev_frame[f], hipEventDisableTiming));
    return 0;
}
int Context::ensure_side_stream() {
    if (side_stream) return 0;
    VK_CHECK(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
    VK_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    VK_CHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    return 0;
}
void Context::destroy() {
    DevBuf* bufs[] = { &od.flows, &od.rig, &od.rig2, &od.depth, &od.cost, &od.priors, &od.pconfs, &od.confs, &od.pose,
                       &cp.flows, &cp.rig, &cp.depth, &cp.cost, &cp.priors, &cp.pconfs, &cp.confs, &cp.pose,
                       &rig_partial, &local_tbl, &p2_map, &p3_map, &blk_counts, &blk_offsets, &valid_mask, &pts2, &pts3, &n_points,
                       &rvecs, &tvecs, &pool, &ms_io, &cams, &tmp, &fb_scratch, &stale_depth, &sp_coop, &xw_jumps, &xw_px_states, &xw_pose_states };
    for (DevBuf* b : bufs) b->release();
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (ev2) (void)hipEventDestroy(ev2);
    if (ev3) (void)hipEventDestroy(ev3);
    if (ev_cams) (void)hipEventDestroy(ev_cams);
    for (int f = 0; f < MAX_FRAMES; f++) { if (ev_frame[f]) (void)hipEventDestroy(ev_frame[f]); ev_frame[f] = nullptr; }
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    copy_stream = nullptr;
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (side_stream) (void)hipStreamDestroy(side_stream);
    side_stream = nullptr; ev_fork = ev_join = nullptr;
    if (h_cams) (void)hipHostFree(h_cams);
    if (h_brief) (void)hipHostFree(h_brief);
    if (h_pb) (void)hipHostFree(h_pb);
    if (h_cams_up) (void)hipHostFree(h_cams_up);
    h_pb = nullptr; h_cams_up = nullptr;
    h_cams = nullptr; h_brief = h_brief_dev = nullptr;
    if (stream) (void)hipStreamDestroy(stream);
    stream = nullptr; ev0 = ev1 = ev2 = ev3 = ev_cams = nullptr;
    od.pose_init = cp.pose_init = false;
}

static std::mutex g_mu;
static std::map<int, Context*> g_ctx;  // one default context per HIP device

Context* default_context() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        fprintf(stderr, "voldor_hip: no HIP device available (hipGetDevice failed); the HIP path has no CPU fallback\n");
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(dev);
    if (it != g_ctx.end()) return it->second;
    Context* c = new Context();
    if (c->init(dev) != 0) { delete c; return nullptr; }
    g_ctx[dev] = c;
    return c;
}

// extra contexts of a device (own stream, own buffers) for windows in flight next to each other (vk_voldor_device_batch)
static std::map<int, std::vector<Context*>> g_pool_ctx;
Context* pool_context(int idx) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    std::vector<Context*>& v = g_pool_ctx[dev];
    while ((int)v.size() <= idx) {
        Context* c = new Context();
        if (c->init(dev) != 0) { delete c; return nullptr; }
        v.push_back(c);
    }
    return v[(size_t)idx];
}
// depth-sampling epoch that the windows of the next vk_voldor_device_batch start from (per device); set != NULL stores
static std::map<int, uint32_t> g_batch_epoch;
uint32_t batch_rand_epoch(int dev, const uint32_t* set) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (set) g_batch_epoch[dev] = *set;
    return g_batch_epoch[dev];
}
int pool_set_rand_epoch(unsigned epoch) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipErrorNoDevice;
    const uint32_t e = epoch;
    batch_rand_epoch(dev, &e);
    return 0;
}

// strict-math mode of entry points that carry no config string (B-inner) and the default of the window call
static int g_strict_math = -1;  // -1: not set -> environment VOLDOR_HIP_STRICT_MATH
bool strict_math_default() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_strict_math < 0) { const char* e = getenv("VOLDOR_HIP_STRICT_MATH"); g_strict_math = (e && e[0] == '1') ? 1 : 0; }
    return g_strict_math != 0;
}
void set_strict_math_default(int on) { std::lock_guard<std::mutex> lk(g_mu); g_strict_math = on ? 1 : 0; }
static int g_reference_svd = -1;  // -1: not set -> environment VOLDOR_HIP_REFERENCE_SVD
bool reference_svd_default() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_reference_svd < 0) { const char* e = getenv("VOLDOR_HIP_REFERENCE_SVD"); g_reference_svd = (e && e[0] == '1') ? 1 : 0; }
    return g_reference_svd != 0;
}

static int g_reference_rng = -1, g_reference_tex = -1;  // -1: not set -> environment
bool reference_rng_default() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_reference_rng < 0) { const char* e = getenv("VOLDOR_HIP_REFERENCE_RNG"); g_reference_rng = (e && e[0] == '1') ? 1 : 0; }
    return g_reference_rng != 0;
}
bool reference_tex_default() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_reference_tex < 0) { const char* e = getenv("VOLDOR_HIP_REFERENCE_TEX"); g_reference_tex = (e && e[0] == '1') ? 1 : 0; }
    return g_reference_tex != 0;
}

// VOLDOR_HIP_BACKTRACE=1: a C backtrace on SIGABRT / SIGSEGV (diagnostics on boxes without a debugger)
#include <execinfo.h>
#include <signal.h>
static void vk_bt_handler(int sig) {
    void* fr[64];
    const int n = backtrace(fr, 64);
    fprintf(stderr, "voldor_hip: signal %d, backtrace:\n", sig);
    backtrace_symbols_fd(fr, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
static const bool g_bt_installed = [] {
    const char* e = getenv("VOLDOR_HIP_BACKTRACE");
    if (e && e[0] == '1') { signal(SIGABRT, vk_bt_handler); signal(SIGSEGV, vk_bt_handler); }
    return true;
}();

static DebugSwitches g_debug;
// ONE table of the verification switches (vk_debug.h): name, field, accepted values.  vk_debug_switch and the VOLDOR_HIP_DEBUG parser both go
// through debug_switch_set, so a value the launch paths were never tested with cannot reach them from either side.
struct DebugEntry { const char* name; int DebugSwitches::*field; int kind; };  // kind 0: 0 | 1; 1: 0 | 12 | 20 | 40; 2: any value >= 0; 3: 0 | 1 | 2; 4: 0 | 5 | 8
static const DebugEntry g_debug_tab[] = {
    { "local_serial", &DebugSwitches::local_serial, 0 }, { "cost_rand_plain", &DebugSwitches::cost_rand_plain, 0 }, { "fb_segment", &DebugSwitches::fb_segment, 1 },
    { "global_split", &DebugSwitches::global_split, 0 }, { "refit_partition", &DebugSwitches::refit_partition, 0 }, { "split_trials", &DebugSwitches::split_trials, 0 },
    { "strict_plain", &DebugSwitches::strict_plain, 0 }, { "strict_pose_coop", &DebugSwitches::strict_pose_coop, 0 },
    { "strict_coop_max_polls", &DebugSwitches::strict_coop_max_polls, 2 }, { "estep_pairs", &DebugSwitches::estep_pairs, 3 }, { "fb_ride", &DebugSwitches::fb_ride, 0 }, { "defer_reduce", &DebugSwitches::defer_reduce, 0 },
    { "fb_side", &DebugSwitches::fb_side, 0 }, { "bootstrap_default", &DebugSwitches::bootstrap_default, 4 },
    { "strict_filter", &DebugSwitches::strict_filter, 3 }, { "strict_table_filter", &DebugSwitches::strict_table_filter, 3 },
};
// returns the previous value; -1: unknown name; -2: a value the switch does not take
static int debug_switch_set(const char* name, int value) {
    for (const DebugEntry& t : g_debug_tab)
        if (strcmp(t.name, name) == 0) {
            if (t.kind == 0) value = value ? 1 : 0;
            else if (t.kind == 1) { if (value != 0 && value != 12 && value != 20 && value != 40) return -2; }
            else if (t.kind == 3) { if (value < 0 || value > 2) return -2; }
            else if (t.kind == 4) { if (value != 0 && value != 5 && value != 8) return -2; }
            else if (value < 0) return -2;
            const int old = g_debug.*t.field;
            g_debug.*t.field = value;
            return old;
        }
    return -1;
}
DebugSwitches& debug_switches() {
    static const bool from_env = [] {  // VOLDOR_HIP_DEBUG="name=value,name=value": the switches of vk_debug.h for a whole process (A/B runs of the test suite)
        const char* e = getenv("VOLDOR_HIP_DEBUG");
        if (e) {
            std::string s(e);
            size_t pos = 0;
            while (pos < s.size()) {
                const size_t end = s.find(',', pos), eq = s.find('=', pos);
                const size_t stop = end == std::string::npos ? s.size() : end;
                if (eq != std::string::npos && eq < stop) {
                    const std::string k = s.substr(pos, eq - pos), v = s.substr(eq + 1, stop - eq - 1);
                    char* endp = nullptr;
                    const long val = strtol(v.c_str(), &endp, 10);
                    if (v.empty() || (endp && *endp != '\0')) fprintf(stderr, "voldor_hip: VOLDOR_HIP_DEBUG: '%s' is not an integer (switch '%s' left alone)\n", v.c_str(), k.c_str());
                    else {
                        const int r = debug_switch_set(k.c_str(), (int)val);
                        if (r == -1) fprintf(stderr, "voldor_hip: VOLDOR_HIP_DEBUG: unknown switch '%s'\n", k.c_str());
                        else if (r == -2) fprintf(stderr, "voldor_hip: VOLDOR_HIP_DEBUG: switch '%s' does not take the value %ld\n", k.c_str(), val);
                    }
                }
                pos = stop + 1;
            }
        }
        return true;
    }();
    (void)from_env;
    return g_debug;
}

int prof_begin(Context* c) { return (int)hipEventRecord(c->ev0, c->stream); }
int prof_end(Context* c, const char* name) {
    VK_CHECK(hipEventRecord(c->ev1, c->stream));
    VK_CHECK(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    VK_CHECK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    ProfEntry& e = c->prof_acc[name];
    e.ms += ms; e.count += 1;
    return 0;
}

// inner scope (a run of identical launches inside an outer scope); resolved when the outer scope ends
int prof_begin_inner(Context* c) { return (int)hipEventRecord(c->ev2, c->stream); }
int prof_end_inner(Context* c, const char* name, int launches) {
    VK_CHECK(hipEventRecord(c->ev3, c->stream));
    VK_CHECK(hipEventSynchronize(c->ev3));
    float ms = 0.f;
    VK_CHECK(hipEventElapsedTime(&ms, c->ev2, c->ev3));
    ProfEntry& e = c->prof_acc[name];
    e.ms += ms; e.count += launches;
    return 0;
}

// -------- helpers ----------------------------------------------------------------------------
static int upload_K(Context* c, ImageSet& S, const float* h_K) {
    // K4 = fx,cx,fy,cy ; K4inv = 1/fx,-cx/fx,1/fy,-cy/fy  (optimize_depth.cu:343-350)
    float k[8] = { h_K[0], h_K[2], h_K[4], h_K[5], 1.f / h_K[0], -h_K[2] / h_K[0], 1.f / h_K[4], -h_K[5] / h_K[4] };
    VK_CHECK(hipMemcpyAsync(S.pb()->K4, k, sizeof k, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}
static int upload_rows(Context* c, float* dst, float* const* src, int n, size_t row_floats) {
    float tmp[MAX_FRAMES * 9];
    for (int f = 0; f < n; f++) memcpy(tmp + (size_t)f * row_floats, src[f], row_floats * sizeof(float));
    VK_CHECK(hipMemcpyAsync(dst, tmp, sizeof(float) * row_floats * n, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}
static int upload_layers(Context* c, DevBuf& buf, float* const* src, int n, size_t layer_floats) {
    for (int f = 0; f < n; f++)
        VK_CHECK(hipMemcpyAsync(buf.as<float>() + (size_t)f * layer_floats, src[f], sizeof(float) * layer_floats,
                                hipMemcpyHostToDevice, c->stream));
    return 0;
}
static int download_layers(Context* c, const DevBuf& buf, float* const* dst, int n, size_t layer_floats) {
    for (int f = 0; f < n; f++)
        VK_CHECK(hipMemcpyAsync(dst[f], buf.as<float>() + (size_t)f * layer_floats, sizeof(float) * layer_floats,
                                hipMemcpyDeviceToHost, c->stream));
    return 0;
}

}  // namespace vk

using namespace vk;

// ============================================================================================
// gpu_kernels.h:44-58 / optimize_depth.cu:293-520
int optimize_depth_gpu(float* h_flows[], float* h_rigidnesses[], float* h_o_rigidnesses[], float* h_depth_priors[],
                       float* h_depth_prior_pconfs[], float* h_depth_prior_confs[], float* h_o_depth_prior_confs[],
                       float* h_depth, float* h_o_depth, float* h_K, float* h_Rs[], float* h_ts[], float* h_dp_Rs[],
                       float* h_dp_ts[], float abs_resize_factor, int N, int N_dp, int w, int h, float basefocal,
                       int n_rand_samples, int global_prop_step, int local_prop_width, float lambda, float omega,
                       float disp_delta, float delta, bool fb_smooth, float s0_ems_prob, float no_change_prob,
                       float range_factor, bool update_rigidness_only) {
    Context* c = default_context();
    if (!c) return (int)hipErrorNoDevice;
    if (N > MAX_FRAMES || N_dp > MAX_DISP_FRAMES || N < 0 || N_dp < 0 || w <= 0 || h <= 0) return (int)hipErrorInvalidValue;
    ImageSet& S = c->od;
    const size_t npx = (size_t)w * h;
    S.w = w; S.h = h;
    if (int e = S.ensure_pose()) return e;
    if (h_K) { if (int e = upload_K(c, S, h_K)) return e; }
    if (int e = S.depth.reserve(sizeof(float) * npx)) return e;
    if (h_depth) VK_CHECK(hipMemcpyAsync(S.depth.p, h_depth, sizeof(float) * npx, hipMemcpyHostToDevice, c->stream));
    if (N > 0) {
        if (h_Rs) { if (int e = upload_rows(c, &S.pb()->Rs[0][0], h_Rs, N, 9)) return e; }
        if (h_ts) { if (int e = upload_rows(c, &S.pb()->ts[0][0], h_ts, N, 3)) return e; }
        if (int e = S.flows.reserve(sizeof(float) * 2 * npx * N)) return e;
        if (h_flows) { if (int e = upload_layers(c, S.flows, h_flows, N, npx * 2)) return e; }
        if (int e = S.rig.reserve(sizeof(float) * npx * N)) return e;
        if (h_rigidnesses) { if (int e = upload_layers(c, S.rig, h_rigidnesses, N, npx)) return e; }
    }
    if (N_dp > 0) {
        if (h_dp_Rs) { if (int e = upload_rows(c, &S.pb()->dpRs[0][0], h_dp_Rs, N_dp, 9)) return e; }
        if (h_dp_ts) { if (int e = upload_rows(c, &S.pb()->dpts[0][0], h_dp_ts, N_dp, 3)) return e; }
        if (int e = S.priors.reserve(sizeof(float) * npx * N_dp)) return e;
        if (h_depth_priors) { if (int e = upload_layers(c, S.priors, h_depth_priors, N_dp, npx)) return e; }
        if (int e = S.pconfs.reserve(sizeof(float) * npx * N_dp)) return e;
        if (h_depth_prior_pconfs) { if (int e = upload_layers(c, S.pconfs, h_depth_prior_pconfs, N_dp, npx)) return e; }
        if (int e = S.confs.reserve(sizeof(float) * npx * N_dp)) return e;
        if (h_depth_prior_confs) { if (int e = upload_layers(c, S.confs, h_depth_prior_confs, N_dp, npx)) return e; }
    }
    OdParams p;
    p.abs_resize_factor = abs_resize_factor; p.N = N; p.N_dp = N_dp; p.w = w; p.h = h; p.basefocal = basefocal;
    p.n_rand_samples = n_rand_samples; p.global_prop_step = global_prop_step; p.local_prop_width = local_prop_width;
    p.lambda = lambda; p.omega = omega; p.disp_delta = disp_delta; p.delta = delta; p.fb_smooth = fb_smooth;
    p.s0_ems_prob = s0_ems_prob; p.no_change_prob = no_change_prob; p.range_factor = range_factor;
    p.update_rigidness_only = update_rigidness_only;
    p.strict = strict_math_default();
    p.ref_rng = p.strict && reference_rng_default(); p.ref_tex = p.strict && reference_tex_default();
    // the frame count comes by argument here: undo whatever a window call (py_voldor_wrapper: device-side truncation,
    // PoseBlock::n_active) left in the shared pose block
    const int all_frames = MAX_FRAMES;
    VK_CHECK(hipMemcpyAsync(&S.pb()->n_active, &all_frames, sizeof(int), hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    if (int e = optimize_depth_device(c, S, p)) return e;
    if (h_o_depth) VK_CHECK(hipMemcpyAsync(h_o_depth, S.depth.p, sizeof(float) * npx, hipMemcpyDeviceToHost, c->stream));
    if (h_o_rigidnesses && N > 0) { if (int e = download_layers(c, S.rig, h_o_rigidnesses, N, npx)) return e; }
    if (h_o_depth_prior_confs && N_dp > 0) { if (int e = download_layers(c, S.confs, h_o_depth_prior_confs, N_dp, npx)) return e; }
    VK_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// gpu_kernels.h:24-35 / collect_p3p_instances.cu:147-250
int collect_p3p_instances(float* h_flows[], float* h_rigidnesses[], float* h_depth, float* h_K, float* h_Rs[], float* h_ts[],
                          float* h_o_p2_map, float* h_o_p3_map, int N, int w, int h, int active_idx, float rigidness_thresh,
                          float rigidness_sum_thresh, float sample_min_depth, float sample_max_depth, int max_trace_on_flow) {
    Context* c = default_context();
    if (!c) return (int)hipErrorNoDevice;
    if (N > MAX_FRAMES || N <= 0 || w <= 0 || h <= 0 || active_idx < 0 || active_idx >= N) return (int)hipErrorInvalidValue;
    ImageSet& S = c->cp;
    const size_t npx = (size_t)w * h;
    S.w = w; S.h = h;
    if (int e = S.ensure_pose()) return e;
    if (h_K) { if (int e = upload_K(c, S, h_K)) return e; }
    if (h_Rs) { if (int e = upload_rows(c, &S.pb()->Rs[0][0], h_Rs, N, 9)) return e; }
    if (h_ts) { if (int e = upload_rows(c, &S.pb()->ts[0][0], h_ts, N, 3)) return e; }
    if (int e = S.flows.reserve(sizeof(float) * 2 * npx * N)) return e;
    if (h_flows) { if (int e = upload_layers(c, S.flows, h_flows, N, npx * 2)) return e; }
    if (int e = S.rig.reserve(sizeof(float) * npx * N)) return e;
    if (h_rigidnesses) { if (int e = upload_layers(c, S.rig, h_rigidnesses, N, npx)) return e; }
    if (int e = S.depth.reserve(sizeof(float) * npx)) return e;
    if (h_depth) VK_CHECK(hipMemcpyAsync(S.depth.p, h_depth, sizeof(float) * npx, hipMemcpyHostToDevice, c->stream));
    if (int e = collect_device(c, S, N, w, h, active_idx, rigidness_thresh, rigidness_sum_thresh, sample_min_depth,
                               sample_max_depth, max_trace_on_flow, nullptr, true, false, strict_math_default() && reference_tex_default()))  // the texture filter is a strict-mode switch (voldor_hip.h)
        return e;
    if (h_o_p2_map) VK_CHECK(hipMemcpyAsync(h_o_p2_map, c->p2_map.p, sizeof(float) * 2 * npx, hipMemcpyDeviceToHost, c->stream));
    if (h_o_p3_map) VK_CHECK(hipMemcpyAsync(h_o_p3_map, c->p3_map.p, sizeof(float) * 3 * npx, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

static int solve_batch_host(float* h_p3s, float* h_p2s, float* h_o_rvecs, float* h_o_tvecs, float* h_K, int N_pts, int N_poses,
                            int solver) {
    Context* c = default_context();
    if (!c) return (int)hipErrorNoDevice;
    if (N_pts <= 0 || N_poses <= 0 || !h_K) return (int)hipErrorInvalidValue;
    if (int e = c->pts2.reserve(sizeof(float) * 2 * (size_t)N_pts)) return e;
    if (int e = c->pts3.reserve(sizeof(float) * 3 * (size_t)N_pts)) return e;
    if (int e = c->ensure_n_points()) return e;
    VK_CHECK(hipMemcpyAsync(c->pts2.p, h_p2s, sizeof(float) * 2 * (size_t)N_pts, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipMemcpyAsync(c->pts3.p, h_p3s, sizeof(float) * 3 * (size_t)N_pts, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipMemcpyAsync(c->n_points.p, &N_pts, sizeof(int), hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    if (int e = solve_device(c, c->pts2.as<float>(), c->pts3.as<float>(), c->n_points.as<int>(), h_K[0], h_K[4], h_K[2], h_K[5],
                             N_poses, solver, strict_math_default(), nullptr, reference_svd_default(), reference_rng_default()))
        return e;
    VK_CHECK(hipMemcpyAsync(h_o_rvecs, c->rvecs.p, sizeof(float) * 3 * (size_t)N_poses, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipMemcpyAsync(h_o_tvecs, c->tvecs.p, sizeof(float) * 3 * (size_t)N_poses, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}
// gpu_kernels.h:37-42
int solve_batch_p3p_ap3p_gpu(float* h_p3s, float* h_p2s, float* h_o_rvecs, float* h_o_tvecs, float* h_K, int N_pts, int N_poses) {
    return solve_batch_host(h_p3s, h_p2s, h_o_rvecs, h_o_tvecs, h_K, N_pts, N_poses, 1);
}
int solve_batch_p3p_lambdatwist_gpu(float* h_p3s, float* h_p2s, float* h_o_rvecs, float* h_o_tvecs, float* h_K, int N_pts,
                                    int N_poses) {
    return solve_batch_host(h_p3s, h_p2s, h_o_rvecs, h_o_tvecs, h_K, N_pts, N_poses, 0);
}

// gpu_kernels.h:11-15 / meanshift.cu:34-150
int meanshift_gpu(float* h_space, float kernel_var, float* h_io_mean, float* h_o_confidence, int* used_iters,
                  bool use_external_init_mean, int N, int dims, float epsilon, int max_iters, int max_init_trials,
                  float good_init_confidence) {
    Context* c = default_context();
    if (!c) return (int)hipErrorNoDevice;
    if (dims > MAX_POSE_DIMS || dims <= 0 || N <= 0) return (int)hipErrorInvalidValue;
    if (int e = c->pool.reserve(sizeof(float) * (size_t)N * dims)) return e;
    if (int e = c->ms_io.reserve(sizeof(float) * 64 + sizeof(int) * 4)) return e;
    float io[64] = { 0 };
    memcpy(io, h_io_mean, sizeof(float) * dims);
    VK_CHECK(hipMemcpyAsync(c->pool.p, h_space, sizeof(float) * (size_t)N * dims, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipMemcpyAsync(c->ms_io.p, io, sizeof io, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    ModeParams mp{};
    mp.dims = dims; mp.kernel_var = kernel_var; mp.ms_epsilon = epsilon; mp.ms_max_iters = max_iters;
    mp.ms_max_init_trials = max_init_trials; mp.ms_good_init_confidence = good_init_confidence;
    mp.use_external_init_mean = use_external_init_mean ? 1 : 0;
    if (used_iters) *used_iters = 0;
    int* ioi = reinterpret_cast<int*>(c->ms_io.as<float>() + 64);
    if (strict_math_default() && N <= 32768) { if (int e = meanshift_strict_device(c, c->pool.as<float>(), N, mp, c->ms_io.as<float>(), ioi)) return e; }
    else if (int e = meanshift_device(c, c->pool.as<float>(), N, mp, c->ms_io.as<float>(), ioi)) return e;
    int hi[2] = { 0, 0 };
    VK_CHECK(hipMemcpyAsync(io, c->ms_io.p, sizeof io, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipMemcpyAsync(hi, ioi, sizeof hi, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    memcpy(h_io_mean, io, sizeof(float) * dims);
    if (max_iters > 0) {
        if (h_o_confidence) *h_o_confidence = io[16];
        if (used_iters) *used_iters = hi[0];
    }
    return 0;
}

// The ONE entry of the verification switches (vk_debug.h; not in include/voldor_hip.h): returns the previous value, -1 for an unknown
// name or a value the switch does not take.
// counters of the verification interface (vk_debug.h): read and cleared; -1 for an unknown name or a device error
extern "C" __attribute__((visibility("default"))) int vk_debug_counter(const char* name) {
    if (!name) return -1;
    vk::Context* c = vk::default_context();
    if (!c) return -1;
    if (strcmp(name, "strict_coop_fallbacks") == 0) return vk::strict_coop_fallbacks(c);
    if (strcmp(name, "fb_blocks_rode") == 0) { const long v = c->dbg_fb_blocks_rode; c->dbg_fb_blocks_rode = 0; return (int)v; }
    if (strcmp(name, "fb_side_passes") == 0) { const long v = c->dbg_fb_side_passes; c->dbg_fb_side_passes = 0; return (int)v; }
    if (strcmp(name, "sf_samples") == 0) return vk::strict_filter_stat(c, 0);
    if (strcmp(name, "sf_sample_survivors") == 0) return vk::strict_filter_stat(c, 1);
    if (strcmp(name, "sf_table_tiles") == 0) return vk::strict_filter_stat(c, 2);
    if (strcmp(name, "sf_table_queued") == 0) return vk::strict_filter_stat(c, 3);
    if (strcmp(name, "reduces_rode") == 0) { const long v = c->dbg_reduces_rode; c->dbg_reduces_rode = 0; return (int)v; }
    return -1;
}
extern "C" __attribute__((visibility("default"))) int vk_debug_switch(const char* name, int value) {
    if (!name) return -1;
    (void)vk::debug_switches();  // the environment is applied first, once
    const int r = vk::debug_switch_set(name, value);
    return r < 0 ? -1 : r;
}

// Verification entry (vk_debug.h): the mode kernel of the WINDOW PIPELINE -- k_pose_mode<REFIT,512>, the packed-pair mean shift
// and, with do_rg, the whitened / distance-partitioned refit on the same registers -- run on a caller-supplied pool of hypotheses, so
// that the kernels the timed path uses can be held against the oracle stage by stage (the host-pointer meanshift_gpu /
// fit_robust_gaussian above are different, generic-dimension kernels).  h_rvecs / h_tvecs: [n_poses][3] (non-finite = dropped, as
// k_solve marks them); io_pose6: rvec, t of the starting pose in, result out.  Returns nonzero on a launch error; *o_success is the
// camera record's success flag.
extern "C" __attribute__((visibility("default"))) int vk_pose_mode_pool(const float* h_rvecs, const float* h_tvecs, int n_poses, int use_external_init_mean,
        float* io_pose6, float kernel_var, float rvec_scale, float ms_epsilon, int ms_max_iters, int ms_max_init_trials, float ms_good_init_confidence,
        int do_rg, float rg_trunc_sigma, float rg_covar_reg_lambda, float rg_epsilon, int rg_max_iters, float rg_pose_scaling,
        float* o_covar36, float* o_density, int* o_sample_count, int* o_ms_iters, int* o_gu_iters, int* o_success) {
    using namespace vk;
    Context* c = default_context();
    if (!c) return (int)hipErrorNoDevice;
    if (n_poses <= 0 || !h_rvecs || !h_tvecs || !io_pose6) return (int)hipErrorInvalidValue;
    if (int e = c->rvecs.reserve(sizeof(float) * 3 * (size_t)n_poses)) return e;
    if (int e = c->tvecs.reserve(sizeof(float) * 3 * (size_t)n_poses)) return e;
    if (int e = c->cams.reserve(sizeof(CamState) * MAX_FRAMES)) return e;
    if (int e = c->ensure_n_points()) return e;
    if (int e = c->od.ensure_pose()) return e;
    std::vector<float> planes((size_t)6 * n_poses);  // coordinate planes [3][n_poses], the layout k_solve writes for the mode kernels
    for (int i = 0; i < n_poses; i++)
        for (int d = 0; d < 3; d++) { planes[(size_t)d * n_poses + i] = h_rvecs[(size_t)i * 3 + d]; planes[(size_t)(3 + d) * n_poses + i] = h_tvecs[(size_t)i * 3 + d]; }
    CamState cam{};
    for (int d = 0; d < 3; d++) { cam.rvec[d] = io_pose6[d]; cam.t[d] = io_pose6[3 + d]; }
    const int enough = 4;
    VK_CHECK(hipMemcpyAsync(c->rvecs.p, planes.data(), sizeof(float) * 3 * (size_t)n_poses, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipMemcpyAsync(c->tvecs.p, planes.data() + (size_t)3 * n_poses, sizeof(float) * 3 * (size_t)n_poses, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipMemcpyAsync(c->cams.p, &cam, sizeof cam, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipMemcpyAsync(c->n_points.p, &enough, sizeof enough, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    ModeParams mp{};
    mp.dims = 6; mp.kernel_var = kernel_var; mp.ms_epsilon = ms_epsilon; mp.ms_max_iters = ms_max_iters; mp.ms_max_init_trials = ms_max_init_trials;
    mp.ms_good_init_confidence = ms_good_init_confidence; mp.use_external_init_mean = use_external_init_mean ? 1 : 0;
    mp.rvec_scale = rvec_scale; mp.rg_pose_scaling = rg_pose_scaling; mp.do_rg = do_rg ? 1 : 0; mp.rg_trunc_sigma = rg_trunc_sigma;
    mp.rg_covar_reg_lambda = rg_covar_reg_lambda; mp.rg_epsilon = rg_epsilon; mp.rg_max_iters = rg_max_iters;
    if (strict_math_default()) {  // the strict mode kernel of the window pipeline (k_pose_strict_par, or k_pose_strict with the "strict_plain" switch)
        if (int e = pose_mode_strict_device(c, n_poses, mp, c->cams.as<CamState>(), c->od.pb(), 0)) return e;
    } else if (int e = pose_mode_device(c, n_poses, mp, c->cams.as<CamState>(), c->od.pb(), 0, !use_external_init_mean)) return e;
    VK_CHECK(hipMemcpyAsync(&cam, c->cams.p, sizeof cam, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    for (int d = 0; d < 3; d++) { io_pose6[d] = cam.rvec[d]; io_pose6[3 + d] = cam.t[d]; }
    if (o_covar36) memcpy(o_covar36, cam.covar, sizeof cam.covar);
    if (o_density) *o_density = cam.pose_density;
    if (o_sample_count) *o_sample_count = cam.pose_sample_count;
    if (o_ms_iters) *o_ms_iters = cam.last_used_ms_iters;
    if (o_gu_iters) *o_gu_iters = cam.last_used_gu_iters;
    if (o_success) *o_success = cam.success;
    return 0;
}

// gpu_kernels.h:17-22 / fit_robust_gaussian.cu:101-286. Returns 0 iff the fit is reliable.
int fit_robust_gaussian(float* h_space, float* h_io_mean, float* h_io_covar, float trunc_sigma, float covar_reg_lambda,
                        float* h_o_density, int* used_iters, int N, int dims, float epsilon, int max_iters) {
    // The reference does a bare `throw` for dims>6 (fit_robust_gaussian.cu:107-108), i.e.
    // std::terminate; a library should not kill its host process, so report an error instead.
    if (dims > 6 || dims <= 0 || N <= 0) return (int)hipErrorInvalidValue;
    Context* c = default_context();
    if (!c) return (int)hipErrorNoDevice;
    if (int e = c->pool.reserve(sizeof(float) * (size_t)N * dims)) return e;
    if (int e = c->ms_io.reserve(sizeof(float) * 64 + sizeof(int) * 4)) return e;
    float io[64] = { 0 };
    memcpy(io, h_io_mean, sizeof(float) * dims);
    memcpy(io + 6, h_io_covar, sizeof(float) * dims * dims);
    VK_CHECK(hipMemcpyAsync(c->pool.p, h_space, sizeof(float) * (size_t)N * dims, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipMemcpyAsync(c->ms_io.p, io, sizeof io, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    ModeParams mp{};
    mp.dims = dims; mp.rg_trunc_sigma = trunc_sigma; mp.rg_covar_reg_lambda = covar_reg_lambda; mp.rg_epsilon = epsilon;
    mp.rg_max_iters = max_iters;
    if (used_iters) *used_iters = 0;
    int* ioi = reinterpret_cast<int*>(c->ms_io.as<float>() + 64);
    if (strict_math_default() && N <= 32768) { if (int e = robust_gaussian_strict_device(c, c->pool.as<float>(), N, mp, c->ms_io.as<float>(), ioi)) return e; }
    else if (int e = robust_gaussian_device(c, c->pool.as<float>(), N, mp, c->ms_io.as<float>(), ioi)) return e;
    int hi[2] = { 0, 0 };
    VK_CHECK(hipMemcpyAsync(io, c->ms_io.p, sizeof io, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipMemcpyAsync(hi, ioi, sizeof hi, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    if (hi[1] == 0) {
        if (h_o_density) *h_o_density = io[42];
        if (used_iters) *used_iters = hi[0];
        memcpy(h_io_mean, io, sizeof(float) * dims);
        memcpy(h_io_covar, io + 6, sizeof(float) * dims * dims);
        return 0;
    }
    return 1;  // !cudaSuccess (fit_robust_gaussian.cu:282-285)
}

// gblur_gpu (gpu-kernels/gblur.cu:47-72) takes GMatf objects in the reference; host-pointer form here.
static int gblur_host(const float* h_src, float* h_dst, int w, int h, int d, float sigma, int ksize) {
    Context* c = default_context();
    if (!c) return (int)hipErrorNoDevice;
    const size_t n = (size_t)w * h * d;
    if (int e = c->tmp.reserve(sizeof(float) * (3 * n + 128))) return e;
    float* src = c->tmp.as<float>(); float* dst = src + n; float* tmp = dst + n; float* gk = tmp + n;
    VK_CHECK(hipMemcpyAsync(src, h_src, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    if (int e = gblur_device(c, src, dst, tmp, gk, w, h, d, sigma, ksize)) return e;
    VK_CHECK(hipMemcpyAsync(h_dst, dst, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// ============================================================================================
// extern "C" facade (include/voldor_hip.h)
extern "C" {

int vk_meanshift_gpu(float* h_space, float kernel_var, float* h_io_mean, float* h_o_confidence, int* used_iters,
                     int use_external_init_mean, int N, int dims, float epsilon, int max_iters, int max_init_trials,
                     float good_init_confidence) {
    return meanshift_gpu(h_space, kernel_var, h_io_mean, h_o_confidence, used_iters, use_external_init_mean != 0, N, dims, epsilon,
                         max_iters, max_init_trials, good_init_confidence);
}
int vk_fit_robust_gaussian(float* h_space, float* h_io_mean, float* h_io_covar, float trunc_sigma, float covar_reg_lambda,
                           float* h_o_density, int* used_iters, int N, int dims, float epsilon, int max_iters) {
    return fit_robust_gaussian(h_space, h_io_mean, h_io_covar, trunc_sigma, covar_reg_lambda, h_o_density, used_iters, N, dims,
                               epsilon, max_iters);
}
int vk_collect_p3p_instances(float** h_flows, float** h_rigidnesses, float* h_depth, float* h_K, float** h_Rs, float** h_ts,
                             float* h_o_p2_map, float* h_o_p3_map, int N, int w, int h, int active_idx, float rigidness_thresh,
                             float rigidness_sum_thresh, float sample_min_depth, float sample_max_depth, int max_trace_on_flow) {
    return collect_p3p_instances(h_flows, h_rigidnesses, h_depth, h_K, h_Rs, h_ts, h_o_p2_map, h_o_p3_map, N, w, h, active_idx,
                                 rigidness_thresh, rigidness_sum_thresh, sample_min_depth, sample_max_depth, max_trace_on_flow);
}
int vk_solve_batch_p3p_ap3p_gpu(float* h_p3s, float* h_p2s, float* h_o_rvecs, float* h_o_tvecs, float* h_K, int N_pts, int N_poses) {
    return solve_batch_p3p_ap3p_gpu(h_p3s, h_p2s, h_o_rvecs, h_o_tvecs, h_K, N_pts, N_poses);
}
int vk_solve_batch_p3p_lambdatwist_gpu(float* h_p3s, float* h_p2s, float* h_o_rvecs, float* h_o_tvecs, float* h_K, int N_pts,
                                       int N_poses) {
    return solve_batch_p3p_lambdatwist_gpu(h_p3s, h_p2s, h_o_rvecs, h_o_tvecs, h_K, N_pts, N_poses);
}
int vk_solve_batch_p3p_lambdatwist_f64_gpu(float* h_p3s, float* h_p2s, float* h_o_rvecs, float* h_o_tvecs, float* h_K, int N_pts,
                                           int N_poses) {
    return solve_batch_host(h_p3s, h_p2s, h_o_rvecs, h_o_tvecs, h_K, N_pts, N_poses, 2);
}
int vk_optimize_depth_gpu(float** h_flows, float** h_rigidnesses, float** h_o_rigidnesses, float** h_depth_priors,
                          float** h_depth_prior_pconfs, float** h_depth_prior_confs, float** h_o_depth_prior_confs, float* h_depth,
                          float* h_o_depth, float* h_K, float** h_Rs, float** h_ts, float** h_dp_Rs, float** h_dp_ts,
                          float abs_resize_factor, int N, int N_dp, int w, int h, float basefocal, int n_rand_samples,
                          int global_prop_step, int local_prop_width, float lambda, float omega, float disp_delta, float delta,
                          int fb_smooth, float s0_ems_prob, float no_change_prob, float range_factor, int update_rigidness_only) {
    return optimize_depth_gpu(h_flows, h_rigidnesses, h_o_rigidnesses, h_depth_priors, h_depth_prior_pconfs, h_depth_prior_confs,
                              h_o_depth_prior_confs, h_depth, h_o_depth, h_K, h_Rs, h_ts, h_dp_Rs, h_dp_ts, abs_resize_factor, N,
                              N_dp, w, h, basefocal, n_rand_samples, global_prop_step, local_prop_width, lambda, omega, disp_delta,
                              delta, fb_smooth != 0, s0_ems_prob, no_change_prob, range_factor, update_rigidness_only != 0);
}
int vk_gblur(const float* h_src, float* h_dst, int w, int h, int d, float sigma, int ksize) {
    return gblur_host(h_src, h_dst, w, h, d, sigma, ksize);
}
// fb_smooth (gpu-kernels/fb_smooth.h:72-108) alone, in place on host maps [n_maps][h][w]
int vk_fb_smooth(float* h_maps, int n_maps, int w, int h, float s0_ems_prob, float no_change_prob) {
    Context* c = default_context();
    if (!c) return (int)hipErrorNoDevice;
    const size_t bytes = sizeof(float) * (size_t)n_maps * w * h;
    if (int e = c->tmp.reserve(bytes)) return e;
    VK_CHECK(hipMemcpyAsync(c->tmp.p, h_maps, bytes, hipMemcpyHostToDevice, c->stream));
    if (strict_math_default()) { if (int e = fb_smooth_strict_device(c, c->tmp.as<float>(), n_maps, w, h, s0_ems_prob, no_change_prob)) return e; }
    else if (int e = fb_smooth_device(c, c->tmp.as<float>(), n_maps, w, h, s0_ems_prob, no_change_prob)) return e;
    VK_CHECK(hipMemcpyAsync(h_maps, c->tmp.p, bytes, hipMemcpyDeviceToHost, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}
// Host copy of the compacted correspondences produced by the last collect call (what the host
// loop of voldor/geometry.cpp:68-80 builds); returns n_points or a negative error.
int vk_get_compacted_points(float* h_o_pts2, float* h_o_pts3, int max_points) {
    Context* c = default_context();
    if (!c || !c->n_points.p) return -1;
    int n = 0;
    if (hipMemcpy(&n, c->n_points.p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -2;
    int m = n < max_points ? n : max_points;
    if (m > 0) {
        if (h_o_pts2 && hipMemcpy(h_o_pts2, c->pts2.p, sizeof(float) * 2 * (size_t)m, hipMemcpyDeviceToHost) != hipSuccess) return -3;
        if (h_o_pts3 && hipMemcpy(h_o_pts3, c->pts3.p, sizeof(float) * 3 * (size_t)m, hipMemcpyDeviceToHost) != hipSuccess) return -3;
    }
    return n;
}
// ---- Middlebury .flo at the boundary (voldor/utils.cpp:23-41 load_flow, slam_py/flow_utils.py:10-34) --------------
// float32 magic 202021.25, int32 width, int32 height, float32 [h][w][2].  vk_read_flo: out == NULL only reports the size;
// returns 0, or 1 cannot open, 2 bad magic, 3 buffer too small (cap_floats), 4 truncated file.
int vk_read_flo(const char* path, int* w, int* h, float* out, size_t cap_floats) {
    FILE* fs = fopen(path, "rb");
    if (!fs) return 1;
    float magic = 0.f; int ww = 0, hh = 0;
    const bool head = fread(&magic, sizeof(float), 1, fs) == 1 && fread(&ww, sizeof(int), 1, fs) == 1 && fread(&hh, sizeof(int), 1, fs) == 1;
    if (!head || magic != 202021.25f || ww <= 0 || hh <= 0) { fclose(fs); return head ? 2 : 4; }
    if (w) *w = ww;
    if (h) *h = hh;
    int rc = 0;
    if (out) {
        const size_t n = (size_t)ww * hh * 2;
        if (cap_floats < n) rc = 3;
        else if (fread(out, sizeof(float), n, fs) != n) rc = 4;
    }
    fclose(fs);
    return rc;
}
int vk_write_flo(const char* path, const float* flow, int w, int h) {
    if (!flow || w <= 0 || h <= 0) return 2;
    FILE* fs = fopen(path, "wb");
    if (!fs) return 1;
    const float magic = 202021.25f;
    const size_t n = (size_t)w * h * 2;
    const bool ok = fwrite(&magic, sizeof(float), 1, fs) == 1 && fwrite(&w, sizeof(int), 1, fs) == 1 && fwrite(&h, sizeof(int), 1, fs) == 1 &&
                    fwrite(flow, sizeof(float), n, fs) == n;
    fclose(fs);
    return ok ? 0 : 4;
}
int vk_set_rand_epoch(unsigned epoch) {
    Context* c = default_context();
    if (!c) return (int)hipErrorNoDevice;
    c->rand_epoch = epoch;
    c->rand_w = c->rand_h = -1;  // explicit seed: the next call adopts its size without resetting
    return pool_set_rand_epoch(epoch);  // and the contexts of vk_voldor_device_batch
}
int vk_set_strict_math(int on) { set_strict_math_default(on); return 0; }
int vk_get_strict_math(void) { return strict_math_default() ? 1 : 0; }
int vk_set_reference_svd(int on) { std::lock_guard<std::mutex> lk(g_mu); g_reference_svd = on ? 1 : 0; return 0; }
int vk_set_reference_rng(int on) { std::lock_guard<std::mutex> lk(g_mu); g_reference_rng = on ? 1 : 0; return 0; }
int vk_get_reference_rng(void) { return reference_rng_default() ? 1 : 0; }
int vk_set_reference_tex(int on) { std::lock_guard<std::mutex> lk(g_mu); g_reference_tex = on ? 1 : 0; return 0; }
int vk_get_reference_tex(void) { return reference_tex_default() ? 1 : 0; }
int vk_get_reference_svd(void) { return reference_svd_default() ? 1 : 0; }
unsigned vk_get_rand_epoch(void) {
    Context* c = default_context();
    return c ? c->rand_epoch : 0u;
}
int vk_profile_enable(int on) {
    Context* c = default_context();
    if (!c) return (int)hipErrorNoDevice;
    c->prof = on != 0;
    if (on) c->prof_acc.clear();
    return 0;
}
int vk_profile_get(const char* name, double* total_ms, long* count) {
    Context* c = default_context();
    if (!c) return (int)hipErrorNoDevice;
    auto it = c->prof_acc.find(name);
    if (it == c->prof_acc.end()) { *total_ms = 0; *count = 0; return 1; }
    *total_ms = it->second.ms; *count = it->second.count;
    return 0;
}
int vk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int vk_set_device(int dev) { return (int)hipSetDevice(dev); }
const char* vk_version(void) { return "voldor_hip 0.1 (gfx950)"; }

}  // extern "C"
