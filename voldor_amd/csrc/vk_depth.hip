// voldor_amd/csrc/vk_depth.hip -- depth / rigidness half of the EM loop on gfx950.
// Replaces gpu-kernels/optimize_depth.cu:84-291 (kernels) and :462-494 (stage order) and
// gpu-kernels/fb_smooth.h:17-109.
//
// Launch structure per optimize_depth call (reference: 20+ launches, 10 of them streaming a
// 48-byte RNG state per pixel):
//   fb_rows / fb_cols           2 launches per map set (reference: 6): one lane per 20-step (40 on very large images) line segment,
//                               segment maps composed as 2x2 projective matrices, chained through LDS
//   cost_rand                   1 launch: cost map + all n_rand samples, depth/cost in regs
//   global_prop x4              1 launch each, one thread per candidate site (step>=2)
//   local_table + local_runs x4 2 launches per pass: per-pixel candidate-cost table, then one wave per chain
//                               resolving the serial chain run by run
//   update_rigidness            1 launch (+ per-block rigidness sums for the density test)
// Every per-pixel kernel clamps its frame count to PoseBlock::n_active (device-side truncation decision) and
// remaps its workgroup id so that an XCD works on one band of the image (xcd_band_tile).
#include "vk_common.hpp"
#include "vk_device.hpp"
#include "vk_strict_model.hpp"
#include "vk_ref_cuda.h"
#include "vk_internal.hpp"
#include "vk_cum_poses.hpp"
#include "vk_fb.hpp"
#include <cstdlib>

namespace vk {

// the kernels templated on the frame bound, and optimize_depth_launch<NMAX, STRICT>, are compiled in vk_depth_i*.hip / vk_depth_s*.hip (vk_depth_impl.hpp)
template <int NMAX, bool STRICT> int optimize_depth_launch(Context* c, ImageSet& S, const OdParams& p, bool cost_only);

__global__ __launch_bounds__(64) static void k_cum_poses(PoseBlock* P, int N, int N_dp, float* world_scale) { cum_poses_block(P, N, N_dp, world_scale); }

// fixed-order second stage: cams[f].pose_rigidness_density = sum(partial[f][:]) / npx
// blocks 0..n_launch-1: rigidness density of one frame; block n_launch (only launched when scale_out != NULL): the pose half
// of normalize_world_scale (voldor.cpp:309-317), scale = n / sum ||t_i|| over the registered frames -- one launch for both
__global__ __launch_bounds__(256) static void k_reduce_density(ReduceArgs a) { reduce_density_block(blockIdx.x, a); }

template <bool VEC4, int FB_SEG>
__global__ __launch_bounds__(256) static void k_fb_rows(const float* maps, float* maps_out /* == maps: in place */, int w, int h, int S, float e0, float p, const int* __restrict__ n_dev,
                                                        int n_maps, PoseBlock* cumP, int cumN, int cumNdp, float* world_scale) {
    __shared__ FbMat sF[2][256], sB[2][256];
    if ((int)blockIdx.y == n_maps) {  // extra layer (launched only with cumP): one workgroup prepares the projective maps the cost kernels need next
        if (blockIdx.x == 0) cum_poses_block(cumP, cumN, cumNdp, world_scale);
        return;
    }
    if (n_dev && (int)blockIdx.y >= *n_dev) return;  // map of a frame the device-side decision has dropped
    fb_rows_body<VEC4, FB_SEG>(maps, maps_out, w, h, S, e0, p, blockIdx.x, blockIdx.y, &sF[0][0], &sB[0][0], threadIdx.x);
}
template <int FB_SEG>
__global__ __launch_bounds__(1024) static void k_fb_cols(float* __restrict__ maps, int w, int h, int S, float e0, float p, const int* __restrict__ n_dev) {
    __shared__ FbMat sF[1024], sB[1024];
    if (n_dev && (int)blockIdx.y >= *n_dev) return;
    fb_cols_body<FB_SEG>(maps, w, h, S, FB_CW, e0, p, blockIdx.x, blockIdx.y, sF, sB, threadIdx.x);
}

template <int SEG>
static void fb_rows_launch(hipStream_t st, const float* maps, float* dst, int n_maps, int w, int h, float e0, float p, const int* n_dev, PoseBlock* cumP, int cumN, int cumNdp,
                           float* world_scale) {
    const int Sr = (w + SEG - 1) / SEG, lpb = 256 / Sr;
    const bool vec4 = (w % 4) == 0 && (reinterpret_cast<uintptr_t>(maps) % 16) == 0 && (reinterpret_cast<uintptr_t>(dst) % 16) == 0;
    const dim3 g((h + lpb - 1) / lpb, n_maps + (cumP ? 1 : 0));
    if (vec4) hipLaunchKernelGGL((k_fb_rows<true, SEG>), g, dim3(256), 0, st, maps, dst, w, h, Sr, e0, p, n_dev, n_maps, cumP, cumN, cumNdp, world_scale);
    else hipLaunchKernelGGL((k_fb_rows<false, SEG>), g, dim3(256), 0, st, maps, dst, w, h, Sr, e0, p, n_dev, n_maps, cumP, cumN, cumNdp, world_scale);
}
template <int SEG>
static void fb_cols_launch(hipStream_t st, float* maps, int n_maps, int w, int h, float e0, float p, const int* n_dev) {
    const int Sc = (h + SEG - 1) / SEG;
    hipLaunchKernelGGL(k_fb_cols<SEG>, dim3((w + FB_CW - 1) / FB_CW, n_maps), dim3(FB_CW * Sc), 0, st, maps, w, h, Sc, e0, p, n_dev);
}
static bool fb_smooth_segmented(int w, int h) { return !(w > 40 * FB_MAX_ROW_SEGS || h > 40 * FB_MAX_COL_SEGS); }
// Steps per lane.  A lane also chains the S - 1 segment matrices of its line up to its own segment (fb_incoming): S - 1 Moebius steps
// next to the 2 x FB_SEG of its segment.  In the latency regime (one or two waves per SIMD: 640x480, 1241x376) short segments
// win -- the dependent chain is what takes the time.  Where the pass fills the chip many times over it is bound by VALU issue (0.91
// at 1080p) and the chaining is half of all instructions with 20-step segments (1920 wide: 95 + 40 steps per lane): 40-step
// segments then do the same work in 37 % fewer instructions.
static void fb_smooth_plan(int w, int h, int n_maps, int* rows_seg, int* cols_seg) {
    // steps per lane: 12 while the maps give fewer than 8 waves per SIMD at 20 (the pass is its dependent chain: shorter segments, more lanes; rows
    // 11.5 -> 10.2 us at 1241x376x5, window -1 % at 8 frames, profiles/r05h_*), 40 from 8 M pixels (fewer segment matrices to scan), 20 where 12 does
    // not cover the line
    const int forced = debug_switches().fb_segment;
    const bool many_waves = (size_t)w * h * n_maps >= ((size_t)8 << 20);
    const int want = forced ? forced : (many_waves ? 40 : 12);
    auto pick = [&](int len, int max_segs) { return (want == 12 && len <= 12 * max_segs) ? 12 : (want <= 20 && len <= 20 * max_segs) ? 20 : 40; };
    *rows_seg = pick(w, FB_MAX_ROW_SEGS);
    *cols_seg = pick(h, FB_MAX_COL_SEGS);
}
void fb_smooth_plan_segments(int w, int h, int n_maps, int* rows_seg, int* cols_seg, bool* segmented) {
    *segmented = fb_smooth_segmented(w, h);
    fb_smooth_plan(w, h, n_maps, rows_seg, cols_seg);
}
int fb_smooth_device(Context* c, float* maps, int n_maps, int w, int h, float s0_ems_prob, float no_change_prob, const int* n_dev, PoseBlock* cumP,
                     int cumN, int cumNdp, float* world_scale, float* dst, hipStream_t st_in) {
    if (n_maps <= 0) return 0;
    hipStream_t st = st_in ? st_in : c->stream;
    if (!fb_smooth_segmented(w, h)) {
        // larger than any segmented launch: one lane per line walks the recurrence step by step (the reference's own structure,
        // fb_smooth.h:26-69; no size limit).  The projective maps of the cost kernels then need their own launch.
        if (cumP) hipLaunchKernelGGL(k_cum_poses, dim3(1), dim3(64), 0, st, cumP, cumN, cumNdp, world_scale);
        return fb_smooth_strict_device(c, maps, n_maps, w, h, s0_ems_prob, no_change_prob, n_dev, dst, st);
    }
    int rs, cs;
    fb_smooth_plan(w, h, n_maps, &rs, &cs);
    float* out = dst ? dst : maps;
    if (rs == 12) fb_rows_launch<12>(st, maps, out, n_maps, w, h, s0_ems_prob, no_change_prob, n_dev, cumP, cumN, cumNdp, world_scale);
    else if (rs == 20) fb_rows_launch<20>(st, maps, out, n_maps, w, h, s0_ems_prob, no_change_prob, n_dev, cumP, cumN, cumNdp, world_scale);
    else fb_rows_launch<40>(st, maps, out, n_maps, w, h, s0_ems_prob, no_change_prob, n_dev, cumP, cumN, cumNdp, world_scale);
    if (cs == 12) fb_cols_launch<12>(st, out, n_maps, w, h, s0_ems_prob, no_change_prob, n_dev);
    else if (cs == 20) fb_cols_launch<20>(st, out, n_maps, w, h, s0_ems_prob, no_change_prob, n_dev);
    else fb_cols_launch<40>(st, out, n_maps, w, h, s0_ems_prob, no_change_prob, n_dev);
    VK_CHECK_LAST();
    return 0;
}

__global__ static void k_scale(float* p, const float* s_dev, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] *= *s_dev;
}

void cum_poses_launch(Context* c, PoseBlock* P, int N, int N_dp, float* world_scale) { hipLaunchKernelGGL(k_cum_poses, dim3(1), dim3(64), 0, c->stream, P, N, N_dp, world_scale); }
void reduce_density_launch(Context* c, const float* partial, int nblk, int npx, PoseBlock* P, int n_launch, float* scale_out, int scale_ready) {
    ReduceArgs a;
    a.partial = partial; a.nblk = nblk; a.npx = npx; a.n_launch = n_launch; a.scale_ready = scale_ready; a.cams = c->cams.as<CamState>(); a.P = P; a.scale_out = scale_out;
    hipLaunchKernelGGL(k_reduce_density, dim3(n_launch + (scale_out ? 1 : 0)), dim3(256), 0, c->stream, a);
}
// a reduction the window pipeline left for the next correspondence trace (Context::pending_reduce) and that no trace has picked up
int flush_pending_reduce(Context* c) {
    if (!c->pending_reduce.partial) return 0;
    const ReduceArgs a = c->pending_reduce;
    c->pending_reduce = ReduceArgs();
    hipLaunchKernelGGL(k_reduce_density, dim3(a.n_launch + (a.scale_out ? 1 : 0)), dim3(256), 0, c->stream, a);
    VK_CHECK_LAST();
    return 0;
}

static int optimize_depth_dispatch(Context* c, ImageSet& S, const OdParams& p, bool cost_only) {
    if (p.strict || p.w < 2 || p.h < 2) {  // parity-pinning mode (two frame bounds are enough); 1-pixel-wide images: the lean bilinear needs 2x2 texels
        if (p.N <= 8) return optimize_depth_launch<8, true>(c, S, p, cost_only);
        return optimize_depth_launch<16, true>(c, S, p, cost_only);
    }
    if (p.N <= 4) return optimize_depth_launch<4, false>(c, S, p, cost_only);
    if (p.N <= 6) return optimize_depth_launch<6, false>(c, S, p, cost_only);  // the SLAM driver's window is 5 flows (voldor_slam.py:85)
    if (p.N <= 8) return optimize_depth_launch<8, false>(c, S, p, cost_only);
    if (p.N <= 12) return optimize_depth_launch<12, false>(c, S, p, cost_only);  // per-frame register arrays are sized by the bound: 139 -> fewer VGPRs than <16>
    return optimize_depth_launch<16, false>(c, S, p, cost_only);
}


int optimize_depth_device(Context* c, ImageSet& S, const OdParams& p) {
    const int w = p.w, h = p.h;
    if (int e = S.cost.reserve(sizeof(float) * (size_t)w * h)) return e;
    if (c->rand_w != w || c->rand_h != h) {  // reference re-inits the RNG when the size changes (:358-361)
        if (c->rand_w != -1) c->rand_epoch = 0;  // (-1: epoch was set explicitly through vk_set_rand_epoch)
        c->rand_w = w; c->rand_h = h;
    }
    const int nblk = ((w + 63) / 64) * ((h + 3) / 4);
    if (int e = c->local_tbl.reserve(sizeof(float) * (size_t)w * h)) return e;
    if (int e = c->rig_partial.reserve(sizeof(float) * (size_t)nblk * MAX_FRAMES)) return e;
    if (int e = c->cams.reserve(sizeof(CamState) * MAX_FRAMES)) return e;
    if (c->prof) prof_begin(c);
    if (int e = optimize_depth_dispatch(c, S, p, false)) return e;
    if (c->prof) prof_end(c, "optimize_depth");
    return 0;
}

// compute_cost_map alone (tests / parity probes)
int cost_map_device(Context* c, ImageSet& S, const OdParams& p) {
    if (int e = S.cost.reserve(sizeof(float) * (size_t)p.w * p.h)) return e;
    return optimize_depth_dispatch(c, S, p, true);
}

// ---- small per-pixel helpers used by the host pipeline -------------------------------------
__global__ static void k_fill(float* p, float v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ static void k_disp_to_depth(const float* disp, float* out, float bf, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bf / disp[i];  // voldor.cpp:33
}
// py_export.cpp:64-73: mean of the registered rigidness maps and the prior confidences
__global__ static void k_depth_conf(const float* rig, const float* confs, float* out, int n_flows, int n_dp, size_t npx) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    float s = 0.f;
    for (int f = 0; f < n_flows; f++) s += rig[(size_t)f * npx + i];
    for (int f = 0; f < n_dp; f++) s += confs[(size_t)f * npx + i];
    out[i] = s * (float)(1.0 / (double)(float)(n_flows + n_dp));  // cv::Mat /= n multiplies by (float)(1./n) (py_export.cpp:74)
}
int fill_device(Context* c, float* p, float v, size_t n) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, p, v, n);
    VK_CHECK_LAST();
    return 0;
}
int scale_device(Context* c, float* p, const float* s_dev, size_t n) {
    hipLaunchKernelGGL(k_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, p, s_dev, n);
    VK_CHECK_LAST();
    return 0;
}
int disp_to_depth_device(Context* c, const float* disp, float* out, float bf, size_t n) {
    hipLaunchKernelGGL(k_disp_to_depth, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, disp, out, bf, n);
    VK_CHECK_LAST();
    return 0;
}
int depth_conf_device(Context* c, const float* rig, const float* confs, float* out, int n_flows, int n_dp, size_t npx) {
    hipLaunchKernelGGL(k_depth_conf, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, c->stream, rig, confs, out, n_flows, n_dp, npx);
    VK_CHECK_LAST();
    return 0;
}

// ---- gblur (gpu-kernels/gblur.cu:12-72; no caller in the reference, named by north_star) ----
__global__ __launch_bounds__(256) static void k_gblur(const float* __restrict__ src, float* __restrict__ dst, int w, int h,
                                                       const float* __restrict__ g, int half, int horizontal) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), d = blockIdx.z;
    if (x >= w || y >= h) return;
    const float* s = src + (size_t)d * w * h;
    float sum = g[0] * s[y * w + x], sw = g[0];
    for (int k = 1; k < half; k++) {
        if (horizontal) {
            if (x + k < w) { sum += g[k] * s[y * w + x + k]; sw += g[k]; }
            if (x - k >= 0) { sum += g[k] * s[y * w + x - k]; sw += g[k]; }
        } else {
            if (y + k < h) { sum += g[k] * s[(y + k) * w + x]; sw += g[k]; }
            if (y - k >= 0) { sum += g[k] * s[(y - k) * w + x]; sw += g[k]; }
        }
    }
    dst[(size_t)d * w * h + y * w + x] = sum / sw;
}
int gblur_device(Context* c, const float* src, float* dst, float* tmp, float* gk_dev, int w, int h, int d, float sigma, int ksize) {
    if (ksize == 0) { ksize = (int)ceilf(6 * sigma); if (ksize < 3) ksize = 3; }
    int half = ksize / 2 + 1;
    if (half > 128) return (int)hipErrorInvalidValue;  // cudaErrorInvalidFilterSetting in the reference
    float g[128];
    for (int i = 0; i < half; i++) g[i] = expf(-(float)(i * i) / (float)(2 * sigma * sigma));
    VK_CHECK(hipMemcpyAsync(gk_dev, g, sizeof(float) * half, hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));  // g lives on this stack frame
    dim3 grid((w + 63) / 64, (h + 3) / 4, d);
    hipLaunchKernelGGL(k_gblur, grid, dim3(256), 0, c->stream, src, tmp, w, h, gk_dev, half, 0);  // vertical first (:67)
    hipLaunchKernelGGL(k_gblur, grid, dim3(256), 0, c->stream, tmp, dst, w, h, gk_dev, half, 1);
    VK_CHECK_LAST();
    return 0;
}

// counter k of the strict passes' fp32 filter (vk_debug.h "sf_*"): read and cleared
int strict_filter_stat(Context* c, int k) {
    if (!c->sf_stats.p) return 0;
    unsigned long long v = 0;
    unsigned long long* d = c->sf_stats.as<unsigned long long>() + k;
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(&v, d, sizeof v, hipMemcpyDeviceToHost) != hipSuccess || hipMemset(d, 0, sizeof v) != hipSuccess) return -1;
    return v > 0x7fffffffull ? 0x7fffffff : (int)v;
}

}  // namespace vk
