// voldor_amd/csrc/vk_depth_impl.hpp -- the per-pixel kernels of the depth / rigidness half, templated on the frame bound NMAX (per-frame arrays stay
// in registers) and on STRICT (the reference's operation order on software transcendentals), and the stage order that launches them
// (optimize_depth_launch<NMAX, STRICT>, optimize_depth.cu:462-494).  Included by one translation unit per (NMAX, STRICT) pair
// (vk_depth_i*.hip / vk_depth_s*.hip) so that the instantiations compile next to each other; the kernels that do not depend on NMAX live in vk_depth.hip.
#pragma once
#include "vk_common.hpp"
#include "vk_device.hpp"
#include "vk_strict_model.hpp"
#include "vk_ref_cuda.h"
#include "vk_internal.hpp"
#include <cstdlib>

namespace vk {

// vk_depth.hip (kernels that do not depend on the frame bound)
void cum_poses_launch(Context* c, PoseBlock* P, int N, int N_dp, float* world_scale);
void reduce_density_launch(Context* c, const float* partial, int nblk, int npx, PoseBlock* P, int n_launch, float* scale_out, int scale_ready);



// phase clocks (profiling builds only, scripts/phase_clocks.sh): thread 0 of the middle workgroup of a launch
#ifdef VK_PHASE_CLOCKS
static __device__ unsigned long long g_phase_d[64];  // (one copy per translation unit: profiling builds read the unit they profile)
#define PHD_ON (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2)
#define PHD_DECL unsigned long long ph_t = __builtin_amdgcn_s_memtime()
#define PHD_MARK(slot) do { if (PHD_ON) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_phase_d[slot], n_ - ph_t); ph_t = n_; } } while (0)
#define PHD_ADD(slot, v) do { if (PHD_ON) atomicAdd(&g_phase_d[slot], (unsigned long long)(v)); } while (0)
#else
#define PHD_DECL do {} while (0)
#define PHD_MARK(slot) do {} while (0)
#define PHD_ADD(slot, v) do {} while (0)
#endif

struct Img {
    const float2* __restrict__ flows;  // [N][h][w]
    float* __restrict__ rig;           // [N][h][w]
    const float* __restrict__ priors;  // [N_dp][h][w]
    const float* __restrict__ pconfs;
    float* __restrict__ confs;
    float* __restrict__ depth;
    float* __restrict__ cost;
    const PoseBlock* __restrict__ P;
    int N, N_dp, w, h;
    float lambda, omega, inv_arf, arf, basefocal, disp_delta, delta;
    // reference mode (strict kernels only; vk_ref_cuda.h): tex = CUDA's linear filter over the STACK of n_layers flow layers (the frame
    // count of the launch, before the device-side truncation clamps N) / of N_dp prior layers; xw = the per-pixel XORWOW states
    int tex, n_layers;
    vrc_xorwow* xw;
    unsigned long long* sf;  // verification counters of the fp32 filter (k_cost_rand_f_strict), or null
};
// at_tex of the reference (gmat.h:175-179) in the strict kernels: D2's exact per-layer bilinear, or CUDA's filter (--reference_tex 1)
__device__ __forceinline__ static float2 fetch_flow_strict(const Img& I, int f, float x, float y) {
    if (I.tex) { float2 r; vrc_tex_fetch2(reinterpret_cast<const float*>(I.flows), x, y, f, I.w, I.h, I.n_layers, &r.x, &r.y); return r; }
    return bilinear2(I.flows + (size_t)f * I.w * I.h, I.w, I.h, x, y);
}
__device__ __forceinline__ static float fetch_prior_strict(const Img& I, const float* __restrict__ stack, int f, float x, float y) {
    if (I.tex) return vrc_tex_fetch1(stack, x, y, f, I.w, I.h, I.N_dp);
    return bilinear1(stack + (size_t)f * I.w * I.h, I.w, I.h, x, y);
}
// Frames still registered: the launch-time count clamped by the device-side decision of this EM iteration
// (PoseBlock::n_active).  Returns false when nothing is left to evaluate (window lost, no priors).
__device__ __forceinline__ bool clamp_active(Img& I) {
    I.N = min(I.N, I.P->n_active);
    return I.N + I.N_dp > 0;
}

// depth-prior term of compute_pixel_cost (optimize_depth.cu:166-190): the hypothesis seen from prior f's camera against the
// prior map, weighted by the prior's confidences (all three sampled bilinearly at the projected position)
__device__ __forceinline__ static void prior_term_strict(const Img& I, const PoseBlock* P, int f, int px, int py, float depth, float& cost_sum, float& wsum) {
#pragma clang fp contract(off)  // strict: one rounding per operation, also across statements (wsum += wg must not become an fma)
    const int w = I.w, h = I.h;
    P3 q = transform(P->dpRs[f], P->dpts[f], backproject(P, (float)px, (float)py, depth));
    float qx2, qy2;
    project(P, q, qx2, qy2);
    if (q.z > 0.f && qx2 >= 0.f && qx2 < (float)w && qy2 >= 0.f && qy2 < (float)h) {
        float td = fetch_prior_strict(I, I.priors, f, qx2, qy2);
        if (td > 0.f) {
            float tpc = fetch_prior_strict(I, I.pconfs, f, qx2, qy2);
            float tc = fetch_prior_strict(I, I.confs, f, qx2, qy2);
            float wg = tpc * tc * ((I.disp_delta > 0.f && f == 0) ? I.disp_delta : I.delta);
            cost_sum = strict::cost_acc(cost_sum, wg, strict::depth_rigidness(q.z, td, I.basefocal, I.omega, I.arf));  // fun_depth_cost, residual_model.h:64-68
            wsum += wg;
        }
    }
}

// compute_pixel_cost, optimize_depth.cu:140-198.
// Three phases so that the N bilinear gathers of one hypothesis are all in flight together instead
// of one L2 round trip per frame (the sampling positions depend on depth and poses only, not on
// the flow values): (1) rigid chain -> positions + validity mask, (2) issue every gather,
// (3) residual model.  NMAX is the compile-time frame bound (arrays stay in registers).
// STRICT kernels: the reference's un-fused fp32 geometry (vk_device.hpp backproject / transform / project, true divisions) and the
// residual model in the reference's operation order on the software transcendentals (vk_strict_model.hpp): bit-identical to the
// oracle in strict mode.  The fast ("lean") kernels further down share the structure, not the arithmetic.
template <int NMAX>
__device__ __forceinline__ static float pixel_cost_strict(const Img& I, int px, int py, float depth) {
#pragma clang fp contract(off)
    const int w = I.w, h = I.h, npx = w * h, pi = py * w + px;
    const PoseBlock* P = I.P;
    float qx[NMAX], qy[NMAX], rdx[NMAX], rdy[NMAX];
    unsigned valid = 0;
    {
        P3 o = backproject(P, (float)px, (float)py, depth);
        float px1 = (float)px, py1 = (float)py;
#pragma unroll
        for (int f = 0; f < NMAX; f++) {
            qx[f] = 0.f; qy[f] = 0.f; rdx[f] = 0.f; rdy[f] = 0.f;
            if (f < I.N) {
                o = transform(P->Rs[f], P->ts[f], o);
                float px2, py2;
                project(P, o, px2, py2);
                if (o.z > 0.f && px1 >= 0.f && px1 < (float)w && py1 >= 0.f && py1 < (float)h) {
                    valid |= 1u << f;
                    qx[f] = px1; qy[f] = py1; rdx[f] = px2 - px1; rdy[f] = py2 - py1;
                    px1 = px2; py1 = py2;  // advances on contributing frames only (:162-164)
                }
            }
        }
    }
    float2 obs[NMAX];
    float wgt[NMAX];
#pragma unroll
    for (int f = 0; f < NMAX; f++) {
        obs[f] = make_float2(0.f, 0.f); wgt[f] = 0.f;
        if (f < I.N) {  // unconditional (clamped) gathers: no divergent branch around the loads
            // frame 0 is sampled at the pixel itself: weights (1,0,0,0), the fetch is the texel (and it does not
            // depend on the depth hypothesis, so it leaves the candidate loops)
            obs[f] = (f == 0) ? I.flows[pi] : fetch_flow_strict(I, f, qx[f], qy[f]);
            wgt[f] = I.rig[(size_t)f * npx + pi];
        }
    }
    float cost_sum = 0.f, wsum = 0.f;
#pragma unroll
    for (int f = 0; f < NMAX; f++) {
        if (f < I.N && ((valid >> f) & 1u)) {
            // product rounded before the add (no fma): same value as the lane-split evaluation cost_split8
            cost_sum = strict::cost_acc(cost_sum, wgt[f], strict::rigidness(rdx[f], rdy[f], obs[f].x, obs[f].y, I.lambda, I.arf));  // fun_cost :45-49
            wsum += wgt[f];
        }
    }
    for (int f = 0; f < I.N_dp; f++) prior_term_strict(I, P, f, px, py, depth, cost_sum, wsum);
    if (wsum == 0.f) return INFINITY;
    return cost_sum / fmaxf(wsum, 1.1920929e-07f);
}

// ---- cost map + all random samples, fused (optimize_depth.cu:279-284 + :269-277 x n_rand) ----
template <int NMAX>
__global__ __launch_bounds__(256) static void k_cost_rand_strict(Img I, int n_rand, uint32_t epoch0, float range_factor) {
#pragma clang fp contract(off)  // the random depth range_factor * u + 1/MAXIMUM_DEPTH is an output value: no fma
    if (!clamp_active(I)) return;
    const int tile = xcd_band_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int x = (tile % gridDim.x) * 64 + (threadIdx.x & 63);
    const int y = (tile / gridDim.x) * 4 + (threadIdx.x >> 6);
    if (x >= I.w || y >= I.h) return;
    const int pi = y * I.w + x;
    float d = I.depth[pi];
    float c = pixel_cost_strict<NMAX>(I, x, y, d);
    vrc_xorwow st;
    if (I.xw) st = I.xw[pi];  // --reference_rng 1: curand_uniform(&_d_rand_states.at(x, y)), the state persists (:273)
    for (int it = 0; it < n_rand; it++) {
        float u = I.xw ? vrc_uniform(vrc_xorwow_next(&st)) : u01(rng3(RAND_SEED, (uint32_t)pi, epoch0 + (uint32_t)it));
        float dn = 1.0f / (range_factor * u + (1.0f / 1e5f));  // MAXIMUM_DEPTH, :15,:273
        float cn = pixel_cost_strict<NMAX>(I, x, y, dn);
        if (cn < c) { c = cn; d = dn; }
    }
    if (I.xw) I.xw[pi] = st;
    I.depth[pi] = d;
    I.cost[pi] = c;
}

// replace_if_better_depth, optimize_depth.cu:201-207
template <int NMAX>
__device__ __forceinline__ static void try_depth_strict(const Img& I, int x, int y, float cand) {
    const int pi = y * I.w + x;
    float c = pixel_cost_strict<NMAX>(I, x, y, cand);
    if (c < I.cost[pi]) { I.depth[pi] = cand; I.cost[pi] = c; }
}

// ---- global propagation (optimize_depth.cu:209-235). With step>=2 the sites of one pass are
// independent (reads x-1, writes x): one thread per site.  dir: 0 L2R,
// 1 T2B, 2 R2L, 3 B2T.
template <int NMAX>
__global__ __launch_bounds__(256) static void k_global_prop_sites_strict(Img I, int dir, int step, int nsites) {
    if (!clamp_active(I)) return;
    const int tile = xcd_band_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int s = (tile % gridDim.x) * blockDim.x + threadIdx.x;  // site index along the pass direction
    const int l = tile / gridDim.x;                               // line (row for 0/2, column for 1/3)
    if (s >= nsites) return;
    if (dir == 0) { int x = 1 + s * step; try_depth_strict<NMAX>(I, x, l, I.depth[l * I.w + x - 1]); }
    else if (dir == 2) { int x = I.w - 2 - s * step; try_depth_strict<NMAX>(I, x, l, I.depth[l * I.w + x + 1]); }
    else if (dir == 1) { int y = 1 + s * step; try_depth_strict<NMAX>(I, l, y, I.depth[(y - 1) * I.w + l]); }
    else { int y = I.h - 2 - s * step; try_depth_strict<NMAX>(I, l, y, I.depth[(y + 1) * I.w + l]); }
}

// Segment geometry of one local pass (optimize_depth.cu:242-265): chain `seg` of `line` visits n
// pixels pi0, pi0+stride, ...; the first candidate is the depth of the pixel before pi0.
struct ChainGeom { int pi0, stride, n, prev0; };
__device__ __forceinline__ ChainGeom chain_geom(int w, int h, int dir, int width, int line, int seg) {
    ChainGeom g;
    if (dir == 0) {        // L2R: x = max(1,px+1) .. min(w,px+width)-1 ascending, candidate depth[x-1]
        const int px = seg * width, x0 = max(1, px + 1);
        g.n = min(w, px + width) - x0; g.pi0 = line * w + x0; g.stride = 1;
    } else if (dir == 2) { // R2L: x = min(w-2,px+width-2) .. max(0,px) descending, candidate depth[x+1]
        const int px = seg * width, x0 = min(w - 2, px + width - 2);
        g.n = x0 - max(0, px) + 1; g.pi0 = line * w + x0; g.stride = -1;
    } else if (dir == 1) { // T2B
        const int py = seg * width, y0 = max(1, py + 1);
        g.n = min(h, py + width) - y0; g.pi0 = y0 * w + line; g.stride = w;
    } else {               // B2T
        const int py = seg * width, y0 = min(h - 2, py + width - 2);
        g.n = y0 - max(0, py) + 1; g.pi0 = y0 * w + line; g.stride = -w;
    }
    g.prev0 = g.pi0 - g.stride;
    return g;
}

// ---- E-step (optimize_depth.cu:84-138) + per-block sums of each rigidness map (the density
// test of voldor.cpp:171 then needs no D2H of the maps).  Same three-phase structure as pixel_cost.
template <int NMAX>
__global__ __launch_bounds__(256) static void k_update_rigidness_strict(Img I, float* __restrict__ partial) {
#pragma clang fp contract(off)
    if (!clamp_active(I)) return;
    const int tile = xcd_band_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int x = (tile % gridDim.x) * 64 + (threadIdx.x & 63);
    const int y = (tile / gridDim.x) * 4 + (threadIdx.x >> 6);
    const bool live = x < I.w && y < I.h;
    const int w = I.w, h = I.h, npx = w * h, pi = live ? y * w + x : 0;
    const PoseBlock* P = I.P;
    __shared__ float s_part[NMAX][4];
    const int blk = tile, nblk = gridDim.x * gridDim.y;
    const float d = live ? I.depth[pi] : 1.f;
    float qx[NMAX], qy[NMAX], rdx[NMAX], rdy[NMAX];
    unsigned valid = 0;
    {
        P3 o = backproject(P, (float)x, (float)y, d);
        float px1 = (float)x, py1 = (float)y;
#pragma unroll
        for (int f = 0; f < NMAX; f++) {
            qx[f] = 0.f; qy[f] = 0.f; rdx[f] = 0.f; rdy[f] = 0.f;
            if (f < I.N) {
                o = transform(P->Rs[f], P->ts[f], o);
                float px2, py2;
                project(P, o, px2, py2);
                if (live && o.z > 0.f && px1 >= 0.f && px1 < (float)w && py1 >= 0.f && py1 < (float)h) {
                    valid |= 1u << f;
                    qx[f] = px1; qy[f] = py1; rdx[f] = px2 - px1; rdy[f] = py2 - py1;
                    px1 = px2; py1 = py2;  // NOT advanced on invalid frames
                }
            }
        }
    }
    float2 obs[NMAX];
#pragma unroll
    for (int f = 0; f < NMAX; f++) {
        obs[f] = make_float2(0.f, 0.f);
        if (f < I.N) obs[f] = (f == 0) ? I.flows[pi] : fetch_flow_strict(I, f, qx[f], qy[f]);
    }
#pragma unroll
    for (int f = 0; f < NMAX; f++) {
        if (f < I.N) {
            float r = 0.f;
            if ((valid >> f) & 1u)
                r = strict::rigidness(rdx[f], rdy[f], obs[f].x, obs[f].y, I.lambda, I.arf);
            if (live) I.rig[(size_t)f * npx + pi] = r;
            float ws = wave_sum(live ? r : 0.f);
            if ((threadIdx.x & 63) == 0) s_part[f][threadIdx.x >> 6] = ws;
        }
    }
    __syncthreads();
    if (threadIdx.x < NMAX && (int)threadIdx.x < I.N) {
        const int f = threadIdx.x;
        partial[(size_t)f * nblk + blk] = (s_part[f][0] + s_part[f][1]) + (s_part[f][2] + s_part[f][3]);
    }
    if (!live) return;
    for (int f = 0; f < I.N_dp; f++) {
        P3 q = transform(P->dpRs[f], P->dpts[f], backproject(P, (float)x, (float)y, d));
        float qx2, qy2;
        project(P, q, qx2, qy2);
        if (q.z > 0.f && qx2 >= 0.f && qx2 < (float)w && qy2 >= 0.f && qy2 < (float)h) {
            float td = fetch_prior_strict(I, I.priors, f, qx2, qy2);
            if (td > 0.f)  // else: the confidence is left untouched (:129)
                I.confs[(size_t)f * npx + pi] = strict::depth_rigidness(q.z, td, I.basefocal, I.omega, I.arf);
        } else
            I.confs[(size_t)f * npx + pi] = 0.f;
    }
}
// =================================================================================================================================
// LEAN fast path.  Same algorithm, same decisions up to rounding as the strict kernels above (which keep the reference's un-fused
// fp32 geometry and operation order: bit-identical to the oracle); the fast kernels are held to them by measured distances
// (tests/test_gpu_strict.py, tests/test_gpu_kernels.py):
//   * the rigid chain of a hypothesis is ONE projective map per frame (PoseBlock::cumM / cumT, k_cum_poses): the homogeneous
//     pixel in frame f+1 is d * (cumM[f] (x,y,1)) + cumT[f] -- 3 fma + 1 v_rcp + 2 mul per hypothesis and frame instead of a 3x3
//     transform, two IEEE divisions and a re-projection; cumM[f] (x,y,1) is shared by all hypotheses of a pixel
//   * a bilinear fetch at a position known to be inside the image needs no clamps (bilinear2_inside)
//   * the observation-only half of the residual model is split off (obs_terms: once per gather, once per PIXEL for frame 0)
// The validity rules (z > 0, previous position inside the image, position advanced on contributing frames only) are unchanged.
struct LeanK { float ia2, qia2, l2q; };  // 1/arf^2, 0.25/arf^2, log2(0.25 lambda^2)
__device__ __forceinline__ LeanK lean_consts(const Img& I) {
    LeanK k;
    k.ia2 = I.inv_arf * I.inv_arf; k.qia2 = 0.25f * k.ia2; k.l2q = fast_log2(0.25f * I.lambda * I.lambda);
    return k;
}
// ONE ARITHMETIC FOR EVERY KERNEL.  The depth search compares the cost a kernel computes now with the cost another kernel stored
// earlier (`cost < io_cost`, optimize_depth.cu:201-207).  Large regions of a converged map share one depth value, so "the neighbour's
// depth" is often the pixel's own: if two kernels round the same cost differently, the comparison fires on the last bit, nothing
// changes but a run starts in k_local_runs (measured at 1080p: 2x the time of that kernel).  Hence a single operation sequence --
// lean_head (frame 0, then the depth priors), lean_rest (frames 1.. summed in log2 units), cs = head + ln2 * rest, cost = cs * rcp(ws)
// -- with explicit fma and no compiler contraction, used verbatim by the cost map, the samples, both propagations, and reproduced
// term by term by the lane-split evaluation of k_local_runs.
// depth-prior term (optimize_depth.cu:166-190), the prior pose as one projective map: weight and -log(confidence) of the hypothesis
__device__ __forceinline__ static bool prior_parts(const Img& I, const PoseBlock* P, int f, float x, float y, float depth, float& wg, float& term) {
#pragma clang fp contract(off)
    const int w = I.w, h = I.h, npx = w * h;
    wg = 0.f; term = 0.f;
    float hz, td, tpc, tc;
    if ((P->dp_ident >> f) & 1) {
        // The usual prior -- the disparity / depth map of the reference frame itself -- sits at the identity pose: every hypothesis of the
        // pixel lands on the pixel, its camera-space depth is the hypothesis, and the three maps are read AT the pixel (independent of the
        // hypothesis: ten random samples share one read).  ~100 instructions of projection, index arithmetic and bilinear weights (all
        // of them 1, 0, 0, 0) per hypothesis less; 1080p N=10 with a disparity prior: the sample pass 504 -> see DESIGN.md.
        if (!(depth > 0.f)) return false;
        const int pi = (int)y * w + (int)x;
        hz = depth;
        td = I.priors[(size_t)f * npx + pi];
        if (!(td > 0.f)) return false;
        tpc = I.pconfs[(size_t)f * npx + pi]; tc = I.confs[(size_t)f * npx + pi];
    } else {
        const H3 a = hom_dir(P->dpM[f], x, y);
        // true divisions here (one per prior, not per frame)
        hz = fmaf(depth, a.z, P->dpT[f][2]);
        const float qx2 = fmaf(depth, a.x, P->dpT[f][0]) / hz, qy2 = fmaf(depth, a.y, P->dpT[f][1]) / hz;
        if (!(hz > 0.f && qx2 >= 0.f && qx2 < (float)w && qy2 >= 0.f && qy2 < (float)h)) return false;
        td = bilinear1(I.priors + (size_t)f * npx, w, h, qx2, qy2);
        if (!(td > 0.f)) return false;
        tpc = bilinear1(I.pconfs + (size_t)f * npx, w, h, qx2, qy2);
        tc = bilinear1(I.confs + (size_t)f * npx, w, h, qx2, qy2);
    }
    wg = tpc * tc * ((I.disp_delta > 0.f && f == 0) ? I.disp_delta : I.delta);
    term = 0.6931471805599453f * fast_log2(1.f + depth_ratio(hz, td, I.basefocal, I.omega, I.inv_arf));
    return true;
}
__device__ __forceinline__ static void prior_term_lean(const Img& I, const PoseBlock* P, int f, float x, float y, float depth, float& cost_sum, float& wsum) {
    float wg, term;
    if (prior_parts(I, P, f, x, y, depth, wg, term)) { cost_sum = fmaf(wg, term, cost_sum); wsum += wg; }
}
// one frame step of the chain: homogeneous pixel -> position in the next frame; returns z > 0
__device__ __forceinline__ static bool lean_step(const PoseBlock* P, int f, float x, float y, float d, float& px2, float& py2) {
#pragma clang fp contract(off)
    const H3 a = hom_dir(P->cumM[f], x, y);
    const float hz = fmaf(d, a.z, P->cumT[f][2]), iz = fast_rcp(hz);
    px2 = fmaf(d, a.x, P->cumT[f][0]) * iz; py2 = fmaf(d, a.y, P->cumT[f][1]) * iz;
    return hz > 0.f;
}
// the same step for two pixels of one column (rows y.x, y.y) on float pairs: the bits of lean_step for either
__device__ __forceinline__ static void lean_step2(const PoseBlock* P, int f, float x, pf2 y, pf2 d, pf2& px2, pf2& py2, bool& zok_a, bool& zok_b) {
#pragma clang fp contract(off)
    const float* __restrict__ M = P->cumM[f];
    const pf2 ax = pk_fma(pk_all(M[0]), pk_all(x), pk_fma(pk_all(M[1]), y, pk_all(M[2])));
    const pf2 ay = pk_fma(pk_all(M[3]), pk_all(x), pk_fma(pk_all(M[4]), y, pk_all(M[5])));
    const pf2 az = pk_fma(pk_all(M[6]), pk_all(x), pk_fma(pk_all(M[7]), y, pk_all(M[8])));
    const pf2 hz = pk_fma(d, az, pk_all(P->cumT[f][2]));
    const pf2 iz = { fast_rcp(hz.x), fast_rcp(hz.y) };
    px2 = pk_fma(d, ax, pk_all(P->cumT[f][0])) * iz; py2 = pk_fma(d, ay, pk_all(P->cumT[f][1])) * iz;
    zok_a = hz.x > 0.f; zok_b = hz.y > 0.f;
}
// frame 0 (observed at the pixel itself) + depth priors of one hypothesis; leaves the position the chain continues from
__device__ __forceinline__ static void lean_head(const Img& I, const LeanK& K, const PoseBlock* P, float x, float y, float d, float2 o0, const ObsTerms& T0,
                                                 float wgt0, float& cs, float& ws, float& px1, float& py1) {
#pragma clang fp contract(off)
    cs = 0.f; ws = 0.f; px1 = x; py1 = y;
    if (I.N > 0) {
        float px2, py2;
        if (lean_step(P, 0, x, y, d, px2, py2)) {  // the pixel itself is always inside the image
            cs = wgt0 * (0.6931471805599453f * fast_log2(1.f + obs_ratio(T0, (px2 - x) - o0.x, (py2 - y) - o0.y, K.qia2)));
            ws = wgt0;
            px1 = px2; py1 = py2;
        }
    }
    for (int f = 0; f < I.N_dp; f++) prior_term_lean(I, P, f, x, y, d, cs, ws);
}
// frames 1.. of compute_pixel_cost (optimize_depth.cu:140-198), CH frames at a time: positions of the chunk, then its gathers, then the
// model.  CH = 1 (one frame after the other) is the default at every size.  Measured (sample pass / table pass per launch): keeping all
// gathers of a hypothesis in flight together (CH = number of frames, the round-2 kernels) costs 7 staging registers per frame --
// 122 VGPRs = 4 waves per SIMD for 12 frames; CH = 4 / 3 / 2 / 1 at 1080p N=10: 423 / 412 / 408 / 383 us and 62.4 / 61.5 / 59.7 / 58.6 us
// (from 457 and 69.6), 1241x376 N=8: 103.5 -> 86.4 us, 640x480 N=5: 45.9 -> 43.1 us.  More resident waves hide the gather latency better
// than more gathers per wave.  The sums run over the frames in the same order for every CH: same bits.
constexpr int LEAN_CHUNK = 1;
template <int NMAX, int CH = LEAN_CHUNK>
__device__ __forceinline__ static void lean_rest(const Img& I, const LeanK& K, const PoseBlock* P, int pi, float x, float y, float d, float px1, float py1,
                                                 float& cs, float& ws) {
#pragma clang fp contract(off)
    const int w = I.w, h = I.h, npx = w * h;
    const float fw = (float)w, fh = (float)h;
    float cl = 0.f;
#pragma unroll
    for (int f0 = 1; f0 < NMAX; f0 += CH) {
        float qx[CH], qy[CH], ex[CH], ey[CH];
        unsigned valid = 0;
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int f = f0 + j;
            if (f < NMAX && f < I.N) {  // uniform.  Inside: selects, no divergent branch (a branch per frame costs the zero-fill of its four slots twice over)
                float px2, py2;
                const bool zok = lean_step(P, f, x, y, d, px2, py2);
                const bool ok = zok && px1 >= 0.f && px1 < fw && py1 >= 0.f && py1 < fh;
                valid |= ok ? (1u << j) : 0u;
                qx[j] = ok ? px1 : 0.f; qy[j] = ok ? py1 : 0.f;  // a frame that does not contribute gathers texel (0,0) and is dropped below
                ex[j] = px2 - px1; ey[j] = py2 - py1;
                px1 = ok ? px2 : px1; py1 = ok ? py2 : py1;  // advances on contributing frames only (:162-164)
            } else { qx[j] = 0.f; qy[j] = 0.f; ex[j] = 0.f; ey[j] = 0.f; }
        }
        float2 obs[CH];
        float wgt[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int f = f0 + j;
            obs[j] = make_float2(0.f, 0.f); wgt[j] = 0.f;
            if (f < NMAX && f < I.N) { obs[j] = bilinear2_inside(I.flows + (size_t)f * npx, w, h, qx[j], qy[j]); wgt[j] = I.rig[(size_t)f * npx + pi]; }
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int f = f0 + j;
            if (f < NMAX && f < I.N && ((valid >> j) & 1u)) {
                const ObsTerms T = obs_terms(obs[j].x, obs[j].y, K.ia2, K.l2q);
                cl = fmaf(wgt[j], fast_log2(1.f + obs_ratio(T, ex[j] - obs[j].x, ey[j] - obs[j].y, K.qia2)), cl);
                ws += wgt[j];
            }
        }
    }
    cs = fmaf(0.6931471805599453f, cl, cs);
}
__device__ __forceinline__ float lean_final(float cs, float ws) {
#pragma clang fp contract(off)
    return ws == 0.f ? INFINITY : cs * fast_rcp(fmaxf(ws, 1.1920929e-07f));
}
// the whole cost of one hypothesis at pixel (px, py)
template <int NMAX, int CH = LEAN_CHUNK>
__device__ __forceinline__ static float pixel_cost_lean(const Img& I, const LeanK& K, int px, int py, float depth) {
    const int pi = py * I.w + px;
    const float x = (float)px, y = (float)py;
    const float2 o0 = I.N > 0 ? I.flows[pi] : make_float2(0.f, 0.f);
    const ObsTerms T0 = obs_terms(o0.x, o0.y, K.ia2, K.l2q);
    const float wgt0 = I.N > 0 ? I.rig[pi] : 0.f;
    float cs, ws, px1, py1;
    lean_head(I, K, I.P, x, y, depth, o0, T0, wgt0, cs, ws, px1, py1);
    lean_rest<NMAX, CH>(I, K, I.P, pi, x, y, depth, px1, py1, cs, ws);
    return lean_final(cs, ws);
}

// one arithmetic switch for the kernels that exist in both modes
template <int NMAX, bool STRICT, int CH = LEAN_CHUNK>
__device__ __forceinline__ static float pixel_cost_any(const Img& I, const LeanK& K, int px, int py, float depth) {
    if constexpr (STRICT) return pixel_cost_strict<NMAX>(I, px, py, depth); else return pixel_cost_lean<NMAX, CH>(I, K, px, py, depth);
}

// ---- cost map + random samples with EXACT EARLY REJECTION and SURVIVOR COMPACTION ---------------------------------------------------
// (a) The cost is sum(w_f c_f) / sum(w_f) over the contributing frames with every c_f >= 0, so after the part that needs no divergent
// gather (lean_head: frame 0 and the depth priors) the final cost is at least cs / (ws + wrest), wrest = the weight frames 1.. can
// add at most.  A random sample is only ever compared with the running best (`cost < best`, optimize_depth.cu:201-207): once that
// lower bound exceeds the incumbent's cost the outcome is decided.  Most random depths are far off and die here, before the gathers at
// positions that differ from lane to lane (64 distinct cache lines per load instruction: what the round-1 kernel was bound by).  The
// 1e-5 margin keeps the float rounding of the final quotient on the safe side: near-ties are evaluated in full.
// (b) Rejection alone spares the loads of a dead sample, not its VALU slots: a wave keeps issuing frames 1.. of sample k as long as ONE
// of its 64 lanes has it alive (measured: same instruction count as without rejection).  So the survivors of a workgroup's 64x4 pixel
// tile are appended to a queue in LDS and frames 1.. are evaluated over the QUEUE, one entry per lane, dense: ~2 full evaluations per
// pixel instead of 10.
//   1. every lane evaluates its pixel's incumbent depth in full (its cost is the rejection bound)
//   2. per round of CRQ_NS samples: lean_head of each sample -> survivors into the queue
//   3. queue entries (pixel, sample, depth, partial sums) are evaluated by whichever lane picks them up; the result goes into the
//      pixel's 64-bit LDS slot with atomicMin on (cost bits << 32 | sample index): the cheapest sample, the earliest among equals --
//      the one the sequential rule of optimize_depth.cu:269-277 would end up with
//   4. the pixel takes the winner if it is strictly cheaper than its running best
// The result does not depend on the order in which entries enter or leave the queue.
constexpr int CRQ_NS = 5;  // samples per round: queue capacity 256 * CRQ_NS entries (20 KB of LDS).  One round of 10 (40 KB): 45 -> 56 us at 640x480, no gain at 1080p -- the second round prunes against the winners of the first
struct CrqEntry { unsigned id; float d, cs, ws; };  // id = lane-in-workgroup | sample << 8
__device__ __forceinline__ float sample_depth(int pi, uint32_t epoch, float range_factor) {
#pragma clang fp contract(off)
    const float u = u01(rng3(RAND_SEED, (uint32_t)pi, epoch));
    return 1.0f / (range_factor * u + (1.0f / 1e5f));  // MAXIMUM_DEPTH, optimize_depth.cu:15,:273 (exact: the depth VALUES are outputs)
}
template <int NMAX>
__global__ __launch_bounds__(256) static void k_cost_rand_q(Img I, int n_rand, uint32_t epoch0, float range_factor) {
    __shared__ unsigned long long s_best[256];
    __shared__ CrqEntry s_q[256 * CRQ_NS];
    __shared__ int s_qn;
    if (!clamp_active(I)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int tile = xcd_band_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int x0 = (tile % gridDim.x) * 64, y0 = (tile / gridDim.x) * 4;
    const int xi = x0 + lane, yi = y0 + (tid >> 6);
    const bool live = xi < I.w && yi < I.h;
    const int w = I.w, npx = w * I.h, pi = live ? yi * w + xi : 0;
    const PoseBlock* P = I.P;
    const LeanK K = lean_consts(I);
    const float x = (float)(live ? xi : 0), y = (float)(live ? yi : 0);
    const float2 o0 = I.N > 0 ? I.flows[pi] : make_float2(0.f, 0.f);
    const ObsTerms T0 = obs_terms(o0.x, o0.y, K.ia2, K.l2q);
    const float wgt0 = I.N > 0 ? I.rig[pi] : 0.f;
    float wrest = 0.f;
    for (int f = 1; f < I.N; f++) wrest += I.rig[(size_t)f * npx + pi];
    // 1. the incumbent
    float d_best = I.depth[pi], c_best;
    {
        float cs, ws, px1, py1;
        lean_head(I, K, P, x, y, d_best, o0, T0, wgt0, cs, ws, px1, py1);
        lean_rest<NMAX>(I, K, P, pi, x, y, d_best, px1, py1, cs, ws);
        c_best = lean_final(cs, ws);
    }
    for (int it = 0; it < n_rand; it += CRQ_NS) {
        const int nh = min(CRQ_NS, n_rand - it);
        s_best[tid] = ~0ull;
        if (tid == 0) s_qn = 0;
        __syncthreads();
        // 2. heads of this round's samples; survivors enter the queue (one LDS atomic per wave and sample)
        for (int k = 0; k < nh; k++) {
            const float d = sample_depth(pi, epoch0 + (uint32_t)(it + k), range_factor);
            float cs, ws, px1, py1;
            lean_head(I, K, P, x, y, d, o0, T0, wgt0, cs, ws, px1, py1);
            const bool alive = live && !(cs > c_best * (ws + wrest) * 1.00001f);
            const unsigned long long m = __ballot(alive);
            int base = 0;
            if (lane == 0 && m) base = atomicAdd(&s_qn, __popcll(m));
            base = __shfl(base, 0, 64);
            if (alive) s_q[base + __popcll(m & ((1ull << lane) - 1ull))] = { (unsigned)tid | ((unsigned)k << 8), d, cs, ws };
        }
        __syncthreads();
        // 3. frames 1.. over the queue
        const int qn = s_qn;
        for (int e = tid; e < qn; e += 256) {
            const CrqEntry q = s_q[e];
            const int t = (int)(q.id & 255u), k = (int)(q.id >> 8);
            const int ex_ = x0 + (t & 63), ey_ = y0 + (t >> 6), epi = ey_ * w + ex_;
            const float fx_ = (float)ex_, fy_ = (float)ey_;
            float px1 = fx_, py1 = fy_;
            if (I.N > 0) {  // where the chain stands after frame 0 (the same step lean_head took)
                float px2, py2;
                if (lean_step(P, 0, fx_, fy_, q.d, px2, py2)) { px1 = px2; py1 = py2; }
            }
            float cs = q.cs, ws = q.ws;
            lean_rest<NMAX>(I, K, P, epi, fx_, fy_, q.d, px1, py1, cs, ws);
            const float c = lean_final(cs, ws);
            if (c == c)  // costs are >= 0: their bit patterns order like the values
                atomicMin(&s_best[t], ((unsigned long long)__float_as_uint(fmaxf(c, 0.f)) << 32) | (unsigned)k);
        }
        __syncthreads();
        // 4. the cheapest survivor against the running best
        const unsigned long long key = s_best[tid];
        if (key != ~0ull) {
            const float c = __uint_as_float((unsigned)(key >> 32));
            if (c < c_best) { c_best = c; d_best = sample_depth(pi, epoch0 + (uint32_t)(it + (int)(key & 0xffu)), range_factor); }
        }
        __syncthreads();
    }
    if (live) { I.depth[pi] = d_best; I.cost[pi] = c_best; }
}

// Verification aid (vk_set_cost_rand_plain): the sample pass in the literal order of optimize_depth.cu:269-284 on the fast arithmetic --
// incumbent, then every random depth evaluated in full and taken if strictly cheaper -- no early rejection, no queue.  k_cost_rand_q must
// give the same depth and cost maps bit for bit: its rejection bound is exact and its winner is the cheapest sample, the earliest among
// equals (tests/test_gpu_kernels.py::test_sample_pass_equals_the_plain_sequential_form).
template <int NMAX>
__global__ __launch_bounds__(256) static void k_cost_rand_plain(Img I, int n_rand, uint32_t epoch0, float range_factor) {
    if (!clamp_active(I)) return;
    const int tile = xcd_band_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int xi = (tile % gridDim.x) * 64 + (threadIdx.x & 63), yi = (tile / gridDim.x) * 4 + (threadIdx.x >> 6);
    if (xi >= I.w || yi >= I.h) return;
    const int pi = yi * I.w + xi;
    const LeanK K = lean_consts(I);
    float d_best = I.depth[pi], c_best = pixel_cost_lean<NMAX>(I, K, xi, yi, d_best);
    for (int k = 0; k < n_rand; k++) {
        const float d = sample_depth(pi, epoch0 + (uint32_t)k, range_factor);
        const float c = pixel_cost_lean<NMAX>(I, K, xi, yi, d);
        if (c < c_best) { c_best = c; d_best = d; }
    }
    I.depth[pi] = d_best; I.cost[pi] = c_best;
}

template <int NMAX>
__device__ __forceinline__ static void try_depth_lean(const Img& I, const LeanK& K, int x, int y, float cand) {
    const int pi = y * I.w + x;
    const float c = pixel_cost_lean<NMAX>(I, K, x, y, cand);
    if (c < I.cost[pi]) { I.depth[pi] = cand; I.cost[pi] = c; }
}
template <int NMAX>
__global__ __launch_bounds__(256) static void k_global_prop_sites_lean(Img I, int dir, int step, int nsites) {
    if (!clamp_active(I)) return;
    const int tile = xcd_band_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    // the lanes of a wave run along x either way: along the sites of a row (row passes: grid = site blocks x rows), along the columns
    // of one site row (column passes: grid = column blocks x site rows -- consecutive lanes on consecutive pixels)
    const bool rowpass = dir == 0 || dir == 2;
    const int a = (tile % gridDim.x) * blockDim.x + threadIdx.x, b = tile / gridDim.x;
    const int s = rowpass ? a : b, l = rowpass ? b : a;
    if (s >= nsites || l >= (rowpass ? I.h : I.w)) return;
    const LeanK K = lean_consts(I);
    if (dir == 0) { int x = 1 + s * step; try_depth_lean<NMAX>(I, K, x, l, I.depth[l * I.w + x - 1]); }
    else if (dir == 2) { int x = I.w - 2 - s * step; try_depth_lean<NMAX>(I, K, x, l, I.depth[l * I.w + x + 1]); }
    else if (dir == 1) { int y = 1 + s * step; try_depth_lean<NMAX>(I, K, l, y, I.depth[(y - 1) * I.w + l]); }
    else { int y = I.h - 2 - s * step; try_depth_lean<NMAX>(I, K, l, y, I.depth[(y + 1) * I.w + l]); }
}
template <int NMAX, bool STRICT = false>
__global__ __launch_bounds__(256) static void k_local_table_lean(Img I, int dir, int width, float* __restrict__ tbl) {
    if (!clamp_active(I)) return;
    const int tile = xcd_band_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int x = (tile % gridDim.x) * 64 + (threadIdx.x & 63), y = (tile / gridDim.x) * 4 + (threadIdx.x >> 6);
    if (x >= I.w || y >= I.h) return;
    const int w = I.w, h = I.h;
    bool member; int nb;
    if (dir == 0) { member = x >= 1 && (x % width) != 0; nb = y * w + x - 1; }
    else if (dir == 2) { member = x <= w - 2 && (x % width) != width - 1; nb = y * w + x + 1; }
    else if (dir == 1) { member = y >= 1 && (y % width) != 0; nb = (y - 1) * w + x; }
    else { member = y <= h - 2 && (y % width) != width - 1; nb = (y + 1) * w + x; }
    if (!member) return;
    tbl[y * w + x] = pixel_cost_any<NMAX, STRICT>(I, lean_consts(I), x, y, I.depth[nb]);
}
// Lane-split evaluation for k_local_runs_lean: LPP (4 or 8) lanes per pixel, lane g owns frames g, g+LPP, .. and priors g, g+LPP, ..  Every lane walks
// the (cheap) chain of positions, evaluates the gathers and residuals of its own frames, and the terms are combined in exactly the order
// of lean_head / lean_rest -- frame 0, priors, frames 1.. in log2 units, one scale by ln 2 -- so the value has the bits pixel_cost_lean
// gives for the same pixel and depth.  The combination reads the owner's term with a group broadcast (LPP = 4: DPP quad_perm, a VALU move,
// no trip through the LDS crossbar) and every lane of the group accumulates the same chain.  LPP = 4 halves the rounds a chain needs (twice
// the pixels per round) and suits up to 8 frames (two per lane); beyond that the per-lane work doubles again and 8 lanes per pixel win.
__device__ __forceinline__ float quad_bcast(float v, int q) {  // quad_perm [q,q,q,q]; q is a constant after unrolling
    switch (q & 3) { case 0: return dpp_mov<0x00>(v); case 1: return dpp_mov<0x55>(v); case 2: return dpp_mov<0xAA>(v); default: return dpp_mov<0xFF>(v); }
}
__device__ __forceinline__ bool quad_bcast(bool b, int q) { return quad_bcast(b ? 1.f : 0.f, q) != 0.f; }
__device__ __forceinline__ float pair_bcast(float v, int q) { return (q & 1) ? dpp_mov<0xF5>(v) : dpp_mov<0xA0>(v); }  // quad_perm [q,q,2+q,2+q]
// lane q of every aligned group of LPP lanes: a DPP move for pairs and quads, the LDS crossbar for groups of eight
template <int LPP> __device__ __forceinline__ float group_bcast(float v, int q) { return LPP == 2 ? pair_bcast(v, q) : LPP == 4 ? quad_bcast(v, q) : __shfl(v, q, LPP); }
template <int LPP> __device__ __forceinline__ bool group_bcast(bool b, int q) { return group_bcast<LPP>(b ? 1.f : 0.f, q) != 0.f; }
template <int NMAX, int LPP>
__device__ __forceinline__ static float cost_split_lean(const Img& I, const LeanK& K, int px, int py, float depth, int g) {
#pragma clang fp contract(off)
    constexpr int S = (NMAX + LPP - 1) / LPP;
    const int w = I.w, h = I.h, npx = w * h, pi = py * w + px;
    const PoseBlock* P = I.P;
    const float x = (float)px, y = (float)py, fw = (float)w, fh = (float)h;
    float qx[S], qy[S], ex[S], ey[S];
    bool vv[S];
#pragma unroll
    for (int k = 0; k < S; k++) { qx[k] = 0.f; qy[k] = 0.f; ex[k] = 0.f; ey[k] = 0.f; vv[k] = false; }
    {
        float px1 = x, py1 = y;
#pragma unroll
        for (int f = 0; f < NMAX; f++) {
            if (f < I.N) {  // uniform trip count: no divergence inside the quad
                float px2, py2;
                const bool zok = lean_step(P, f, x, y, depth, px2, py2);
                const bool valid = f == 0 ? zok : (zok && px1 >= 0.f && px1 < fw && py1 >= 0.f && py1 < fh);
                const bool mine = f % LPP == g;  // selects, no divergent branches (see lean_rest)
                vv[f / LPP] = mine ? valid : vv[f / LPP]; qx[f / LPP] = mine ? px1 : qx[f / LPP]; qy[f / LPP] = mine ? py1 : qy[f / LPP];
                ex[f / LPP] = mine ? px2 - px1 : ex[f / LPP]; ey[f / LPP] = mine ? py2 - py1 : ey[f / LPP];
                px1 = valid ? px2 : px1; py1 = valid ? py2 : py1;
            }
        }
    }
    // own terms: frame g + LPP k -> (weight, log2(1 + ratio)); the owner of frame 0 applies ln 2 and the weight like lean_head.
    // All gathers first (an unused slot reads texel (0,0) of its layer, or of layer 0), then the model.
    float2 ob[S];
    float wt[S], lt[S];
#pragma unroll
    for (int k = 0; k < S; k++) {
        const int f = g + LPP * k, fl = f < I.N ? f : 0;
        const bool use = f < I.N && vv[k];
        ob[k] = (k == 0 && g == 0) ? I.flows[pi] : bilinear2_inside(I.flows + (size_t)fl * npx, w, h, use ? qx[k] : 0.f, use ? qy[k] : 0.f);
        wt[k] = I.rig[(size_t)fl * npx + pi];
    }
#pragma unroll
    for (int k = 0; k < S; k++) {
        const int f = g + LPP * k;
        const bool use = f < I.N && vv[k];
        const ObsTerms T = obs_terms(ob[k].x, ob[k].y, K.ia2, K.l2q);
        float l = fast_log2(1.f + obs_ratio(T, ex[k] - ob[k].x, ey[k] - ob[k].y, K.qia2));
        if (k == 0 && g == 0) l = wt[k] * (0.6931471805599453f * l);  // = the cs lean_head starts from
        lt[k] = use ? l : 0.f; wt[k] = use ? wt[k] : 0.f; vv[k] = use;
    }
    // combine in the order of lean_head / lean_rest
    float cs = 0.f, ws = 0.f;
    if (I.N > 0 && group_bcast<LPP>(vv[0], 0)) { cs = group_bcast<LPP>(lt[0], 0); ws = group_bcast<LPP>(wt[0], 0); }
    // depth priors, LPP at a time: lane g evaluates prior f0 + g, the group adds them up in order (no per-slot arrays: there are up to 16)
#pragma unroll 1
    for (int f0 = 0; f0 < I.N_dp; f0 += LPP) {
        float pw = 0.f, pt = 0.f;
        bool pk = false;
        if (f0 + g < I.N_dp) pk = prior_parts(I, P, f0 + g, x, y, depth, pw, pt);
#pragma unroll
        for (int q = 0; q < LPP; q++) {
            const bool ok = group_bcast<LPP>(pk, q);
            const float wg = group_bcast<LPP>(pw, q), term = group_bcast<LPP>(pt, q);
            if (f0 + q < I.N_dp && ok) { cs = fmaf(wg, term, cs); ws += wg; }
        }
    }
    float cl = 0.f;
#pragma unroll
    for (int f = 1; f < NMAX; f++) {
        if (f < I.N) {
            const bool ok = group_bcast<LPP>(vv[f / LPP], f % LPP);
            const float wg = group_bcast<LPP>(wt[f / LPP], f % LPP), l = group_bcast<LPP>(lt[f / LPP], f % LPP);
            if (ok) { cl = fmaf(wg, l, cl); ws += wg; }
        }
    }
    cs = fmaf(0.6931471805599453f, cl, cs);
    return lean_final(cs, ws);
}

// ---- STRICT arithmetic on the fast launch structures (round 4) -----------------------------------------------------------------------
// The identity tests of the fast kernels (survivor queue vs plain loop, planned runs vs step-by-step chain, lanes per site vs one lane)
// show that those STRUCTURES change no result; what separates a fast kernel from a strict one is arithmetic only.  So strict mode runs
// on the same structures with pixel_cost_strict's operation sequence: 22 of the 100 ms of a strict cfg2 window were one lane per
// 31-step chain each evaluating 5 frames x 7 software transcendentals in fp64 one after the other (k_local_serial<., true>).
// Lane-split form of pixel_cost_strict: every lane of the group walks the rigid chain (the un-fused reference geometry), lane g owns
// frames g, g + LPP, .. and priors g, g + LPP, ..: gathers, strict::rigidness and the product weight * logf(rigidness) (the rounded
// product cost_acc subtracts); the group then replays cost_sum - term / wsum + weight in the order of pixel_cost_strict: frames, then
// priors.  Same bits as the one-lane evaluation.
// the weight of prior f for a hypothesis (and what its term is taken from): geometry and three gathers, no transcendental
__device__ __forceinline__ static bool prior_weight_strict(const Img& I, const PoseBlock* P, int f, int px, int py, float depth, float& wg, float& qz, float& td) {
#pragma clang fp contract(off)
    const int w = I.w, h = I.h;
    wg = 0.f; qz = 0.f; td = 0.f;
    P3 q = transform(P->dpRs[f], P->dpts[f], backproject(P, (float)px, (float)py, depth));
    float qx2, qy2;
    project(P, q, qx2, qy2);
    if (!(q.z > 0.f && qx2 >= 0.f && qx2 < (float)w && qy2 >= 0.f && qy2 < (float)h)) return false;
    td = fetch_prior_strict(I, I.priors, f, qx2, qy2);
    if (!(td > 0.f)) return false;
    const float tpc = fetch_prior_strict(I, I.pconfs, f, qx2, qy2);
    const float tc = fetch_prior_strict(I, I.confs, f, qx2, qy2);
    wg = tpc * tc * ((I.disp_delta > 0.f && f == 0) ? I.disp_delta : I.delta);
    qz = q.z;
    return true;
}
__device__ __forceinline__ static bool prior_parts_strict(const Img& I, const PoseBlock* P, int f, int px, int py, float depth, float& wg, float& term) {
#pragma clang fp contract(off)
    float qz, td;
    term = 0.f;
    if (!prior_weight_strict(I, P, f, px, py, depth, wg, qz, td)) return false;
    term = wg * vsm_logf(strict::depth_rigidness(qz, td, I.basefocal, I.omega, I.arf));  // the product cost_acc subtracts (fun_depth_cost, residual_model.h:64-68)
    return true;
}
template <int NMAX, int LPP>
__device__ __forceinline__ static float cost_split_strict(const Img& I, int px, int py, float depth, int g) {
#pragma clang fp contract(off)
    constexpr int S = (NMAX + LPP - 1) / LPP;
    const int w = I.w, h = I.h, npx = w * h, pi = py * w + px;
    const PoseBlock* P = I.P;
    float qx[S], qy[S], rdx[S], rdy[S];
    bool vv[S];
#pragma unroll
    for (int k = 0; k < S; k++) { qx[k] = 0.f; qy[k] = 0.f; rdx[k] = 0.f; rdy[k] = 0.f; vv[k] = false; }
    {
        P3 o = backproject(P, (float)px, (float)py, depth);
        float px1 = (float)px, py1 = (float)py;
#pragma unroll
        for (int f = 0; f < NMAX; f++) {
            if (f < I.N) {
                o = transform(P->Rs[f], P->ts[f], o);
                float px2, py2;
                project(P, o, px2, py2);
                const bool valid = o.z > 0.f && px1 >= 0.f && px1 < (float)w && py1 >= 0.f && py1 < (float)h;
                const bool mine = f % LPP == g;
                vv[f / LPP] = mine ? valid : vv[f / LPP]; qx[f / LPP] = (mine && valid) ? px1 : qx[f / LPP]; qy[f / LPP] = (mine && valid) ? py1 : qy[f / LPP];
                rdx[f / LPP] = (mine && valid) ? px2 - px1 : rdx[f / LPP]; rdy[f / LPP] = (mine && valid) ? py2 - py1 : rdy[f / LPP];
                px1 = valid ? px2 : px1; py1 = valid ? py2 : py1;  // advances on contributing frames only (:162-164)
            }
        }
    }
    float tm[S], wt[S];
#pragma unroll
    for (int k = 0; k < S; k++) {
        const int f = g + LPP * k, fl = f < I.N ? f : 0;
        float2 ob = make_float2(0.f, 0.f);
        wt[k] = 0.f;
        if (f < I.N) {  // (a slot without a frame fetches nothing: with N = 0 -- depth priors only -- there may be no flow layer at all)
            ob = (f == 0) ? I.flows[pi] : fetch_flow_strict(I, fl, qx[k], qy[k]);
            wt[k] = I.rig[(size_t)fl * npx + pi];
        }
        tm[k] = wt[k] * vsm_logf(strict::rigidness(rdx[k], rdy[k], ob.x, ob.y, I.lambda, I.arf));
        vv[k] = vv[k] && f < I.N;
    }
    float cost_sum = 0.f, wsum = 0.f;
#pragma unroll
    for (int f = 0; f < NMAX; f++) {
        if (f < I.N) {
            const bool ok = group_bcast<LPP>(vv[f / LPP], f % LPP);
            const float t = group_bcast<LPP>(tm[f / LPP], f % LPP), wg = group_bcast<LPP>(wt[f / LPP], f % LPP);
            if (ok) { cost_sum = cost_sum - t; wsum += wg; }
        }
    }
#pragma unroll 1
    for (int f0 = 0; f0 < I.N_dp; f0 += LPP) {
        float pw = 0.f, pt = 0.f;
        bool pk = false;
        if (f0 + g < I.N_dp) pk = prior_parts_strict(I, P, f0 + g, px, py, depth, pw, pt);
#pragma unroll
        for (int q = 0; q < LPP; q++) {
            const bool ok = group_bcast<LPP>(pk, q);
            const float wg = group_bcast<LPP>(pw, q), term = group_bcast<LPP>(pt, q);
            if (f0 + q < I.N_dp && ok) { cost_sum = cost_sum - term; wsum += wg; }
        }
    }
    if (wsum == 0.f) return INFINITY;
    return cost_sum / fmaxf(wsum, 1.1920929e-07f);
}
template <int NMAX, int LPP, bool STRICT>
__device__ __forceinline__ static float cost_split_any(const Img& I, const LeanK& K, int px, int py, float depth, int g) {
    if constexpr (STRICT) return cost_split_strict<NMAX, LPP>(I, px, py, depth, g); else return cost_split_lean<NMAX, LPP>(I, K, px, py, depth, g);
}

// The sample pass in strict arithmetic: exact PROGRESSIVE rejection over a survivor queue.  The strict cost is
// fl(cost_sum / max(wsum, eps)) with cost_sum = ((0 - t_0) - t_1 ..) - priors.., every t = weight * logf(rigidness) <= 0 and every weight >= 0,
// both sums taken left to right (pixel_cost_strict).  Rounded addition is monotone in each operand, so dropping terms from the numerator
// chain can only lower it and adding terms to the denominator chain can only raise it: after frames 0 .. f of a hypothesis
//   cost >= fl(L_f / max(U_f, eps)),  L_f = the chain over frames 0 .. f (the true prefix) and the priors,
//                                     U_f = the chain over the contributing frames of 0 .. f, ALL of frames f+1 .. and the contributing priors
// -- an EXACT bound in the reference's own rounding (no margin).  A sample whose bound is >= the incumbent's cost cannot win the
// `cost < best` test (optimize_depth.cu:201-207).  A strict frame is ~1100 fp64-heavy instructions and the pass is bound by their issue,
// so the bound is applied after EVERY frame: the samples still alive sit in an LDS queue (pixel, sample, partial sums), each stage
// evaluates frame f of the queue's entries dense -- one entry per lane, whichever lane -- and re-compacts the survivors; a typical random
// depth dies after one or two frames instead of N.  The chain of positions is re-walked per stage from the sample's depth (the
// un-fused geometry, ~60 instructions per frame: the same values every time).  NaNs fail the >= and stay alive to the end.  A sample that
// survives all frames carries exactly pixel_cost_strict's sums; the winner rule is k_cost_rand_q's (cheapest, earliest among equals).
// Up to one depth prior rides along in the entry (its term and weight are needed at the end of the chain and in every bound); with more
// priors the launcher takes the plain kernel.
struct CrqsEntry { unsigned id; float cs, ws, tp, wp; };  // id = lane-in-workgroup | sample << 8 | prior contributes << 16
template <int NMAX>
__global__ __launch_bounds__(256) static void k_cost_rand_q_strict(Img I, int n_rand, uint32_t epoch0, float range_factor) {
#pragma clang fp contract(off)
    __shared__ unsigned long long s_best[256];
    __shared__ float s_cbest[256];
    __shared__ CrqsEntry s_q[256 * CRQ_NS];
    __shared__ float s_d[CRQ_NS][256];  // the random depths of this round: whichever lane evaluates an entry reads the pixel's sample here
    __shared__ int s_qn, s_qw;
    if (!clamp_active(I)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int tile = xcd_band_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int x0 = (tile % gridDim.x) * 64, y0 = (tile / gridDim.x) * 4;
    const int xi = x0 + lane, yi = y0 + (tid >> 6);
    const bool live = xi < I.w && yi < I.h;
    const int w = I.w, h = I.h, npx = w * h, px = live ? xi : 0, py = live ? yi : 0, pi = py * w + px;
    const PoseBlock* P = I.P;
    const float2 o0 = I.N > 0 ? I.flows[pi] : make_float2(0.f, 0.f);
    const float wgt0 = I.N > 0 ? I.rig[pi] : 0.f;
    const strict::RigObs ob0 = strict::rigidness_obs(o0.x, o0.y, I.lambda, I.arf);  // frame 0 observes at the pixel: shared by the ten samples
    float wall = 0.f;  // weights of frames 1.. (the denominator bound of stage 0 adds them one by one: same chain as below)
    (void)wall;
    float d_best = I.depth[pi], c_best = pixel_cost_strict<NMAX>(I, px, py, d_best);
    vrc_xorwow st;
    if (I.xw) st = I.xw[pi];  // --reference_rng 1: this pixel's cuRAND state (optimize_depth.cu:273), one draw per sample, stored back at the end
    // bound of an entry whose sums stand after frame f: dead iff it cannot beat the pixel's incumbent
    auto dead_after = [&](int epi, int f, float cs, float ws, bool pk, float tp, float wp, float cb) {
        float U = ws;
        for (int g = f + 1; g < I.N; g++) U += I.rig[(size_t)g * npx + epi];
        float L = cs;
        if (pk) { L = L - tp; U += wp; }
        return U == 0.f || (L / fmaxf(U, 1.1920929e-07f)) >= cb;
    };
    for (int it = 0; it < n_rand; it += CRQ_NS) {
        const int nh = min(CRQ_NS, n_rand - it);
        s_best[tid] = ~0ull;
        s_cbest[tid] = c_best;
        if (tid == 0) s_qn = 0;
        for (int k = 0; k < nh; k++)
            s_d[k][tid] = I.xw ? 1.0f / (range_factor * vrc_uniform(vrc_xorwow_next(&st)) + (1.0f / 1e5f)) : sample_depth(pi, epoch0 + (uint32_t)(it + k), range_factor);
        __syncthreads();
        // ---- stage 0: frame 0 (observed at the pixel itself) and the prior of every sample of this round, by the pixel's own lane
        for (int k = 0; k < nh; k++) {
            const float d = s_d[k][tid];
            float cs = 0.f, ws = 0.f, tp = 0.f, wp = 0.f;
            bool pk = false;
            if (I.N > 0) {
                const P3 o = transform(P->Rs[0], P->ts[0], backproject(P, (float)px, (float)py, d));
                float px2, py2;
                project(P, o, px2, py2);
                if (o.z > 0.f) {  // the pixel itself is inside the image
                    cs = cs - wgt0 * vsm_logf(strict::rigidness_with(ob0, px2 - (float)px, py2 - (float)py, I.arf));
                    ws += wgt0;
                }
            }
            if (I.N_dp > 0) pk = prior_parts_strict(I, P, 0, px, py, d, wp, tp);
            const bool alive = live && !dead_after(pi, 0, cs, ws, pk, tp, wp, c_best);
            const unsigned long long m = __ballot(alive);
            int base = 0;
            if (lane == 0 && m) base = atomicAdd(&s_qn, __popcll(m));
            base = __shfl(base, 0, 64);
            if (alive) s_q[base + __popcll(m & ((1ull << lane) - 1ull))] = { (unsigned)tid | ((unsigned)k << 8) | (pk ? 1u << 16 : 0u), cs, ws, tp, wp };
        }
        __syncthreads();
        // ---- stages 1 .. N-1: frame f of every entry still alive; survivors re-compacted in place (an entry is read before the barrier
        // that precedes the writes of its batch, and writes only go to slots at or below the batch that was just read)
        for (int f = 1; f < I.N; f++) {
            const int qn = s_qn;
            if (qn == 0) break;
            if (tid == 0) s_qw = 0;
            __syncthreads();
            for (int e0 = 0; e0 < qn; e0 += 256) {
                const int e = e0 + tid;
                const bool have = e < qn;
                CrqsEntry q = { 0u, 0.f, 0.f, 0.f, 0.f };
                if (have) q = s_q[e];
                const int t = (int)(q.id & 255u), k = (int)((q.id >> 8) & 255u);
                const bool pk = (q.id >> 16) & 1u;
                const int ex_ = x0 + (t & 63), ey_ = y0 + (t >> 6), epi = have ? ey_ * w + ex_ : 0;
                bool alive = false;
                if (have) {
                    const float d = s_d[k][t];
                    // the chain of positions up to frame f (pixel_cost_strict's own walk: positions advance on contributing frames only)
                    P3 o = backproject(P, (float)ex_, (float)ey_, d);
                    float px1 = (float)ex_, py1 = (float)ey_, px2 = 0.f, py2 = 0.f;
                    bool valid = false;
                    for (int g = 0; g <= f; g++) {
                        o = transform(P->Rs[g], P->ts[g], o);
                        project(P, o, px2, py2);
                        valid = o.z > 0.f && px1 >= 0.f && px1 < (float)w && py1 >= 0.f && py1 < (float)h;
                        if (g < f && valid) { px1 = px2; py1 = py2; }
                    }
                    if (valid) {
                        const float2 obs = fetch_flow_strict(I, f, px1, py1);
                        const float wg = I.rig[(size_t)f * npx + epi];
                        q.cs = q.cs - wg * vsm_logf(strict::rigidness(px2 - px1, py2 - py1, obs.x, obs.y, I.lambda, I.arf));
                        q.ws += wg;
                    }
                    alive = (f == I.N - 1) || !dead_after(epi, f, q.cs, q.ws, pk, q.tp, q.wp, s_cbest[t]);
                }
                __syncthreads();  // every entry of this batch has been read
                const unsigned long long m = __ballot(alive);
                int base = 0;
                if (lane == 0 && m) base = atomicAdd(&s_qw, __popcll(m));
                base = __shfl(base, 0, 64);
                if (alive) s_q[base + __popcll(m & ((1ull << lane) - 1ull))] = q;
                __syncthreads();
            }
            if (tid == 0) s_qn = s_qw;
            __syncthreads();
        }
        // ---- the entries that lived through every frame: the prior closes the chain, the cheapest sample of a pixel wins
        {
            const int qn = s_qn;
            for (int e = tid; e < qn; e += 256) {
                const CrqsEntry q = s_q[e];
                const int t = (int)(q.id & 255u), k = (int)((q.id >> 8) & 255u);
                float cs = q.cs, ws = q.ws;
                if ((q.id >> 16) & 1u) { cs = cs - q.tp; ws += q.wp; }
                const float c = ws == 0.f ? INFINITY : cs / fmaxf(ws, 1.1920929e-07f);
                if (c == c) atomicMin(&s_best[t], ((unsigned long long)__float_as_uint(fmaxf(c, 0.f)) << 32) | (unsigned)k);
            }
        }
        __syncthreads();
        const unsigned long long key = s_best[tid];
        if (key != ~0ull) {
            const float c = __uint_as_float((unsigned)(key >> 32));
            if (c < c_best) { c_best = c; d_best = s_d[key & 0xffu][tid]; }
        }
        __syncthreads();
    }
    if (live) { I.depth[pi] = d_best; I.cost[pi] = c_best; if (I.xw) I.xw[pi] = st; }
}

// ---- the strict sample pass behind an fp32 filter (round 6) --------------------------------------------------------------------------
// A random depth only matters if its cost is BELOW the pixel's current cost (`cost < best`, optimize_depth.cu:201-207); 98-99 % of the
// samples are evaluated to be thrown away, and in strict arithmetic an evaluation is ~900 fp64-heavy instructions per frame.  So each
// sample is first walked on the hardware transcendentals:
//   * the SAME positions and observations as the strict evaluation -- the un-fused chain (backproject / transform / project) and
//     fetch_flow_strict, bit for bit, so the two evaluations of a frame differ in the residual model alone: -logf(rigidness) from
//     strict::rigidness on vsm_* against ln2 log2(1 + obs_ratio) on v_log_f32 / v_exp_f32, two approximations of one real function of the
//     same four floats.  Analytic bound of their difference: arguments of magnitude <= 60 in the log2 domain carry 1 ulp each, exponents
//     c + 1 <= 2: <= 1e-5 absolute + 1e-5 relative.  MEASURED over the whole input range (tests/test_gpu_strict_filter.py): 2.5 % of
//     SF_ABS + SF_REL * value (vk_device.hpp), i.e. the margin is 40 times the largest difference seen;
//   * a lower bound of the strict numerator chain L from the filter's chain Lf:  L >= Lf (1 - SF_REL) - SF_ABS * (weights so far), every term
//     being weight * (-log rigidness) >= 0 (the depth priors' terms come first: same margin, fun_depth_rigidness on the strict q.z and prior depth); an upper bound U of the strict
//     denominator: the chain over ALL frames' weights and the contributing priors' weights (their exact values: geometry and gathers only)
//     -- monotone rounding, as for k_cost_rand_q_strict;
//   * discarded iff  Lf (1 - SF_REL) - SF_ABS ws >= best * max(U, eps)  -- then fl(L / max(wsum, eps)) >= best in the reference's own rounding.
//     A NaN anywhere fails the >= and the sample stays.
// What is left (the winners, near-ties, the unlucky: 1-2 % without priors) is evaluated in strict arithmetic with one (sample, term) per lane
// -- frames and priors side by side, the chain of positions re-walked by each lane -- and summed per sample in pixel_cost_strict's order: its
// bits.  The filter never changes a result, only which losers are looked at closely (identity: vk_debug_switch "strict_filter" 0, and every
// strict test against the oracle).
// (SF_REL / SF_ABS and filt_neglog: vk_device.hpp, next to obs_ratio)
// what the depth priors add to a sample at depth d: their exact weights to the denominator bound U (wall = the chain over all frames' weights at the pixel) and, on the
// hardware transcendentals, their terms to the filter's numerator (fun_depth_rigidness's inputs qz, td are the strict ones: same margin as for a frame)
struct FiltChain { P3 o; float px1, py1, Lf, ws; };
__device__ __forceinline__ static float filt_begin(FiltChain& C, const Img& I, const PoseBlock* P, int px, int py, float d, float wall) {
#pragma clang fp contract(off)
    C.o = backproject(P, (float)px, (float)py, d); C.px1 = (float)px; C.py1 = (float)py; C.Lf = 0.f; C.ws = 0.f;
    float U = wall;
    for (int q = 0; q < I.N_dp; q++) {
        float wg, qz, td;
        if (prior_weight_strict(I, P, q, px, py, d, wg, qz, td)) {
            U += wg;
            C.Lf += wg * (0.6931471805599453f * fast_log2(1.f + depth_ratio(qz, td, I.basefocal, I.omega, I.inv_arf)));
            C.ws += wg;
        }
    }
    return U;
}
// frame f of the chain in three steps: geometry (-> does the frame contribute, where is it observed), the gather (unconditional, at the pixel itself where the
// frame does not contribute), the model; filt_model returns "cannot win against best"
struct FiltStep { float px2, py2; bool valid; };
__device__ __forceinline__ static FiltStep filt_geom(FiltChain& C, const Img& I, const PoseBlock* P, int f) {
#pragma clang fp contract(off)
    FiltStep S;
    C.o = transform(P->Rs[f], P->ts[f], C.o);
    project(P, C.o, S.px2, S.py2);
    S.valid = C.o.z > 0.f && C.px1 >= 0.f && C.px1 < (float)I.w && C.py1 >= 0.f && C.py1 < (float)I.h;
    return S;
}
__device__ __forceinline__ static bool filt_model(FiltChain& C, const FiltStep& S, const LeanK& K, float2 ob, float wg, const ObsTerms& T, float U, float best) {
#pragma clang fp contract(off)
    if (S.valid) {
        C.Lf += wg * filt_neglog(T, S.px2 - C.px1, S.py2 - C.py1, ob.x, ob.y, K.qia2);
        C.ws += wg;
        C.px1 = S.px2; C.py1 = S.py2;
    }
    return U == 0.f || (C.Lf * (1.f - SF_REL) - SF_ABS * C.ws) >= best * fmaxf(U, 1.1920929e-07f);
}
// term j of hypothesis (ex, ey, d) in strict arithmetic: frame j < N (the chain re-walked up to it: pixel_cost_strict's own walk), prior j - N beyond
struct SfTerm { float t, w; bool ok; };
__device__ __forceinline__ static SfTerm strict_term(const Img& I, const PoseBlock* P, int ex, int ey, float d, int j) {
#pragma clang fp contract(off)
    SfTerm r = { 0.f, 0.f, false };
    float mag = 0.f, diff = 0.f, kstr = 0.f;  // the lanes of a wave hold frames and priors: their ways part here and meet again in ONE evaluation of the model (strict::rig_core)
    if (j < I.N) {
        const int w = I.w, h = I.h;
        P3 o = backproject(P, (float)ex, (float)ey, d);
        float px1 = (float)ex, py1 = (float)ey, px2 = 0.f, py2 = 0.f;
        bool valid = false;
        for (int g = 0; g <= j; g++) {
            o = transform(P->Rs[g], P->ts[g], o);
            project(P, o, px2, py2);
            valid = o.z > 0.f && px1 >= 0.f && px1 < (float)w && py1 >= 0.f && py1 < (float)h;
            if (g < j && valid) { px1 = px2; py1 = py2; }
        }
        if (valid) {
            const int epi = ey * w + ex;
            const float2 ob = j == 0 ? I.flows[epi] : fetch_flow_strict(I, j, px1, py1);
            r.w = I.rig[(size_t)j * w * h + epi];
            r.ok = true;
            // strict::rigidness (fun_rigidness, residual_model.h:34-42) up to its call of rig_core
            const float dx1 = px2 - px1, dy1 = py2 - py1;
            mag = sqrtf(ob.x * ob.x + ob.y * ob.y) / I.arf;
            const float fx = dx1 - ob.x, fy = dy1 - ob.y;
            diff = sqrtf(fx * fx + fy * fy) / I.arf;
            kstr = I.lambda;
        }
    } else {
        float qz, td;
        r.ok = prior_weight_strict(I, P, j - I.N, ex, ey, d, r.w, qz, td);
        if (r.ok) {  // strict::depth_rigidness (fun_depth_rigidness, :51-61) up to its call of rig_core
            const float disp1 = (I.basefocal / qz) / I.arf, disp2 = (I.basefocal / td) / I.arf;
            mag = disp2; diff = fabsf(disp1 - disp2); kstr = I.omega;
        }
    }
    if (r.ok) r.t = r.w * vsm_logf(strict::rig_core(mag, diff, kstr));
    return r;
}
// the terms of one hypothesis, NT of them from s_t / s_w / s_v at `at`, in pixel_cost_strict's order
__device__ __forceinline__ static float strict_close(const float* s_t, const float* s_w, const unsigned char* s_v, int at, int NT) {
#pragma clang fp contract(off)
    float cs = 0.f, ws = 0.f;
    for (int j = 0; j < NT; j++)
        if (s_v[at + j]) { cs = cs - s_t[at + j]; ws += s_w[at + j]; }
    return ws == 0.f ? INFINITY : cs / fmaxf(ws, 1.1920929e-07f);
}

// The sample pass in two launches: the cost of the current depth in strict arithmetic (k_cost_strict: the registers of pixel_cost_strict's gathers-in-flight form stay out
// of the second kernel, 175 -> 123 VGPRs = 2 -> 4 waves per SIMD; 1241x376: 346 -> 291 us), then the random depths of a round through the filter and the survivors in
// strict arithmetic.  Any number of depth priors.  Launched with N + N_dp <= 64.  (Measured and left: two samples per lane walked side by side, gathers of both in
// flight -- 144 registers, 3 waves, 291 -> 346 us; five waves per SIMD at the price of 76 bytes of scratch -- 107 / 226 / 854 us against 119 / 219 / 890.)
template <int NMAX>
__global__ __launch_bounds__(256) static void k_cost_strict(Img I) {
    if (!clamp_active(I)) return;
    const int tile = xcd_band_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int x = (tile % gridDim.x) * 64 + (threadIdx.x & 63), y = (tile / gridDim.x) * 4 + (threadIdx.x >> 6);
    if (x < I.w && y < I.h) I.cost[y * I.w + x] = pixel_cost_strict<NMAX>(I, x, y, I.depth[y * I.w + x]);
}
template <int NMAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) static void k_cost_rand_f_strict(Img I, int n_rand, uint32_t epoch0, float range_factor) {
#pragma clang fp contract(off)
    __shared__ unsigned long long s_best[256];
    __shared__ unsigned short s_q[256 * CRQ_NS];  // lane-in-workgroup | sample << 8
    __shared__ float s_d[CRQ_NS][256];
    __shared__ float s_f0[3][CRQ_NS][256];  // what frame 0 left of a sample: U, Lf, ws
    __shared__ float s_t[256], s_w[256];
    __shared__ unsigned char s_v[256];
    __shared__ int s_qn;
    if (!clamp_active(I)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int tile = xcd_band_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int x0 = (tile % gridDim.x) * 64, y0 = (tile / gridDim.x) * 4;
    const int xi = x0 + lane, yi = y0 + (tid >> 6);
    const bool live = xi < I.w && yi < I.h;
    const int w = I.w, npx = w * I.h, px = live ? xi : 0, py = live ? yi : 0, pi = py * w + px;
    const PoseBlock* P = I.P;
    const LeanK K = lean_consts(I);
    const float2 o0 = I.N > 0 ? I.flows[pi] : make_float2(0.f, 0.f);
    const ObsTerms T0 = obs_terms(o0.x, o0.y, K.ia2, K.l2q);
    float wall = 0.f;
    for (int f = 0; f < I.N; f++) wall += I.rig[(size_t)f * npx + pi];
    float d_best = I.depth[pi], c_best = I.cost[pi];  // (k_cost_strict, the launch before this one)
    vrc_xorwow st;
    if (I.xw) st = I.xw[pi];
    const int NT = I.N + I.N_dp, per = 256 / NT;
    for (int it = 0; it < n_rand; it += CRQ_NS) {
        const int nh = min(CRQ_NS, n_rand - it);
        s_best[tid] = ~0ull;
        if (tid == 0) s_qn = 0;
        for (int k = 0; k < nh; k++)
            s_d[k][tid] = I.xw ? 1.0f / (range_factor * vrc_uniform(vrc_xorwow_next(&st)) + (1.0f / 1e5f)) : sample_depth(pi, epoch0 + (uint32_t)(it + k), range_factor);
        __syncthreads();
        // ---- the filter.  Measured (a step counter in a diagnostic build, profiles/r06_summary.md I): a sample lives 1.25 frames on average -- four out of five die on frame 0, which needs no
        // gather (the pixel's own texel, its ObsTerms shared by all samples) -- but the slowest lane of a wave walks 2.7 per sample.  So frame 0 of every sample first,
        // all lanes in step; then each lane walks what its pixel has left, one frame per trip, whichever sample it is at.
        unsigned surv = 0u;
        if (live) {
            if (I.N == 0) surv = (1u << nh) - 1u;
            else {
                const float wg0 = I.rig[pi];
                unsigned rest = 0u;
                for (int k = 0; k < nh; k++) {
                    FiltChain C;
                    const float U = filt_begin(C, I, P, px, py, s_d[k][tid], wall);
                    const FiltStep S = filt_geom(C, I, P, 0);
                    const bool dead = filt_model(C, S, K, o0, wg0, T0, U, c_best);
                    s_f0[0][k][tid] = U; s_f0[1][k][tid] = C.Lf; s_f0[2][k][tid] = C.ws;
                    if (!dead) { if (I.N == 1) surv |= 1u << k; else rest |= 1u << k; }
                }
                FiltChain C;
                float U = 0.f;
                int k = 0, f = 0;
                C.o = backproject(P, (float)px, (float)py, 1.f); C.px1 = (float)px; C.py1 = (float)py; C.Lf = 0.f; C.ws = 0.f;
                while (rest) {
                    if (f == 0) {  // the sample's frame 0 again, geometry only: where the chain stands, and the sums it left
                        k = __ffs((int)rest) - 1;
                        C.o = backproject(P, (float)px, (float)py, s_d[k][tid]); C.px1 = (float)px; C.py1 = (float)py;
                        const FiltStep S0 = filt_geom(C, I, P, 0);
                        if (S0.valid) { C.px1 = S0.px2; C.py1 = S0.py2; }
                        U = s_f0[0][k][tid]; C.Lf = s_f0[1][k][tid]; C.ws = s_f0[2][k][tid];
                        f = 1;
                    }
                    const FiltStep S = filt_geom(C, I, P, f);
                    const float2 ob = fetch_flow_strict(I, f, S.valid ? C.px1 : (float)px, S.valid ? C.py1 : (float)py);
                    const float wg = I.rig[(size_t)f * npx + pi];
                    const bool dead = filt_model(C, S, K, ob, wg, obs_terms(ob.x, ob.y, K.ia2, K.l2q), U, c_best);
                    if (dead || f == I.N - 1) { if (!dead) surv |= 1u << k; rest &= rest - 1u; f = 0; }
                    else f++;
                }
            }
        }
        for (int k = 0; k < nh; k++) {
            const bool alive = (surv >> k) & 1u;
            const unsigned long long m = __ballot(alive);
            int base = 0;
            if (lane == 0 && m) base = atomicAdd(&s_qn, __popcll(m));
            base = __shfl(base, 0, 64);
            if (alive) s_q[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(tid | (k << 8));
        }
        __syncthreads();
        // ---- the survivors in strict arithmetic: one (survivor, term) per lane
        const int qn = s_qn;
        if (I.sf) {
            const unsigned long long lv = __ballot(live);
            if (lane == 0) atomicAdd(I.sf + 0, (unsigned long long)(__popcll(lv) * nh));
            if (tid == 0) atomicAdd(I.sf + 1, (unsigned long long)qn);
        }
        for (int e0 = 0; e0 < qn; e0 += per) {
            const int el = tid / NT, j = tid - el * NT, e = e0 + el;
            const bool have = el < per && e < qn;
            SfTerm r = { 0.f, 0.f, false };
            int t = 0, k = 0;
            if (have) {
                const unsigned id = s_q[e];
                t = (int)(id & 255u); k = (int)(id >> 8);
                r = strict_term(I, P, x0 + (t & 63), y0 + (t >> 6), s_d[k][t], j);
            }
            s_t[tid] = r.t; s_w[tid] = r.w; s_v[tid] = r.ok ? 1 : 0;
            __syncthreads();
            if (have && j == 0) {
                const float c = strict_close(s_t, s_w, s_v, tid, NT);
                if (c == c) atomicMin(&s_best[t], ((unsigned long long)__float_as_uint(fmaxf(c, 0.f)) << 32) | (unsigned)k);
            }
            __syncthreads();
        }
        const unsigned long long key = s_best[tid];
        if (key != ~0ull) {
            const float c = __uint_as_float((unsigned)(key >> 32));
            if (c < c_best) { c_best = c; d_best = s_d[key & 0xffu][tid]; }
        }
        __syncthreads();
    }
    if (live) { I.depth[pi] = d_best; I.cost[pi] = c_best; if (I.xw) I.xw[pi] = st; }
}

// The table pass of the strict local propagation behind the same filter, in two launches (from 1 M pixels: optimize_depth_launch).  The runs kernel only asks `table < cost` and takes
// the table value when that holds, so (1) every chain pixel walks its predecessor's depth through the filter (fp32, 41 registers): a reject is written as +inf, anything else -- the
// accepts, 35-40 % of the entries, and the near-ties -- is appended to one of SFQ queues in global memory (one atomic per 64x4 tile: an atomic on one address takes ~90 ns and they
// queue up, so the tiles are spread over 256 addresses); (2) the queues are evaluated in strict arithmetic one entry per lane, densely packed.  WHICH queue decides whether that pays: a
// queue is one of 32 stretches of one XCD's band of the image (xcd_band_tile) and is walked by consecutive workgroups of THAT XCD, so the gathers of the strict evaluation find the flow
// layers in the L2 the plain kernel's would -- with the tiles dealt round-robin over the queues the packed kernel took as long as evaluating every entry (1920x1080: 479 us against 490;
// with the stretches: 245).  (Other forms, all bit-identical, profiles/r06_summary.md I: (entry, term) lanes inside the table kernel; flagged entries packed per tile into the tile's own waves.)
constexpr int SFQ = 256;
__device__ __forceinline__ static bool table_member(int w, int h, int dir, int width, int x, int y, int& nb) {
    if (dir == 0) { nb = y * w + x - 1; return x >= 1 && (x % width) != 0; }
    if (dir == 2) { nb = y * w + x + 1; return x <= w - 2 && (x % width) != width - 1; }
    if (dir == 1) { nb = (y - 1) * w + x; return y >= 1 && (y % width) != 0; }
    nb = (y + 1) * w + x; return y <= h - 2 && (y % width) != width - 1;
}
template <int NMAX>
__global__ __launch_bounds__(256) static void k_local_table_filter(Img I, int dir, int width, float* __restrict__ tbl, unsigned* __restrict__ qcnt, unsigned* __restrict__ qlist, int qcap) {
#pragma clang fp contract(off)
    __shared__ int s_n, s_base;
    if (!clamp_active(I)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    const int tile = xcd_band_tile(bid, gridDim.x * gridDim.y);
    const int x = (tile % gridDim.x) * 64 + lane, y = (tile / gridDim.x) * 4 + (tid >> 6);
    if (tid == 0) s_n = 0;
    __syncthreads();
    int nb = 0;
    const bool member = x < I.w && y < I.h && table_member(I.w, I.h, dir, width, x, y, nb);
    const int npx = I.w * I.h, pi = member ? y * I.w + x : 0;
    const PoseBlock* P = I.P;
    bool maybe = member;
    if (member && I.N > 0) {
        const float d = I.depth[nb], own = I.cost[pi];
        const LeanK K = lean_consts(I);
        const float2 o0 = I.flows[pi];
        const ObsTerms T0 = obs_terms(o0.x, o0.y, K.ia2, K.l2q);
        float wall = 0.f;
        for (int f = 0; f < I.N; f++) wall += I.rig[(size_t)f * npx + pi];
        FiltChain C;
        const float U = filt_begin(C, I, P, x, y, d, wall);
        bool dead = false;
        for (int f = 0; f < I.N && !dead; f++) {
            const FiltStep S = filt_geom(C, I, P, f);
            const bool far = S.valid && f > 0;
            const float2 ob = fetch_flow_strict(I, f, far ? C.px1 : (float)x, far ? C.py1 : (float)y);
            const float wg = I.rig[(size_t)f * npx + pi];
            dead = filt_model(C, S, K, f == 0 ? o0 : ob, wg, f == 0 ? T0 : obs_terms(ob.x, ob.y, K.ia2, K.l2q), U, own);
        }
        maybe = !dead;
        if (dead) tbl[pi] = INFINITY;
    }
    const unsigned long long m = __ballot(maybe);
    int wbase = 0;
    if (lane == 0 && m) wbase = atomicAdd(&s_n, __popcll(m));
    wbase = __shfl(wbase, 0, 64);
    __syncthreads();
    // the queue of this tile: workgroup bid runs on XCD bid % 8 and xcd_band_tile hands it tile bid / 8 of that XCD's band of the image; queue (band, j) collects the j-th of SFQ / 8 stretches
    // of the band, so that the strict kernel can walk a queue on the XCD whose L2 holds that stretch of the flow layers
    const int per = (int)(gridDim.x * gridDim.y) / 8;
    const int q = (bid % 8) + 8 * (bid < per * 8 ? min((bid / 8) * (SFQ / 8) / max(per, 1), SFQ / 8 - 1) : SFQ / 8 - 1);
    if (tid == 0 && s_n > 0) s_base = (int)atomicAdd(qcnt + q, (unsigned)s_n);
    __syncthreads();
    if (maybe) qlist[(size_t)q * qcap + s_base + wbase + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned)pi;
    if (I.sf && tid == 0) { atomicAdd(I.sf + 2, 1ull); atomicAdd(I.sf + 3, (unsigned long long)s_n); }
}
template <int NMAX>
__global__ __launch_bounds__(256) static void k_local_table_exact(Img I, int dir, float* __restrict__ tbl, const unsigned* __restrict__ qcnt, const unsigned* __restrict__ qlist, int qcap) {
    if (!clamp_active(I)) return;
    // workgroup B runs on XCD B % 8: it takes chunk (B / 8) % chunks of queue (band B % 8, stretch (B / 8) / chunks) -- consecutive workgroups of an XCD walk one stretch of its band
    const int chunks = qcap / 256, m = blockIdx.x / 8;
    const int q = (blockIdx.x % 8) + 8 * (m / chunks), e = (m % chunks) * 256 + threadIdx.x;
    if (e >= (int)qcnt[q]) return;
    const int pi = (int)qlist[(size_t)q * qcap + e], x = pi % I.w, y = pi / I.w;
    const int nb = dir == 0 ? pi - 1 : dir == 2 ? pi + 1 : dir == 1 ? pi - I.w : pi + I.w;
    tbl[pi] = pixel_cost_strict<NMAX>(I, x, y, I.depth[nb]);
}

// Global propagation with the candidate of a site evaluated by LPP lanes (cost_split_lean: the bits of pixel_cost_lean).  A pass has
// only w*h/step sites: with one lane per site it is a few hundred (640x480) to a few thousand (1080p) waves, each walking all frames
// of its 64 sites one after the other -- latency, 26 % VALU issue at 1080p.  LPP lanes per site = LPP times the waves, each lane with
// ceil(N / LPP) gathers and residuals in flight.  Same decisions, same maps as k_global_prop_sites_lean (vk_set_global_split).
template <int NMAX, int LPP, bool STRICT = false>
__global__ __launch_bounds__(64) static void k_global_prop_split_lean(Img I, int dir, int step, int nsites) {
    if (!clamp_active(I)) return;
    constexpr int SPW = 64 / LPP;  // sites per wave
    const int tile = xcd_band_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const bool rowpass = dir == 0 || dir == 2;
    const int g = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const int a = (tile % gridDim.x) * SPW + slot, b = tile / gridDim.x;
    const int s = rowpass ? a : b, l = rowpass ? b : a;
    const bool live = s < nsites && l < (rowpass ? I.h : I.w);
    int x = 0, y = 0, sx = 0, sy = 0;  // an idle group evaluates pixel (0,0) under its own depth: the group broadcasts stay uniform
    if (live) {
        if (dir == 0) { x = 1 + s * step; y = l; sx = x - 1; sy = y; }
        else if (dir == 2) { x = I.w - 2 - s * step; y = l; sx = x + 1; sy = y; }
        else if (dir == 1) { y = 1 + s * step; x = l; sx = x; sy = y - 1; }
        else { y = I.h - 2 - s * step; x = l; sx = x; sy = y + 1; }
    }
    const LeanK K = lean_consts(I);
    const int pi = y * I.w + x;
    const float cand = I.depth[sy * I.w + sx];
    const float c0 = I.cost[pi];
    const float c = cost_split_any<NMAX, LPP, STRICT>(I, K, x, y, cand, g);
    if (live && g == 0 && c < c0) { I.depth[pi] = cand; I.cost[pi] = c; }
}
// Pass 2 of a local propagation: one chain per HALF lanes (HALF = 64: one chain per wave, up to 64 steps; HALF = 32: two chains
// of up to 32 steps share a wave -- the default width 32 gives chains of 31 steps, so a wave per chain leaves half the lanes idle
// and needs two rounds of waves at 640x480; with two chains per wave the whole pass is resident at once).  Lane j of a half holds
// pixel j's old depth, old cost and table value.  A chain is an automaton with two states:
//   fresh   the predecessor kept its depth, so the step's candidate cost is the table value: the next ACCEPT is found with a ballot;
//   run     an accepted value v keeps propagating; c(x, v) has to be evaluated (LPP lanes per pixel, frames split over the lanes,
//           cost_split_lean) until a step rejects it -- the step after the rejecting one is fresh again.
// What costs time is the dependent latency of an evaluation round (~3 us: position chain, gathers at positions that miss the L2, model),
// and the pass ends with its slowest chain (measured: 80 % of the chains have no accept at all, a few per launch have 5-7 runs).  So a
// round is filled with HALF/LPP pixels that are LIKELY to be needed:
//   in a run     the next HALF/LPP pixels with v (the costs of a run are independent given v), accept / reject scanned with a ballot;
//   when fresh   the next R accepts s_0 < s_1 < .. of the table (s_r+1 = the first accept >= s_r + 2) with the first NP pixels of the run each
//                would start (measured: an accept is followed by a short run four times out of five).  Run r's evaluations are the right
//                ones if run r - 1 ended before s_r - 1 -- then s_r is the first accept after it and its predecessor is untouched;
//                otherwise they, and the later ones, are dropped.
// Every cost that is used was evaluated for exactly the (pixel, value) the step-by-step chain evaluates: identical maps (tests:
// vk_set_local_serial).  Both chains of a wave share one evaluation per round whatever state each is in; only the cheap
// bookkeeping diverges.
// The chains of ONE wave (two of up to 32 steps, or one of up to 64): cg / n = the chain of this lane's half (n <= 0: none).  Called by
// k_local_runs_lean (one launch per pass).
template <int HALF, int NMAX, int LPP, bool STRICT>
__device__ __forceinline__ static void local_runs_body(const Img& I, const LeanK& K, const ChainGeom cg, const int n, const float* __restrict__ tbl) {
    PHD_DECL;
    constexpr int NG = HALF / LPP;  // pixels per round
    constexpr int NP = NG >= 8 ? 4 : NG / 2, R = NG / NP;  // pixels per planned run, planned runs per round
    const int lane = threadIdx.x & 63, half = lane / HALF, hl = lane % HALF, g = hl / LPP, sub = hl % LPP;
    const unsigned long long hmask = HALF == 64 ? ~0ull : (0xffffffffull << (32 * half));
    const int hshift = HALF == 64 ? 0 : 32 * half;
    if (n <= 0) return;  // (a whole half leaves together; the other half's ballots are masked to itself)
    const bool has = hl < n;
    const int mypi = has ? cg.pi0 + hl * cg.stride : cg.pi0;
    const float d0 = has ? I.depth[mypi] : 0.f, c0 = has ? I.cost[mypi] : 0.f;
    const float first_cand = I.depth[cg.prev0];
    float t0 = INFINITY;
    if (tbl) t0 = has ? tbl[mypi] : INFINITY;
    else if (has) t0 = pixel_cost_any<NMAX, STRICT>(I, K, mypi % I.w, mypi / I.w, I.depth[mypi - cg.stride]);  // the table entry of my own step
    const unsigned long long tacc = (__ballot(has && t0 < c0) & hmask) >> hshift;  // steps whose table cost beats their current cost
    // number of leading groups of [g0, g0 + cnt) whose lanes are set in `m` (one bit per lane, a group's lanes agree)
    auto lead = [](unsigned long long m, int g0, int cnt) {
        const unsigned long long r = (~m >> (LPP * g0)) & (cnt * LPP >= 64 ? ~0ull : ((1ull << (LPP * cnt)) - 1ull));
        return r != 0ull ? (__ffsll((long long)r) - 1) / LPP : cnt;
    };
    int x = 0;
    bool running = false;
    float vrun = 0.f;
#ifdef VK_PHASE_CLOCKS
    if (d0 + c0 + t0 + first_cand == 123456.f) return;
    int rounds_ = 0;
#endif
    PHD_MARK(0);
    for (;;) {
#ifdef VK_PHASE_CLOCKS
        rounds_++;
#endif
        // ---- this round's pixel and value of my group
        int px, sr[R];
        float v, vr[R];
        bool act;
#pragma unroll
        for (int r = 0; r < R; r++) { sr[r] = -1; vr[r] = 0.f; }
        if (running) { px = x + g; v = vrun; act = px < n; }
        else {
            unsigned long long m = (tacc >> x) << x;  // accepts at steps >= x (x < n <= 64)
            if (m == 0ull) break;  // no accept left: the rest of the chain keeps its values
            int s = -1;
            v = 0.f;
#pragma unroll
            for (int r = 0; r < R; r++) {  // accept r = the first one at least two steps after accept r - 1
                sr[r] = m != 0ull ? __ffsll((long long)m) - 1 : -1;
                m = (sr[r] >= 0 && sr[r] + 2 < 64) ? (m >> (sr[r] + 2)) << (sr[r] + 2) : 0ull;
                const float dp = __shfl(d0, max(sr[r] - 1, 0), HALF);
                vr[r] = sr[r] == 0 ? first_cand : dp;
                if (g / NP == r) { s = sr[r]; v = vr[r]; }
            }
            px = s + 1 + g % NP; act = s >= 0 && px < n;
        }
        // ---- one evaluation for the whole wave
        const int pi = cg.pi0 + (act ? px : 0) * cg.stride;
        const float c = cost_split_any<NMAX, LPP, STRICT>(I, K, pi % I.w, pi / I.w, v, sub);
        const float c0p = __shfl(c0, min(max(px, 0), HALF - 1), HALF);
        const bool acc = act && c < c0p;
        const unsigned long long accm = (__ballot(acc) & hmask) >> hshift;
        PHD_MARK(1); PHD_ADD(6, 1);
        // ---- bookkeeping (uniform within a half)
        if (running) {
            const int L = lead(accm, 0, NG);  // a step past the end of the chain counts as rejecting
            if (g < L && sub == 0) { I.depth[pi] = v; I.cost[pi] = c; }
            x += L;
            if (L < NG) { running = false; x += 1; }  // the step at x rejected v: its successor is fresh again
        } else {
            bool go = true;  // still fresh, consuming the planned runs in order
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (go) {
                    const int s = sr[r];
                    if (s < 0) { x = n; go = false; }  // no table accept from x on: the chain is finished
                    else if (r > 0 && s < x) go = false;  // the previous run went over this accept: planned with the wrong state, dropped
                    else {
                        if (hl == s) { I.depth[mypi] = vr[r]; I.cost[mypi] = t0; }  // replace_if_better_depth (:201-207)
                        const int L = lead(accm, r * NP, NP);
                        if (g >= r * NP && g < r * NP + L && sub == 0) { I.depth[pi] = v; I.cost[pi] = c; }
                        if (L == NP) { running = true; vrun = vr[r]; x = s + 1 + NP; go = false; }  // still going: a run in progress
                        else { x = s + L + 2; if (x >= n) go = false; }  // rejected at s + 1 + L (or the chain ended there): s + L + 2 is fresh
                    }
                }
            }
        }
        PHD_MARK(2);
        if (x >= n) break;
    }
    PHD_MARK(3); PHD_ADD(8, 1);
#ifdef VK_PHASE_CLOCKS
    if (hl == 0) atomicAdd(&g_phase_d[16 + min(rounds_, 31)], 1ull);  // histogram of the rounds a chain's half took part in
#endif
}
// (round 5: the chain's own table entry with all gathers of the hypothesis in flight -- lean_rest<NMAX, NMAX - 1> -- measured: 23.3 us per launch either way,
// profiles/r05f_*: four resident waves per SIMD already hide that latency)
template <int HALF, int NMAX, int LPP, bool STRICT = false>
__global__ __launch_bounds__(64) static void k_local_runs_lean(Img I, int dir, int width, const float* __restrict__ tbl, int lines, int nchains) {
    if (!clamp_active(I)) return;
    constexpr int NH = 64 / HALF;  // chains per wave
    const int half = threadIdx.x / HALF;
    const int tile = xcd_band_tile(blockIdx.x, gridDim.x);
    const int chain = tile * NH + half;
    const bool in_range = chain < nchains;
    const ChainGeom cg = chain_geom(I.w, I.h, dir, width, in_range ? chain % lines : 0, in_range ? chain / lines : 0);
    local_runs_body<HALF, NMAX, LPP, STRICT>(I, lean_consts(I), cg, in_range ? cg.n : 0, tbl);
}
// what the E-step does to the pixel besides the rigidness maps: normalize_world_scale's depth half, the confidence of the depth priors
__device__ __forceinline__ static void estep_pixel_tail(const Img& I, const PoseBlock* P, int pi, float x, float y, float d, const float* __restrict__ world_scale) {
    const int w = I.w, h = I.h, npx = w * h;
    const float fw = (float)w, fh = (float)h;
    if (world_scale) I.depth[pi] = d * *world_scale;  // normalize_world_scale's depth half (voldor.cpp:314): the E-step above saw the unscaled map
    for (int f = 0; f < I.N_dp; f++) {
        if ((P->dp_ident >> f) & 1) {  // prior at the identity pose: sampled at the pixel itself (prior_parts)
            if (d > 0.f) {
                const float td = I.priors[(size_t)f * npx + pi];
                if (td > 0.f) I.confs[(size_t)f * npx + pi] = fast_rcp(1.f + depth_ratio(d, td, I.basefocal, I.omega, I.inv_arf));
            } else
                I.confs[(size_t)f * npx + pi] = 0.f;
            continue;
        }
        const H3 a = hom_dir(P->dpM[f], x, y);
        const float hz = fmaf(d, a.z, P->dpT[f][2]);
        const float qx2 = fmaf(d, a.x, P->dpT[f][0]) / hz, qy2 = fmaf(d, a.y, P->dpT[f][1]) / hz;  // see prior_parts
        if (hz > 0.f && qx2 >= 0.f && qx2 < fw && qy2 >= 0.f && qy2 < fh) {
            const float td = bilinear1(I.priors + (size_t)f * npx, w, h, qx2, qy2);
            if (td > 0.f) I.confs[(size_t)f * npx + pi] = fast_rcp(1.f + depth_ratio(hz, td, I.basefocal, I.omega, I.inv_arf));
        } else
            I.confs[(size_t)f * npx + pi] = 0.f;
    }
}
// E-step (optimize_depth.cu:84-138), lean geometry and model; per-block rigidness sums as k_update_rigidness
template <int NMAX>
__global__ __launch_bounds__(256) static void k_update_rigidness_lean(Img I, float* __restrict__ partial, const float* __restrict__ world_scale) {
    if (!clamp_active(I)) return;
    const int tile = xcd_band_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int xi = (tile % gridDim.x) * 64 + (threadIdx.x & 63);
    const int yi = (tile / gridDim.x) * 4 + (threadIdx.x >> 6);
    const bool live = xi < I.w && yi < I.h;
    const int w = I.w, h = I.h, npx = w * h, pi = live ? yi * w + xi : 0;
    const PoseBlock* P = I.P;
    const LeanK K = lean_consts(I);
    __shared__ float s_part[NMAX][4];
    const int blk = tile, nblk = gridDim.x * gridDim.y;
    const float d = live ? I.depth[pi] : 1.f;
    const float x = (float)(live ? xi : 0), y = (float)(live ? yi : 0), fw = (float)w, fh = (float)h;
    // one frame after the other (see lean_rest): position, gather, model, store, block sum
    {
        float px1 = x, py1 = y;
#pragma unroll
        for (int f = 0; f < NMAX; f++) {
            if (f < I.N) {  // uniform; selects inside
                float px2, py2;
                const bool zok = lean_step(P, f, x, y, d, px2, py2);
                const bool ok = live && zok && px1 >= 0.f && px1 < fw && py1 >= 0.f && py1 < fh;
                const float rdx = px2 - px1, rdy = py2 - py1;
                const float2 obs = (f == 0) ? I.flows[pi] : bilinear2_inside(I.flows + (size_t)f * npx, w, h, ok ? px1 : 0.f, ok ? py1 : 0.f);
                px1 = ok ? px2 : px1; py1 = ok ? py2 : py1;  // NOT advanced on invalid frames
                float r = 0.f;
                if (ok) {
                    const ObsTerms T = obs_terms(obs.x, obs.y, K.ia2, K.l2q);
                    r = fast_rcp(1.f + obs_ratio(T, rdx - obs.x, rdy - obs.y, K.qia2));
                }
                if (live) I.rig[(size_t)f * npx + pi] = r;
                const float ws = wave_sum(live ? r : 0.f);
                if ((threadIdx.x & 63) == 0) s_part[f][threadIdx.x >> 6] = ws;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < NMAX && (int)threadIdx.x < I.N) {
        const int f = threadIdx.x;
        partial[(size_t)f * nblk + blk] = (s_part[f][0] + s_part[f][1]) + (s_part[f][2] + s_part[f][3]);
    }
    if (!live) return;
    estep_pixel_tail(I, P, pi, x, y, d, world_scale);
}
// The same E-step with TWO pixels per lane (rows r and r + 2 of the 64 x 4 tile, 128 threads): geometry and model on float pairs (packed fp32: one
// instruction for both pixels where a packed form exists), gathers, validity and the wave sums per pixel.  The wave sums cover the same rows in
// the same order and land in the same slots: every output bit equals k_update_rigidness_lean's.
template <int NMAX>
__global__ __launch_bounds__(128) static void k_update_rigidness_pairs(Img I, float* __restrict__ partial, const float* __restrict__ world_scale) {
    if (!clamp_active(I)) return;
    const int tile = xcd_band_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int xi = (tile % gridDim.x) * 64 + (threadIdx.x & 63), wv = threadIdx.x >> 6;
    const int ya = (tile / gridDim.x) * 4 + wv, yb = ya + 2;
    const bool live_a = xi < I.w && ya < I.h, live_b = xi < I.w && yb < I.h;
    const int w = I.w, h = I.h, npx = w * h, pia = live_a ? ya * w + xi : 0, pib = live_b ? yb * w + xi : 0;
    const PoseBlock* P = I.P;
    const LeanK K = lean_consts(I);
    __shared__ float s_part[NMAX][4];
    const int blk = tile, nblk = gridDim.x * gridDim.y;
    const pf2 d = { live_a ? I.depth[pia] : 1.f, live_b ? I.depth[pib] : 1.f };
    const float x = (float)(xi < w ? xi : 0), fw = (float)w, fh = (float)h;
    const pf2 y = { (float)(live_a ? ya : 0), (float)(live_b ? yb : 0) };
    {
        pf2 px1 = pk_all(x), py1 = y;
#pragma unroll
        for (int f = 0; f < NMAX; f++) {
            if (f < I.N) {
                pf2 px2, py2;
                bool zok_a, zok_b;
                lean_step2(P, f, x, y, d, px2, py2, zok_a, zok_b);
                const bool ok_a = live_a && zok_a && px1.x >= 0.f && px1.x < fw && py1.x >= 0.f && py1.x < fh;
                const bool ok_b = live_b && zok_b && px1.y >= 0.f && px1.y < fw && py1.y >= 0.f && py1.y < fh;
                const pf2 rdx = px2 - px1, rdy = py2 - py1;
                const float2* __restrict__ layer = I.flows + (size_t)f * npx;
                const float2 oa = (f == 0) ? I.flows[pia] : bilinear2_inside(layer, w, h, ok_a ? px1.x : 0.f, ok_a ? py1.x : 0.f);
                const float2 ob = (f == 0) ? I.flows[pib] : bilinear2_inside(layer, w, h, ok_b ? px1.y : 0.f, ok_b ? py1.y : 0.f);
                px1 = pf2{ ok_a ? px2.x : px1.x, ok_b ? px2.y : px1.y }; py1 = pf2{ ok_a ? py2.x : py1.x, ok_b ? py2.y : py1.y };
                const pf2 ox = { oa.x, ob.x }, oy = { oa.y, ob.y };
                const ObsTerms2 T = obs_terms2(ox, oy, K.ia2, K.l2q);
                const pf2 den = 1.f + obs_ratio2(T, rdx - ox, rdy - oy, K.qia2);
                const float ra = ok_a ? fast_rcp(den.x) : 0.f, rb = ok_b ? fast_rcp(den.y) : 0.f;
                if (live_a) I.rig[(size_t)f * npx + pia] = ra;
                if (live_b) I.rig[(size_t)f * npx + pib] = rb;
                const float wsa = wave_sum(live_a ? ra : 0.f), wsb = wave_sum(live_b ? rb : 0.f);
                if ((threadIdx.x & 63) == 0) { s_part[f][wv] = wsa; s_part[f][wv + 2] = wsb; }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < NMAX && (int)threadIdx.x < I.N) {
        const int f = threadIdx.x;
        partial[(size_t)f * nblk + blk] = (s_part[f][0] + s_part[f][1]) + (s_part[f][2] + s_part[f][3]);
    }
    if (live_a) estep_pixel_tail(I, P, pia, x, y.x, d.x, world_scale);
    if (live_b) estep_pixel_tail(I, P, pib, x, y.y, d.y, world_scale);
}

// ---- serial-chain fallbacks, both modes: global propagation with step 1, local segments longer than one wave
// step==1: a true serial chain per line (not used by any shipped config; kept for parity)
template <int NMAX, bool STRICT>
__global__ static void k_global_prop_serial(Img I, int dir) {
    if (!clamp_active(I)) return;
    const LeanK K = lean_consts(I);
    auto try_depth = [&](int x, int y, float cand) { if constexpr (STRICT) try_depth_strict<NMAX>(I, x, y, cand); else try_depth_lean<NMAX>(I, K, x, y, cand); };
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (dir == 0 || dir == 2) {
        if (l >= I.h) return;
        if (dir == 0) for (int x = 1; x < I.w; x++) try_depth(x, l, I.depth[l * I.w + x - 1]);
        else for (int x = I.w - 2; x >= 0; x--) try_depth(x, l, I.depth[l * I.w + x + 1]);
    } else {
        if (l >= I.w) return;
        if (dir == 1) for (int y = 1; y < I.h; y++) try_depth(l, y, I.depth[(y - 1) * I.w + l]);
        else for (int y = I.h - 2; y >= 0; y--) try_depth(l, y, I.depth[(y + 1) * I.w + l]);
    }
}

// Fallback for segments longer than one wave can hold (width > 65): one thread walks one chain.
template <int NMAX, bool STRICT>
__global__ __launch_bounds__(64) static void k_local_serial(Img I, int dir, int width) {
    if (!clamp_active(I)) return;
    const int line = blockIdx.x * 64 + threadIdx.x;
    if (line >= ((dir == 0 || dir == 2) ? I.h : I.w)) return;
    const ChainGeom cg = chain_geom(I.w, I.h, dir, width, line, blockIdx.y);
    const LeanK K = lean_consts(I);
    float cand = cg.n > 0 ? I.depth[cg.prev0] : 0.f;
    for (int k = 0; k < cg.n; k++) {
        const int pi = cg.pi0 + k * cg.stride;
        float c;
        if constexpr (STRICT) c = pixel_cost_strict<NMAX>(I, pi % I.w, pi / I.w, cand); else c = pixel_cost_lean<NMAX>(I, K, pi % I.w, pi / I.w, cand);
        if (c < I.cost[pi]) { I.depth[pi] = cand; I.cost[pi] = c; }
        else cand = I.depth[pi];
    }
}

static Img make_img(const ImageSet& S, const OdParams& p) {
    Img I;
    I.flows = S.flows.as<float2>(); I.rig = S.rig.as<float>();
    I.priors = S.priors.as<float>(); I.pconfs = S.pconfs.as<float>(); I.confs = S.confs.as<float>();
    I.depth = S.depth.as<float>(); I.cost = S.cost.as<float>(); I.P = S.pb();
    I.N = p.N; I.N_dp = p.N_dp; I.w = p.w; I.h = p.h;
    I.lambda = p.lambda; I.omega = p.omega; I.inv_arf = 1.f / p.abs_resize_factor; I.arf = p.abs_resize_factor;
    I.basefocal = p.basefocal; I.disp_delta = p.disp_delta; I.delta = p.delta;
    I.tex = (p.strict && p.ref_tex) ? 1 : 0; I.n_layers = p.N; I.xw = nullptr; I.sf = nullptr;
    return I;
}


// Device-resident optimize_depth: all inputs already in `S`. Stage order optimize_depth.cu:462-494.
// STRICT: the reference's operation order on software transcendentals, plain launch structure (one thread per serial chain).
// Fast: k_cum_poses, then the lean kernels (cost + samples through the survivor queue, table + runs for the local passes).
template <int NMAX, bool STRICT>
int optimize_depth_launch(Context* c, ImageSet& S, const OdParams& p, bool cost_only) {
    const int w = p.w, h = p.h;
    Img I = make_img(S, p);
    const bool stale = STRICT && p.stale_depth != nullptr;  // OdParams::stale_depth: the reference's own device copy of the map (Appendix B-1)
    if (stale) {
        if (p.stale_refresh) VK_CHECK(hipMemcpyAsync(p.stale_depth, S.depth.p, sizeof(float) * (size_t)w * h, hipMemcpyDeviceToDevice, c->stream));
        I.depth = p.stale_depth;
    }
    const dim3 gpx((w + 63) / 64, (h + 3) / 4), bpx(256);
    if (STRICT && debug_switches().strict_filter == 2) {
        const bool fresh = c->sf_stats.p == nullptr;
        if (int e = c->sf_stats.reserve(sizeof(unsigned long long) * 4)) return e;
        if (fresh) VK_CHECK(hipMemsetAsync(c->sf_stats.p, 0, sizeof(unsigned long long) * 4, c->stream));
        I.sf = c->sf_stats.as<unsigned long long>();
    }
    // the table pass of the local propagation behind the filter (k_local_table_filter + k_local_table_exact) where the pass is several rounds of waves long: at 1920x1080 the packed strict
    // kernel takes 245 us against 490 for every entry (+ 100 for the filter); at 1241x376 / 640x480 every wave of the plain kernel is resident at once and the pass lasts as long as ONE
    // wave's dependent chain, whichever number of waves: 77 + 30 us against 103, 41 + 14 against 39 -- there the plain kernel stays
    const bool table_filter = STRICT && !cost_only && !p.update_rigidness_only && p.local_prop_width > 0 && debug_switches().strict_filter && p.N + p.N_dp <= 64 &&
                              (debug_switches().strict_table_filter == 2 || (debug_switches().strict_table_filter == 1 && (size_t)w * h >= 1000000));
    if (table_filter) {  // its queues
        const int ntiles = gpx.x * gpx.y, qcap = (ntiles / SFQ + 10) * 256;
        if (int e = c->sf_qcnt.reserve(sizeof(unsigned) * 4 * SFQ)) return e;
        if (int e = c->sf_qlist.reserve(sizeof(unsigned) * (size_t)SFQ * qcap)) return e;
        VK_CHECK(hipMemsetAsync(c->sf_qcnt.p, 0, sizeof(unsigned) * 4 * SFQ, c->stream));
    }
    const bool plain = STRICT && debug_switches().strict_plain;  // strict mode on the plain launch structures of rounds 1-3 (verification: same bits either way)
    // fast mode: the projective maps of the chain (and the world-scale factor) are prepared by an extra workgroup of the first fb_smooth
    // launch when there is one, by their own small launch otherwise
    const int fb_done = cost_only ? 0 : p.fb_done;  // (OdParams::fb_done: which stacks' fb_smooth ran during the pose half)
    const bool cum_in_fb = !STRICT && !cost_only && !p.update_rigidness_only && p.fb_smooth && p.N > 0 && !(fb_done & 1);
    if constexpr (!STRICT) { if (!cum_in_fb && !((fb_done & 1) && p.cum_done)) cum_poses_launch(c, S.pb(), p.N, p.N_dp, p.world_scale_out); }
    auto cost_rand = [&](int n_rand, uint32_t epoch) {
        if constexpr (STRICT) {
            const bool filt = debug_switches().strict_filter && p.N + p.N_dp <= 64;  // hardware-fp32 pre-filter, strict arithmetic for what it cannot discard
            if (plain || debug_switches().cost_rand_plain || (p.N_dp > 1 && !filt)) hipLaunchKernelGGL(k_cost_rand_strict<NMAX>, gpx, bpx, 0, c->stream, I, n_rand, epoch, p.range_factor);
            else if (filt) {
                hipLaunchKernelGGL(k_cost_strict<NMAX>, gpx, bpx, 0, c->stream, I);
                if (n_rand > 0) hipLaunchKernelGGL(k_cost_rand_f_strict<NMAX>, gpx, bpx, 0, c->stream, I, n_rand, epoch, p.range_factor);
            }
            else hipLaunchKernelGGL(k_cost_rand_q_strict<NMAX>, gpx, bpx, 0, c->stream, I, n_rand, epoch, p.range_factor);
        }
        else if (debug_switches().cost_rand_plain) hipLaunchKernelGGL(k_cost_rand_plain<NMAX>, gpx, bpx, 0, c->stream, I, n_rand, epoch, p.range_factor);
        else hipLaunchKernelGGL(k_cost_rand_q<NMAX>, gpx, bpx, 0, c->stream, I, n_rand, epoch, p.range_factor);
    };
    if (cost_only) {
        cost_rand(0, 0u);
        VK_CHECK_LAST();
        return 0;
    }
    if (!p.update_rigidness_only) {
        if (p.fb_smooth && fb_done != 3) {
            if (STRICT) {
                if (!(fb_done & 1)) { if (int e = fb_smooth_strict_device(c, I.rig, p.N, w, h, p.s0_ems_prob, p.no_change_prob, &S.pb()->n_active)) return e; }
                if (!(fb_done & 2)) { if (int e = fb_smooth_strict_device(c, I.confs, p.N_dp, w, h, p.s0_ems_prob, p.no_change_prob, nullptr)) return e; }
            } else {
                if (!(fb_done & 1)) { if (int e = fb_smooth_device(c, I.rig, p.N, w, h, p.s0_ems_prob, p.no_change_prob, &S.pb()->n_active, cum_in_fb ? S.pb() : nullptr, p.N, p.N_dp,
                                                                   p.world_scale_out)) return e; }
                if (!(fb_done & 2)) { if (int e = fb_smooth_device(c, I.confs, p.N_dp, w, h, p.s0_ems_prob, p.no_change_prob, nullptr)) return e; }
            }
        }
        if (c->prof) prof_begin_inner(c);
        if (STRICT && p.ref_rng && p.n_rand_samples > 0) {  // --reference_rng 1: the per-pixel cuRAND states, standing rand_epoch draws after curand_init
            if (int e = xorwow_pixel_states_device(c, w * h, c->rand_epoch)) return e;
            I.xw = c->xw_px_states.as<vrc_xorwow>();
        }
        cost_rand(p.n_rand_samples, c->rand_epoch);
        if (I.xw) { c->xw_px_epoch += (uint32_t)p.n_rand_samples; I.xw = nullptr; }
        if (c->prof) prof_end_inner(c, "cost_rand", 1);
        c->rand_epoch += (uint32_t)(p.n_rand_samples > 0 ? p.n_rand_samples : 0);
        const int order[4] = { 0, 3, 2, 1 };  // L2R, B2T, R2L, T2B (:481-484, :487-490)
        if (p.global_prop_step > 0) {
            for (int k = 0; k < 4; k++) {
                const int dir = order[k];
                const bool rowpass = (dir == 0 || dir == 2);
                const int len = rowpass ? w : h, lines = rowpass ? h : w;
                if (p.global_prop_step >= 2) {
                    const int nsites = (len - 1 + p.global_prop_step - 1) / p.global_prop_step;
                    if (nsites <= 0) continue;
                    if (STRICT && plain) hipLaunchKernelGGL(k_global_prop_sites_strict<NMAX>, dim3((nsites + 63) / 64, lines), dim3(64), 0, c->stream, I, dir, p.global_prop_step, nsites);
                    else if (STRICT) {  // a site's strict evaluation is ~10 us of dependent fp64 on one lane: a group of lanes per site at every size, one frame per lane
                        constexpr int GL = 8, SPW = 64 / GL;
                        if (rowpass) hipLaunchKernelGGL((k_global_prop_split_lean<NMAX, GL, STRICT>), dim3((nsites + SPW - 1) / SPW, lines), dim3(64), 0, c->stream, I, dir, p.global_prop_step, nsites);
                        else hipLaunchKernelGGL((k_global_prop_split_lean<NMAX, GL, STRICT>), dim3((lines + SPW - 1) / SPW, nsites), dim3(64), 0, c->stream, I, dir, p.global_prop_step, nsites);
                    }
                    else if (debug_switches().global_split && (size_t)w * h <= 600000) {  // latency regime only (640x480: 5.9 -> 5.0 us per pass, 1241x376: 11.2 -> 10.6); at 1080p the pass is throughput bound and eight lanes re-walking the chain cost 35 -> 52 us
                        constexpr int GL = NMAX <= 8 ? 4 : 8, SPW = 64 / GL;  // lanes per site as in the run evaluations of the local pass
                        if (rowpass) hipLaunchKernelGGL((k_global_prop_split_lean<NMAX, GL>), dim3((nsites + SPW - 1) / SPW, lines), dim3(64), 0, c->stream, I, dir, p.global_prop_step, nsites);
                        else hipLaunchKernelGGL((k_global_prop_split_lean<NMAX, GL>), dim3((lines + SPW - 1) / SPW, nsites), dim3(64), 0, c->stream, I, dir, p.global_prop_step, nsites);
                    }
                    else if (rowpass) hipLaunchKernelGGL(k_global_prop_sites_lean<NMAX>, dim3((nsites + 63) / 64, lines), dim3(64), 0, c->stream, I, dir, p.global_prop_step, nsites);
                    else hipLaunchKernelGGL(k_global_prop_sites_lean<NMAX>, dim3((lines + 63) / 64, nsites), dim3(64), 0, c->stream, I, dir, p.global_prop_step, nsites);
                } else  // step 1: a true serial chain per line (no shipped config uses it)
                    hipLaunchKernelGGL((k_global_prop_serial<NMAX, STRICT>), dim3((lines + 63) / 64), dim3(64), 0, c->stream, I, dir);
            }
        }
        if (p.local_prop_width > 0) {
            if (c->prof) prof_begin_inner(c);
            for (int k = 0; k < 4; k++) {
                const int dir = order[k];
                const bool rowpass = (dir == 0 || dir == 2);
                const int len = rowpass ? w : h, lines = rowpass ? h : w;
                const int nseg = (len + p.local_prop_width - 1) / p.local_prop_width;
                if (p.local_prop_width <= 65 && !(STRICT ? plain : debug_switches().local_serial != 0)) {  // chains of <= 64 steps: table + one wave per chain
                    // small images (the pass is a few thousand waves, its time is latency): every chain tabulates its own steps, one lane per
                    // pixel, at the head of the runs kernel -- one launch less per pass (cfg2: 28.0 -> 27.0 us).  Larger ones are throughput
                    // bound and the tiled table kernel reads coalesced (column chains do not): measured neutral at 1241x376, 5 % slower at 1080p
                    const bool own_table = (size_t)w * h <= 400000 && !STRICT;  // (strict: the tiled table kernel at every size; own table measured equal)
                    bool tabled = own_table;
                    if constexpr (STRICT) {
                        if (table_filter) {
                            const int ntiles = gpx.x * gpx.y, qcap = (ntiles / SFQ + 10) * 256;
                            unsigned* qcnt = c->sf_qcnt.as<unsigned>() + k * SFQ;
                            hipLaunchKernelGGL(k_local_table_filter<NMAX>, gpx, bpx, 0, c->stream, I, dir, p.local_prop_width, c->local_tbl.as<float>(), qcnt, c->sf_qlist.as<unsigned>(), qcap);
                            hipLaunchKernelGGL(k_local_table_exact<NMAX>, dim3((qcap / 256) * SFQ), bpx, 0, c->stream, I, dir, c->local_tbl.as<float>(), qcnt, c->sf_qlist.as<unsigned>(), qcap);
                            tabled = true;
                        }
                    }
                    if (!tabled) hipLaunchKernelGGL((k_local_table_lean<NMAX, STRICT>), gpx, bpx, 0, c->stream, I, dir, p.local_prop_width, c->local_tbl.as<float>());
                    const float* tblp = own_table ? nullptr : c->local_tbl.as<float>();
                    const int nchains = lines * nseg;
                    // lanes per pixel of a run evaluation (cost_split_lean): quads up to 8 frames, eight beyond.  (Pairs -- 16 pixels per round, four planned
                    // runs -- halve the rounds again but need 137 registers: 3 waves per SIMD for a pass of 4.7, 46 us instead of 27.)
                    constexpr int LR_LPP = NMAX <= 8 ? 4 : 8;
                    if (p.local_prop_width <= 33)  // chains of <= 32 steps: two per wave
                        hipLaunchKernelGGL((k_local_runs_lean<32, NMAX, LR_LPP, STRICT>), dim3((nchains + 1) / 2), dim3(64), 0, c->stream, I, dir, p.local_prop_width, tblp, lines, nchains);
                    else
                        hipLaunchKernelGGL((k_local_runs_lean<64, NMAX, LR_LPP, STRICT>), dim3(nchains), dim3(64), 0, c->stream, I, dir, p.local_prop_width, tblp, lines, nchains);
                } else
                    hipLaunchKernelGGL((k_local_serial<NMAX, STRICT>), dim3((lines + 63) / 64, nseg), dim3(64), 0, c->stream, I, dir, p.local_prop_width);
            }
            if (c->prof) prof_end_inner(c, "local_pass", 4);
        }
    }
    const int nblk = gpx.x * gpx.y;
    if constexpr (STRICT) hipLaunchKernelGGL(k_update_rigidness_strict<NMAX>, gpx, bpx, 0, c->stream, I, c->rig_partial.as<float>());
    else if (debug_switches().estep_pairs == 2 || (debug_switches().estep_pairs == 1 && (size_t)w * h >= 1500000))
        // two pixels per lane on packed fp32 where the pass fills the chip several times over (1920x1080 N=10: 76.9 -> 68.3 us; in the latency regime half
        // the waves cost more than the instructions save: 1241x376 N=8 17.9 -> 19.9 us; profiles/r05i_*, r05j_*)
        hipLaunchKernelGGL(k_update_rigidness_pairs<NMAX>, gpx, dim3(128), 0, c->stream, I, c->rig_partial.as<float>(), p.N > 0 ? p.world_scale_out : nullptr);
    else hipLaunchKernelGGL(k_update_rigidness_lean<NMAX>, gpx, bpx, 0, c->stream, I, c->rig_partial.as<float>(), p.N > 0 ? p.world_scale_out : nullptr);
    if (p.N > 0) {
        if (!STRICT && p.defer_reduce) {  // left for the next correspondence trace (OdParams::defer_reduce)
            ReduceArgs& a = c->pending_reduce;
            a.partial = c->rig_partial.as<float>(); a.nblk = nblk; a.npx = w * h; a.n_launch = p.N; a.scale_ready = 1; a.cams = c->cams.as<CamState>(); a.P = S.pb(); a.scale_out = p.world_scale_out;
        } else
            reduce_density_launch(c, c->rig_partial.as<float>(), nblk, w * h, S.pb(), p.N, p.world_scale_out, STRICT ? 0 : 1);
    }
    // normalize_world_scale's depth half: strict mode as its own pass after the E-step; the fast E-step kernel has stored the scaled map
    if (stale) VK_CHECK(hipMemcpyAsync(S.depth.p, p.stale_depth, sizeof(float) * (size_t)w * h, hipMemcpyDeviceToDevice, c->stream));  // d_depth.copy_to_host(h_o_depth): the host's map, which alone is normalised
    if (STRICT && p.N > 0 && p.world_scale_out) { if (int e = scale_device(c, S.depth.as<float>(), p.world_scale_out, (size_t)w * h)) return e; }
    VK_CHECK_LAST();
    return 0;
}

}  // namespace vk

#if defined(VK_PHASE_CLOCKS) && defined(VK_PHASE_UNIT)  // profiling builds (scripts/phase_clocks.sh): the clocks of ONE translation unit (vk_depth_i6.hip: the 640x480 N=5 window)
extern "C" __attribute__((visibility("default"))) int vk_phase_read_depth(unsigned long long* out, int n, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vk::g_phase_d), sizeof(unsigned long long) * (size_t)(n < 64 ? n : 64)) != hipSuccess) return 2;
    if (reset) { unsigned long long z[64] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(vk::g_phase_d), z, sizeof z) != hipSuccess) return 3; }
    return 0;
}
#endif
