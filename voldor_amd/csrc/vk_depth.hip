// voldor_amd/csrc/vk_depth.hip -- depth / rigidness half of the EM loop on gfx950.
// Replaces gpu-kernels/optimize_depth.cu:84-291 (kernels) and :462-494 (stage order) and
// gpu-kernels/fb_smooth.h:17-109.
//
// Launch structure per optimize_depth call (reference: 20+ launches, 10 of them streaming a
// 48-byte RNG state per pixel):
//   fb_rows / fb_cols           2 launches per map set (reference: 6), LDS-transposed rows
//   cost_rand                   1 launch: cost map + all n_rand samples, depth/cost in regs
//   global_prop x4              1 launch each, one thread per candidate site (step>=2)
//   local_prop  x4              1 launch each, one thread per (segment, line) chain
//   update_rigidness            1 launch (+ per-block rigidness sums for the density test)
#include "vk_common.hpp"
#include "vk_device.hpp"

namespace vk {

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
    float lambda, omega, inv_arf, basefocal, disp_delta, delta;
};

// compute_pixel_cost, optimize_depth.cu:140-198
__device__ static float pixel_cost(const Img& I, int px, int py, float depth) {
    const int w = I.w, h = I.h, npx = w * h, pi = py * w + px;
    const PoseBlock* P = I.P;
    float cost_sum = 0.f, wsum = 0.f;
    P3 o = backproject(P, (float)px, (float)py, depth);
    float px1 = (float)px, py1 = (float)py;
    for (int f = 0; f < I.N; f++) {
        o = transform(P->Rs[f], P->ts[f], o);
        float px2, py2;
        project(P, o, px2, py2);
        if (o.z > 0.f && px1 >= 0.f && px1 < (float)w && py1 >= 0.f && py1 < (float)h) {
            float2 d2 = bilinear2(I.flows + (size_t)f * npx, w, h, px1, py1);
            float wgt = I.rig[(size_t)f * npx + pi];
            cost_sum += wgt * neglog_rigidness_from_flows(px2 - px1, py2 - py1, d2.x, d2.y, I.lambda, I.inv_arf);
            wsum += wgt;
            px1 = px2; py1 = py2;
        }
    }
    for (int f = 0; f < I.N_dp; f++) {
        P3 q = transform(P->dpRs[f], P->dpts[f], backproject(P, (float)px, (float)py, depth));
        float qx, qy;
        project(P, q, qx, qy);
        if (q.z > 0.f && qx >= 0.f && qx < (float)w && qy >= 0.f && qy < (float)h) {
            float td = bilinear1(I.priors + (size_t)f * npx, w, h, qx, qy);
            if (td > 0.f) {
                float tpc = bilinear1(I.pconfs + (size_t)f * npx, w, h, qx, qy);
                float tc = bilinear1(I.confs + (size_t)f * npx, w, h, qx, qy);
                float wgt = tpc * tc * ((I.disp_delta > 0.f && f == 0) ? I.disp_delta : I.delta);
                cost_sum += wgt * __logf(1.f + depth_ratio(q.z, td, I.basefocal, I.omega, I.inv_arf));
                wsum += wgt;
            }
        }
    }
    if (wsum == 0.f) return INFINITY;
    return cost_sum / fmaxf(wsum, 1.1920929e-07f);
}

// ---- cost map + all random samples, fused (optimize_depth.cu:279-284 + :269-277 x n_rand) ----
__global__ __launch_bounds__(256) static void k_cost_rand(Img I, int n_rand, uint32_t epoch0, float range_factor) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= I.w || y >= I.h) return;
    const int pi = y * I.w + x;
    float d = I.depth[pi];
    float c = pixel_cost(I, x, y, d);
    for (int it = 0; it < n_rand; it++) {
        float u = u01(rng3(RAND_SEED, (uint32_t)pi, epoch0 + (uint32_t)it));
        float dn = 1.0f / (range_factor * u + (1.0f / 1e5f));  // MAXIMUM_DEPTH, :15,:273
        float cn = pixel_cost(I, x, y, dn);
        if (cn < c) { c = cn; d = dn; }
    }
    I.depth[pi] = d;
    I.cost[pi] = c;
}

// replace_if_better_depth, optimize_depth.cu:201-207
__device__ __forceinline__ static void try_depth(const Img& I, int x, int y, float cand) {
    const int pi = y * I.w + x;
    float c = pixel_cost(I, x, y, cand);
    if (c < I.cost[pi]) { I.depth[pi] = cand; I.cost[pi] = c; }
}

// ---- global propagation (optimize_depth.cu:209-235). With step>=2 the sites of one pass are
// independent (reads x-1, writes x): one thread per site.  dir: 0 L2R,
// 1 T2B, 2 R2L, 3 B2T.
__global__ __launch_bounds__(256) static void k_global_prop_sites(Img I, int dir, int step, int nsites) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;  // site index along the pass direction
    const int l = blockIdx.y;                             // line (row for 0/2, column for 1/3)
    if (s >= nsites) return;
    if (dir == 0) { int x = 1 + s * step; try_depth(I, x, l, I.depth[l * I.w + x - 1]); }
    else if (dir == 2) { int x = I.w - 2 - s * step; try_depth(I, x, l, I.depth[l * I.w + x + 1]); }
    else if (dir == 1) { int y = 1 + s * step; try_depth(I, l, y, I.depth[(y - 1) * I.w + l]); }
    else { int y = I.h - 2 - s * step; try_depth(I, l, y, I.depth[(y + 1) * I.w + l]); }
}
// step==1: a true serial chain per line (not used by any shipped config; kept for parity)
__global__ static void k_global_prop_serial(Img I, int dir) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (dir == 0 || dir == 2) {
        if (l >= I.h) return;
        if (dir == 0) for (int x = 1; x < I.w; x++) try_depth(I, x, l, I.depth[l * I.w + x - 1]);
        else for (int x = I.w - 2; x >= 0; x--) try_depth(I, x, l, I.depth[l * I.w + x + 1]);
    } else {
        if (l >= I.w) return;
        if (dir == 1) for (int y = 1; y < I.h; y++) try_depth(I, l, y, I.depth[(y - 1) * I.w + l]);
        else for (int y = I.h - 2; y >= 0; y--) try_depth(I, l, y, I.depth[(y + 1) * I.w + l]);
    }
}

// ---- local propagation (optimize_depth.cu:237-267): serial chain inside each `width` segment.
// One thread per (segment, line). Row passes put adjacent lanes on adjacent rows, so each lane
// walks its own cache line; column passes put adjacent lanes on adjacent columns (coalesced).
__global__ __launch_bounds__(64) static void k_local_prop(Img I, int dir, int width) {
    const int w = I.w, h = I.h;
    if (dir == 0 || dir == 2) {
        const int y = blockIdx.x * 64 + threadIdx.x, seg = blockIdx.y;
        if (y >= h) return;
        const int px = seg * width;
        if (dir == 0) {
            for (int x = max(1, px + 1); x < min(w, px + width); x++) try_depth(I, x, y, I.depth[y * w + x - 1]);
        } else {
            for (int x = min(w - 2, px + width - 2); x >= max(0, px); x--) try_depth(I, x, y, I.depth[y * w + x + 1]);
        }
    } else {
        const int x = blockIdx.x * 64 + threadIdx.x, seg = blockIdx.y;
        if (x >= w) return;
        const int py = seg * width;
        if (dir == 1) {
            for (int y = max(1, py + 1); y < min(h, py + width); y++) try_depth(I, x, y, I.depth[(y - 1) * w + x]);
        } else {
            for (int y = min(h - 2, py + width - 2); y >= max(0, py); y--) try_depth(I, x, y, I.depth[(y + 1) * w + x]);
        }
    }
}

// ---- E-step (optimize_depth.cu:84-138) + per-block sums of each rigidness map (the density
// test of voldor.cpp:171 then needs no D2H of the maps).
__global__ __launch_bounds__(256) static void k_update_rigidness(Img I, float* __restrict__ partial) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const bool live = x < I.w && y < I.h;
    const int w = I.w, h = I.h, npx = w * h, pi = y * w + x;
    const PoseBlock* P = I.P;
    __shared__ float s_part[4];
    const int blk = blockIdx.y * gridDim.x + blockIdx.x, nblk = gridDim.x * gridDim.y;
    float d = live ? I.depth[pi] : 1.f;
    P3 o = backproject(P, (float)x, (float)y, d);
    float px1 = (float)x, py1 = (float)y;
    for (int f = 0; f < I.N; f++) {
        o = transform(P->Rs[f], P->ts[f], o);
        float px2, py2;
        project(P, o, px2, py2);
        float r = 0.f;
        if (live && o.z > 0.f && px1 >= 0.f && px1 < (float)w && py1 >= 0.f && py1 < (float)h) {
            float2 d2 = bilinear2(I.flows + (size_t)f * npx, w, h, px1, py1);
            r = rigidness_from_flows(px2 - px1, py2 - py1, d2.x, d2.y, I.lambda, I.inv_arf);
            px1 = px2; py1 = py2;
        }
        if (live) I.rig[(size_t)f * npx + pi] = r;
        float ws = wave_sum(live ? r : 0.f);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = ws;
        __syncthreads();
        if (threadIdx.x == 0) partial[(size_t)f * nblk + blk] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        __syncthreads();
    }
    if (!live) return;
    for (int f = 0; f < I.N_dp; f++) {
        P3 q = transform(P->dpRs[f], P->dpts[f], backproject(P, (float)x, (float)y, d));
        float qx, qy;
        project(P, q, qx, qy);
        if (q.z > 0.f && qx >= 0.f && qx < (float)w && qy >= 0.f && qy < (float)h) {
            float td = bilinear1(I.priors + (size_t)f * npx, w, h, qx, qy);
            if (td > 0.f)
                I.confs[(size_t)f * npx + pi] = 1.f / (1.f + depth_ratio(q.z, td, I.basefocal, I.omega, I.inv_arf));
        } else
            I.confs[(size_t)f * npx + pi] = 0.f;
    }
}
// fixed-order second stage: cams[f].pose_rigidness_density = sum(partial[f][:]) / npx
__global__ static void k_reduce_density(const float* __restrict__ partial, int nblk, int npx, CamState* cams) {
    const int f = blockIdx.x;
    __shared__ float s[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) acc += partial[(size_t)f * nblk + i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) cams[f].pose_rigidness_density = ((s[0] + s[1]) + (s[2] + s[3])) / (float)npx;
}

// ---- forward-backward smoothing (fb_smooth.h:26-70) -------------------------------------
// Row pass: one wave owns 64 rows of one map and walks the columns in 64-wide chunks that
// are staged through LDS, so global traffic is coalesced (the reference reads with a
// row-pitch stride between adjacent lanes) while each lane runs its row's serial recurrence.
__global__ __launch_bounds__(64) static void k_fb_rows(float* __restrict__ maps, float* __restrict__ fwd, int w, int h,
                                                        float e0, float p) {
    __shared__ float tE[64][65];
    __shared__ float tF[64][65];
    const int lane = threadIdx.x, r0 = blockIdx.x * 64;
    float* m = maps + (size_t)blockIdx.y * w * h;
    float* fw = fwd + (size_t)blockIdx.y * w * h;
    const int row = r0 + lane;
    const bool live = row < h;
    const int nchunk = (w + 63) / 64;
    const float q = 1.f - p;
    float prev = live ? m[(size_t)row * w] : 0.5f;
    for (int ch = 0; ch < nchunk; ch++) {  // forward messages, FB_MSG_L2R :27-36
        const int c0 = ch * 64;
        for (int r = 0; r < 64; r++) {
            int rr = r0 + r, cc = c0 + lane;
            tE[r][lane] = (rr < h && cc < w) ? m[(size_t)rr * w + cc] : 0.5f;
        }
        __syncthreads();
        const int nc = min(64, w - c0);
        for (int j = 0; j < nc; j++) {
            float e1 = tE[lane][j];
            float s0 = (prev * q + (1.f - prev) * p) * e0;
            float s1 = (prev * p + (1.f - prev) * q) * e1;
            prev = s1 / (s0 + s1);
            tF[lane][j] = prev;
        }
        __syncthreads();
        for (int r = 0; r < 64; r++) {
            int rr = r0 + r, cc = c0 + lane;
            if (rr < h && cc < w) fw[(size_t)rr * w + cc] = tF[r][lane];
        }
        __syncthreads();
    }
    prev = live ? m[(size_t)row * w + (w - 1)] : 0.5f;
    for (int ch = nchunk - 1; ch >= 0; ch--) {  // backward messages :37-46 fused with posterior :65-69
        const int c0 = ch * 64;
        for (int r = 0; r < 64; r++) {
            int rr = r0 + r, cc = c0 + lane;
            bool ok = rr < h && cc < w;
            tE[r][lane] = ok ? m[(size_t)rr * w + cc] : 0.5f;
            tF[r][lane] = ok ? fw[(size_t)rr * w + cc] : 0.5f;
        }
        __syncthreads();
        const int nc = min(64, w - c0);
        for (int j = nc - 1; j >= 0; j--) {
            float e1 = tE[lane][j];
            float s0 = prev * e1 * q + (1.f - prev) * p * e0;
            float s1 = prev * e1 * p + (1.f - prev) * q * e0;
            prev = s1 / (s0 + s1);
            float F = tF[lane][j];
            float a1 = F * prev, a0 = (1.f - F) * (1.f - prev);
            tE[lane][j] = a1 / (a0 + a1);
        }
        __syncthreads();
        for (int r = 0; r < 64; r++) {
            int rr = r0 + r, cc = c0 + lane;
            if (rr < h && cc < w) m[(size_t)rr * w + cc] = tE[r][lane];
        }
        __syncthreads();
    }
}
// Column pass: lane = column, naturally coalesced (FB_MSG_T2B/B2T :47-64 + posterior).
__global__ __launch_bounds__(64) static void k_fb_cols(float* __restrict__ maps, float* __restrict__ fwd, int w, int h,
                                                        float e0, float p) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= w) return;
    float* m = maps + (size_t)blockIdx.y * w * h + x;
    float* fw = fwd + (size_t)blockIdx.y * w * h + x;
    const float q = 1.f - p;
    float prev = m[0];
    for (int i = 0; i < h; i++) {
        float e1 = m[(size_t)i * w];
        float s0 = (prev * q + (1.f - prev) * p) * e0;
        float s1 = (prev * p + (1.f - prev) * q) * e1;
        prev = s1 / (s0 + s1);
        fw[(size_t)i * w] = prev;
    }
    prev = m[(size_t)(h - 1) * w];
    for (int i = h - 1; i >= 0; i--) {
        float e1 = m[(size_t)i * w];
        float s0 = prev * e1 * q + (1.f - prev) * p * e0;
        float s1 = prev * e1 * p + (1.f - prev) * q * e0;
        prev = s1 / (s0 + s1);
        float F = fw[(size_t)i * w];
        float a1 = F * prev, a0 = (1.f - F) * (1.f - prev);
        m[(size_t)i * w] = a1 / (a0 + a1);
    }
}

int fb_smooth_device(Context* c, float* maps, int n_maps, int w, int h, float s0_ems_prob, float no_change_prob) {
    if (n_maps <= 0) return 0;
    if (int e = c->fb_scratch.reserve(sizeof(float) * (size_t)w * h * n_maps)) return e;
    float* fwd = c->fb_scratch.as<float>();
    hipLaunchKernelGGL(k_fb_rows, dim3((h + 63) / 64, n_maps), dim3(64), 0, c->stream, maps, fwd, w, h, s0_ems_prob, no_change_prob);
    hipLaunchKernelGGL(k_fb_cols, dim3((w + 63) / 64, n_maps), dim3(64), 0, c->stream, maps, fwd, w, h, s0_ems_prob, no_change_prob);
    VK_CHECK_LAST();
    return 0;
}

static Img make_img(const ImageSet& S, const OdParams& p) {
    Img I;
    I.flows = S.flows.as<float2>(); I.rig = S.rig.as<float>();
    I.priors = S.priors.as<float>(); I.pconfs = S.pconfs.as<float>(); I.confs = S.confs.as<float>();
    I.depth = S.depth.as<float>(); I.cost = S.cost.as<float>(); I.P = S.pb();
    I.N = p.N; I.N_dp = p.N_dp; I.w = p.w; I.h = p.h;
    I.lambda = p.lambda; I.omega = p.omega; I.inv_arf = 1.f / p.abs_resize_factor;
    I.basefocal = p.basefocal; I.disp_delta = p.disp_delta; I.delta = p.delta;
    return I;
}

// Device-resident optimize_depth: all inputs already in `S`. Stage order optimize_depth.cu:462-494.
int optimize_depth_device(Context* c, ImageSet& S, const OdParams& p) {
    const int w = p.w, h = p.h;
    if (int e = S.cost.reserve(sizeof(float) * (size_t)w * h)) return e;
    if (c->rand_w != w || c->rand_h != h) {  // reference re-inits the RNG when the size changes (:358-361)
        if (c->rand_w != -1) c->rand_epoch = 0;  // (-1: epoch was set explicitly through vk_set_rand_epoch)
        c->rand_w = w; c->rand_h = h;
    }
    Img I = make_img(S, p);
    const dim3 gpx((w + 63) / 64, (h + 3) / 4), bpx(256);
    if (c->prof) prof_begin(c);
    if (!p.update_rigidness_only) {
        if (p.fb_smooth) {
            if (int e = fb_smooth_device(c, I.rig, p.N, w, h, p.s0_ems_prob, p.no_change_prob)) return e;
            if (int e = fb_smooth_device(c, I.confs, p.N_dp, w, h, p.s0_ems_prob, p.no_change_prob)) return e;
        }
        hipLaunchKernelGGL(k_cost_rand, gpx, bpx, 0, c->stream, I, p.n_rand_samples, c->rand_epoch, p.range_factor);
        c->rand_epoch += (uint32_t)(p.n_rand_samples > 0 ? p.n_rand_samples : 0);
        if (p.global_prop_step > 0) {
            const int order[4] = { 0, 3, 2, 1 };  // L2R, B2T, R2L, T2B (:481-484)
            for (int k = 0; k < 4; k++) {
                const int dir = order[k];
                const bool rowpass = (dir == 0 || dir == 2);
                const int len = rowpass ? w : h, lines = rowpass ? h : w;
                if (p.global_prop_step >= 2) {
                    const int nsites = (len - 1 + p.global_prop_step - 1) / p.global_prop_step;
                    if (nsites > 0)
                        hipLaunchKernelGGL(k_global_prop_sites, dim3((nsites + 63) / 64, lines), dim3(64), 0, c->stream,
                                           I, dir, p.global_prop_step, nsites);
                } else
                    hipLaunchKernelGGL(k_global_prop_serial, dim3((lines + 63) / 64), dim3(64), 0, c->stream, I, dir);
            }
        }
        if (p.local_prop_width > 0) {
            const int order[4] = { 0, 3, 2, 1 };  // (:487-490)
            for (int k = 0; k < 4; k++) {
                const int dir = order[k];
                const bool rowpass = (dir == 0 || dir == 2);
                const int len = rowpass ? w : h, lines = rowpass ? h : w;
                const int nseg = (len + p.local_prop_width - 1) / p.local_prop_width;
                hipLaunchKernelGGL(k_local_prop, dim3((lines + 63) / 64, nseg), dim3(64), 0, c->stream, I, dir, p.local_prop_width);
            }
        }
    }
    const int nblk = gpx.x * gpx.y;
    if (int e = c->rig_partial.reserve(sizeof(float) * (size_t)nblk * MAX_FRAMES)) return e;
    if (int e = c->cams.reserve(sizeof(CamState) * MAX_FRAMES)) return e;
    hipLaunchKernelGGL(k_update_rigidness, gpx, bpx, 0, c->stream, I, c->rig_partial.as<float>());
    if (p.N > 0)
        hipLaunchKernelGGL(k_reduce_density, dim3(p.N), dim3(256), 0, c->stream, c->rig_partial.as<float>(), nblk, w * h,
                           c->cams.as<CamState>());
    VK_CHECK_LAST();
    if (c->prof) prof_end(c, "optimize_depth");
    return 0;
}

// compute_cost_map alone (tests / parity probes)
int cost_map_device(Context* c, ImageSet& S, const OdParams& p) {
    if (int e = S.cost.reserve(sizeof(float) * (size_t)p.w * p.h)) return e;
    Img I = make_img(S, p);
    hipLaunchKernelGGL(k_cost_rand, dim3((p.w + 63) / 64, (p.h + 3) / 4), dim3(256), 0, c->stream, I, 0, 0u, p.range_factor);
    VK_CHECK_LAST();
    return 0;
}

// ---- small per-pixel helpers used by the host pipeline -------------------------------------
__global__ static void k_fill(float* p, float v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ static void k_scale(float* p, const float* s_dev, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] *= *s_dev;
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
    out[i] = s / (float)(n_flows + n_dp);
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

}  // namespace vk
