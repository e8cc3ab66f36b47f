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
#include <cstdlib>

namespace vk {

// the kernels templated on the frame bound, and optimize_depth_launch<NMAX, STRICT>, are compiled in vk_depth_i*.hip / vk_depth_s*.hip (vk_depth_impl.hpp)
template <int NMAX, bool STRICT> int optimize_depth_launch(Context* c, ImageSet& S, const OdParams& p, bool cost_only);

// voldor.cpp:309-317: scale = n / sum ||t_i|| over the registered frames (frames dropped by this iteration's decision do not count)
__device__ __forceinline__ static float world_scale_factor(const PoseBlock* P, int n_launch, const float (*ts)[3]) {
    const int n = min(n_launch, P->n_active);
    if (n <= 0) return 1.f;  // window lost: nothing to normalise (deviation D6)
    float ws = 0.f;
    for (int i = 0; i < n; i++) {
        const float* t = ts[i];
        ws = (float)((double)ws + sqrt((double)t[0] * t[0] + (double)t[1] * t[1] + (double)t[2] * t[2]));  // float += double (cv::norm, voldor.cpp:312)
    }
    return (float)n / ws;
}
// world_scale (may be NULL): the factor of normalize_world_scale (voldor.cpp:309-317), n / sum ||t_i|| over the registered frames, from
// the poses this optimize_depth call runs with; the E-step kernel stores the scaled depth, k_reduce_density then scales the poses.
__device__ __forceinline__ static void cum_poses_block(PoseBlock* P, int N, int N_dp, float* world_scale) {
    __shared__ double Rc[9], tc[3];
    __shared__ float sR[MAX_FRAMES][9], sT[MAX_FRAMES][3];  // one round trip to the pose block instead of one per frame of the chain
    const int l = threadIdx.x;
    for (int i = l; i < N * 9; i += 64) sR[i / 9][i % 9] = P->Rs[i / 9][i % 9];
    for (int i = l; i < N * 3; i += 64) sT[i / 3][i % 3] = P->ts[i / 3][i % 3];
    const double fx = P->K4[0], cx = P->K4[1], fy = P->K4[2], cy = P->K4[3];
    __syncthreads();
    auto emit = [&](const double* R, const double* t, float* M, float* T) {  // K R K^-1 and K t
        if (l < 9) {
            const int r = l / 3, c = l % 3;
            double kr[3];
            for (int j = 0; j < 3; j++) kr[j] = r == 0 ? fx * R[j] + cx * R[6 + j] : (r == 1 ? fy * R[3 + j] + cy * R[6 + j] : R[6 + j]);
            const float v = (float)(c == 0 ? kr[0] / fx : (c == 1 ? kr[1] / fy : kr[2] - kr[0] * cx / fx - kr[1] * cy / fy));
            M[l] = v;
        } else if (l < 12) {
            const int r = l - 9;
            const float v = (float)(r == 0 ? fx * t[0] + cx * t[2] : (r == 1 ? fy * t[1] + cy * t[2] : t[2]));
            T[r] = v;
        }
    };
    for (int f = 0; f < N; f++) {
        const float* R = sR[f]; const float* t = sT[f];
        double nv = 0.0;
        if (l < 9) {
            const int r = l / 3, c = l % 3;
            nv = f == 0 ? (double)R[l] : (double)R[r * 3] * Rc[c] + (double)R[r * 3 + 1] * Rc[3 + c] + (double)R[r * 3 + 2] * Rc[6 + c];
        } else if (l < 12) {
            const int r = l - 9;
            nv = f == 0 ? (double)t[r] : (double)R[r * 3] * tc[0] + (double)R[r * 3 + 1] * tc[1] + (double)R[r * 3 + 2] * tc[2] + (double)t[r];
        }
        __syncthreads();
        if (l < 9) Rc[l] = nv; else if (l < 12) tc[l - 9] = nv;
        __syncthreads();
        emit(Rc, tc, P->cumM[f], P->cumT[f]);
    }
    for (int f = 0; f < N_dp; f++) {
        double R[9], t[3];
        for (int k = 0; k < 9; k++) R[k] = P->dpRs[f][k];
        for (int k = 0; k < 3; k++) t[k] = P->dpts[f][k];
        emit(R, t, P->dpM[f], P->dpT[f]);
    }
    if (l == 0) {
        int ident = 0;
        for (int f = 0; f < N_dp; f++) {
            bool id = true;
            for (int k = 0; k < 9; k++) id = id && P->dpRs[f][k] == ((k % 4 == 0) ? 1.f : 0.f);
            for (int k = 0; k < 3; k++) id = id && P->dpts[f][k] == 0.f;
            ident |= id ? (1 << f) : 0;
        }
        P->dp_ident = ident;
    }
    if (world_scale && l == 0) *world_scale = world_scale_factor(P, N, sT);
}
__global__ __launch_bounds__(64) static void k_cum_poses(PoseBlock* P, int N, int N_dp, float* world_scale) { cum_poses_block(P, N, N_dp, world_scale); }

// fixed-order second stage: cams[f].pose_rigidness_density = sum(partial[f][:]) / npx
// blocks 0..n_launch-1: rigidness density of one frame; block n_launch (only launched when scale_out != NULL): the pose half
// of normalize_world_scale (voldor.cpp:309-317), scale = n / sum ||t_i|| over the registered frames -- one launch for both
__global__ static void k_reduce_density(const float* __restrict__ partial, int nblk, int npx, CamState* cams, PoseBlock* P, int n_launch,
                                        float* scale_out, int scale_ready) {
    const int f = blockIdx.x;
    if (f == n_launch) {
        if (threadIdx.x != 0) return;
        const int n = min(n_launch, P->n_active);  // frames dropped by this iteration's decision do not count
        // scale_ready: the factor was computed from these poses at the start of the call (cum_poses_block) and the depth map already
        // carries it (k_update_rigidness_lean); otherwise it is computed here and k_scale follows
        const float s = scale_ready ? *scale_out : world_scale_factor(P, n_launch, P->ts);
        for (int i = 0; i < n; i++)
            for (int d = 0; d < 3; d++) { P->ts[i][d] *= s; cams[i].t[d] = P->ts[i][d]; }
        *scale_out = s;
        return;
    }
    if (f >= P->n_active) return;
    __shared__ float s[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) acc += partial[(size_t)f * nblk + i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) cams[f].pose_rigidness_density = ((s[0] + s[1]) + (s[2] + s[3])) / (float)npx;
}

// ---- forward-backward smoothing (fb_smooth.h:26-70) -------------------------------------
// A line (row or column) is a serial recurrence of 640 / 480 steps and there are only N*h (N*w) lines:
// run one lane per line, as the reference does, and the pass is a few dozen waves each walking a ~1300
// step dependent chain -- 100 us on a chip that is 99 % idle.  Two things remove that:
//
// (1) one step is a projective-linear map.  Forward (fb_smooth.h:27-36):
//         s0 = (x(1-p) + (1-x)p) e0 ;  s1 = (xp + (1-x)(1-p)) e ;  x' = s1 / (s0 + s1)
//     is  (a1,a0)' = diag(e,e0) T (a1,a0),  x = a1/(a1+a0),  T = [[p,q],[q,p]],  q = 1-p;  backward
//     (:37-46) is  (b1,b0)' = T diag(e,e0) (b1,b0).  Maps compose, so a line is cut into segments of
//     <= FB_SEG steps, one LANE per segment: each lane multiplies up the 2x2 matrix of its segment
//     (all entries positive: no cancellation; renormalised every 8 steps), the segment matrices of a
//     line are chained through LDS (<= a few dozen 2x2 applications), and every lane then re-walks its
//     own segment from the now-known incoming message.  Dependent chain: ~2*FB_SEG steps instead of
//     2*w, on 16x more lanes.
// (2) the re-walk uses the Moebius form of the step,  x' = (c1 + c2 x) / (c3 + c4 x), whose coefficients
//     depend on the emission only and sit off the chain: fma -> v_rcp_f32 -> mul.
//
// A segment lives in registers (emissions + forward messages), so each map is read once and written
// once per pass and the forward-message scratch of the reference (fb_smooth.h:14-15) is not needed.
// Rounding differs from the step-by-step evaluation by a few ulp per step (the recurrence contracts,
// nothing accumulates): deviation D7 in DESIGN.md, stage parity test_fb_smooth_alone_matches_oracle.
// Steps per lane: template parameter FB_SEG of everything below (multiple of 4: 16-byte row accesses).  20 where the line fits
// its workgroup (rows up to 5120 pixels, columns up to 1280): the row pass of a 640x480 x 5 window is only 600 waves and its time is
// the dependent chain of 2 x FB_SEG steps (40 -> 20: 15.6 -> 12.5 us per pass); 40 for lines up to twice that (2560x1440, 4K,
// portrait 1080x1920); beyond (10240 x 2560) the pass falls back to one lane per line (fb_smooth_strict_device: any size).
struct FbCoef { float p, q, dd, e0, e0p, e0dd, qe0, pqe0, pq; };
__device__ __forceinline__ FbCoef fb_coef(float e0, float p) {
    FbCoef k;
    k.p = p; k.q = 1.f - p; k.dd = p - k.q; k.e0 = e0; k.e0p = e0 * p; k.e0dd = e0 * k.dd; k.qe0 = k.q * e0; k.pq = p + k.q;
    k.pqe0 = k.pq * e0;
    return k;
}
struct FbMat { float a, b, c, d; };  // acts on (x, 1-x): x' = (a x + b (1-x)) / ((a+c) x + (b+d)(1-x))
__device__ __forceinline__ float fb_apply(const FbMat& M, float x) {
    const float y = 1.f - x, n1 = M.a * x + M.b * y, n0 = M.c * x + M.d * y;
    return n1 * fast_rcp(n1 + n0);
}
// segment matrices: F = A_{n-1} ... A_0 with A_t = diag(e_t, e0) T ; B = C_0 ... C_{n-1} with C_t = T diag(e_t, e0)
template <int FB_SEG>
__device__ __forceinline__ void fb_compose(const FbCoef& K, const float (&e)[FB_SEG], int n, FbMat& F, FbMat& B) {
    F = { 1.f, 0.f, 0.f, 1.f }; B = { 1.f, 0.f, 0.f, 1.f };
#pragma unroll
    for (int k = 0; k < FB_SEG; k++) {
        if (k < n) {
            const float e1 = e[k];
            const float fa = (K.p * F.a + K.q * F.c) * e1, fb = (K.p * F.b + K.q * F.d) * e1;
            const float fc = (K.q * F.a + K.p * F.c) * K.e0, fd = (K.q * F.b + K.p * F.d) * K.e0;
            F = { fa, fb, fc, fd };
            const float ba = (B.a * K.p + B.b * K.q) * e1, bb = (B.a * K.q + B.b * K.p) * K.e0;
            const float bc = (B.c * K.p + B.d * K.q) * e1, bd = (B.c * K.q + B.d * K.p) * K.e0;
            B = { ba, bb, bc, bd };
            if ((k & 7) == 7) {
                const float sf = fast_rcp((F.a + F.b) + (F.c + F.d)), sb = fast_rcp((B.a + B.b) + (B.c + B.d));
                F = { F.a * sf, F.b * sf, F.c * sf, F.d * sf };
                B = { B.a * sb, B.b * sb, B.c * sb, B.d * sb };
            }
        }
    }
}
// re-walk of one segment: forward messages, then backward messages fused with the posterior (:65-69);
// e[] is overwritten with the smoothed values
template <int FB_SEG>
__device__ __forceinline__ void fb_walk(const FbCoef& K, float (&e)[FB_SEG], int n, float xf, float xb) {
    float Fm[FB_SEG];
#pragma unroll
    for (int k = 0; k < FB_SEG; k++) {
        Fm[k] = 0.f;
        if (k < n) {
            const float c1 = e[k] * K.q, c2 = e[k] * K.dd;
            xf = fmaf(c2, xf, c1) * fast_rcp(fmaf(c2 - K.e0dd, xf, K.e0p + c1));
            Fm[k] = xf;
        }
    }
#pragma unroll
    for (int k = FB_SEG - 1; k >= 0; k--) {
        if (k < n) {
            xb = fmaf(e[k] * K.p - K.qe0, xb, K.qe0) * fast_rcp(fmaf(K.pq * e[k] - K.pqe0, xb, K.pqe0));
            const float a1 = Fm[k] * xb, a0 = (1.f - Fm[k]) * (1.f - xb);
            e[k] = a1 * fast_rcp(a0 + a1);
        }
    }
}
// incoming messages of segment `seg` of a line whose segment matrices are sF[i*stride], sB[i*stride], i < S, chained step by step (column pass)
__device__ __forceinline__ void fb_incoming(const FbMat* sF, const FbMat* sB, int stride, int seg, int S, float first, float last,
                                            float& xf, float& xb) {
    xf = first;  // the chains start from the raw end values (fb_smooth.h:28, :38)
    for (int i = 0; i < seg; i++) xf = fb_apply(sF[i * stride], xf);
    xb = last;
    for (int i = S - 1; i > seg; i--) xb = fb_apply(sB[i * stride], xb);
}
// Incoming messages of every segment of every line of the workgroup.  Lane (line, seg) needs F_{seg-1} .. F_0 applied to the line's first
// value and B_{seg+1} .. B_{S-1} applied to its last one: chained lane by lane that is S - 1 dependent Moebius steps per lane (and, the
// lanes of a wave covering all segments, 2 (S - 1) steps of issue per wave: at 1920 wide more than the segments themselves).  The maps
// compose, so the prefix / suffix products come from a Hillis-Steele scan over the segment matrices in LDS instead: ceil(log2 S) rounds,
// each one 2x2 product per direction (left factor = the later segments), renormalised (the entries are products of probabilities),
// double buffered -> one barrier per round.  ALL threads of the workgroup call it (barriers); `valid` = the thread owns a segment slot.
// sF / sB: [2][nt]; on entry buffer 0 holds the segment matrices (written by the caller, barrier included); stride: distance between
// consecutive segments of a line in the thread index.  Row pass (rows 16.2 -> 12.8 us at 1241x376, 17.9 -> 11.8 on one 1080p map); in the
// column pass the same scan did not pay (15.4 -> 16.3 us): it keeps the chain.
__device__ __forceinline__ FbMat fb_mul(const FbMat& M, const FbMat& N) {  // M after N
    FbMat r = { M.a * N.a + M.b * N.c, M.a * N.b + M.b * N.d, M.c * N.a + M.d * N.c, M.c * N.b + M.d * N.d };
    const float sc = fast_rcp((r.a + r.b) + (r.c + r.d));
    return { r.a * sc, r.b * sc, r.c * sc, r.d * sc };
}
__device__ __forceinline__ void fb_incoming_scan(FbMat* sF, FbMat* sB, int nt /* threads: sF / sB are [2][nt] */, FbMat f, FbMat b, bool valid, int tid, int stride,
                                                 int seg, int S, float first, float last, float& xf, float& xb) {
    int cur = 0;
    for (int d = 1; d < S; d <<= 1) {
        if (valid) {
            if (seg >= d) f = fb_mul(f, sF[cur * nt + tid - d * stride]);
            if (seg + d < S) b = fb_mul(b, sB[cur * nt + tid + d * stride]);
            sF[(cur ^ 1) * nt + tid] = f; sB[(cur ^ 1) * nt + tid] = b;
        }
        __syncthreads();
        cur ^= 1;
    }
    xf = first; xb = last;  // the chains start from the raw end values (fb_smooth.h:28, :38)
    if (valid) {
        if (seg > 0) xf = fb_apply(sF[cur * nt + tid - stride], first);
        if (seg < S - 1) xb = fb_apply(sB[cur * nt + tid + stride], last);
    }
}

// Row pass: thread = (row, segment), segments of a row on adjacent lanes -> a wave reads whole contiguous row
// pieces (160 bytes per lane).  256 threads = floor(256/S) rows.
struct __attribute__((packed, aligned(4))) FbQuad { float x, y, z, w; };  // 16 bytes at 4-byte alignment: one global_load_dwordx4 on gfx950
template <bool VEC4, int FB_SEG>
__global__ __launch_bounds__(256) static void k_fb_rows(const float* maps, float* maps_out /* == maps: in place */, int w, int h, int S, float e0, float p, const int* __restrict__ n_dev,
                                                        int n_maps, PoseBlock* cumP, int cumN, int cumNdp, float* world_scale) {
    __shared__ FbMat sF[2][256], sB[2][256];
    if ((int)blockIdx.y == n_maps) {  // extra layer (launched only with cumP): one workgroup prepares the projective maps the cost kernels need next
        if (blockIdx.x == 0) cum_poses_block(cumP, cumN, cumNdp, world_scale);
        return;
    }
    if (n_dev && (int)blockIdx.y >= *n_dev) return;  // map of a frame the device-side decision has dropped
    const int lpb = 256 / S, tid = threadIdx.x;
    const int ll = tid / S, seg = tid - ll * S, row = blockIdx.x * lpb + ll;
    const bool live = ll < lpb && row < h;
    const int c0 = seg * FB_SEG, n = live ? min(FB_SEG, w - c0) : 0;
    const size_t line_off = (size_t)blockIdx.y * w * h + (size_t)(live ? row : 0) * w;
    const float* m = maps + line_off;
    const FbCoef K = fb_coef(e0, p);
    float e[FB_SEG];
    if (VEC4) {
#pragma unroll
        for (int k = 0; k < FB_SEG / 4; k++) {
            float4 t = make_float4(0.5f, 0.5f, 0.5f, 0.5f);
            if (4 * k < n) t = *reinterpret_cast<const float4*>(m + c0 + 4 * k);
            e[4 * k] = t.x; e[4 * k + 1] = t.y; e[4 * k + 2] = t.z; e[4 * k + 3] = t.w;
        }
    } else {  // rows that do not start on 16 bytes (w % 4 != 0): the same 16-byte accesses with 4-byte alignment, scalar for a ragged tail
#pragma unroll
        for (int k = 0; k < FB_SEG / 4; k++) {
            if (4 * k + 3 < n) {
                const FbQuad t = *reinterpret_cast<const FbQuad*>(m + c0 + 4 * k);
                e[4 * k] = t.x; e[4 * k + 1] = t.y; e[4 * k + 2] = t.z; e[4 * k + 3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) e[4 * k + j] = (4 * k + j < n) ? m[c0 + 4 * k + j] : 0.5f;
            }
        }
    }
    const float first = m[0], last = m[w - 1];
    FbMat F, B;
    fb_compose<FB_SEG>(K, e, n, F, B);
    sF[0][tid] = F; sB[0][tid] = B;
    __syncthreads();
    float xf, xb;
    fb_incoming_scan(&sF[0][0], &sB[0][0], 256, F, B, ll < lpb, tid, 1, seg, S, first, last, xf, xb);
    if (!live) return;
    fb_walk<FB_SEG>(K, e, n, xf, xb);
    float* mo = maps_out + line_off;  // every value of the line this lane reads was read above: in place or not makes no difference to the result
    if (VEC4) {
#pragma unroll
        for (int k = 0; k < FB_SEG / 4; k++)
            if (4 * k < n) *reinterpret_cast<float4*>(mo + c0 + 4 * k) = make_float4(e[4 * k], e[4 * k + 1], e[4 * k + 2], e[4 * k + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < FB_SEG / 4; k++) {
            if (4 * k + 3 < n) *reinterpret_cast<FbQuad*>(mo + c0 + 4 * k) = FbQuad{ e[4 * k], e[4 * k + 1], e[4 * k + 2], e[4 * k + 3] };
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) if (4 * k + j < n) mo[c0 + 4 * k + j] = e[4 * k + j];
            }
        }
    }
}
// Column pass: thread = (segment, column); FB_CW adjacent columns share a workgroup, so every access is a
// contiguous 64-byte row piece and a workgroup is FB_CW * S threads (S <= 64).
constexpr int FB_CW = 16;
template <int FB_SEG>
__global__ __launch_bounds__(1024) static void k_fb_cols(float* __restrict__ maps, int w, int h, int S, float e0, float p, const int* __restrict__ n_dev) {
    __shared__ FbMat sF[1024], sB[1024];
    if (n_dev && (int)blockIdx.y >= *n_dev) return;
    const int tid = threadIdx.x, seg = tid / FB_CW, cl = tid - seg * FB_CW, col = blockIdx.x * FB_CW + cl;
    const bool live = col < w;
    const int r0 = seg * FB_SEG, n = live ? min(FB_SEG, h - r0) : 0;
    float* m = maps + (size_t)blockIdx.y * w * h + (live ? col : 0);
    const FbCoef K = fb_coef(e0, p);
    float e[FB_SEG];
#pragma unroll
    for (int k = 0; k < FB_SEG; k++) e[k] = (k < n) ? m[(size_t)(r0 + k) * w] : 0.5f;
    const float first = m[0], last = m[(size_t)(h - 1) * w];
    FbMat F, B;
    fb_compose<FB_SEG>(K, e, n, F, B);
    sF[tid] = F; sB[tid] = B;
    __syncthreads();
    if (!live) return;
    float xf, xb;
    fb_incoming(sF + cl, sB + cl, FB_CW, seg, S, first, last, xf, xb);  // (the scan of the row pass does not pay here: measured 15.4 -> 16.3 us at 1241x376)
    fb_walk<FB_SEG>(K, e, n, xf, xb);
#pragma unroll
    for (int k = 0; k < FB_SEG; k++) if (k < n) m[(size_t)(r0 + k) * w] = e[k];
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
constexpr int FB_MAX_ROW_SEGS = 256, FB_MAX_COL_SEGS = 1024 / FB_CW;
bool fb_smooth_segmented(int w, int h) { return !(w > 40 * FB_MAX_ROW_SEGS || h > 40 * FB_MAX_COL_SEGS); }
// dst (optional): the smoothed maps go THERE and `maps` stays as it is (the row pass writes dst, the column pass runs in place on dst) -- only with
// fb_smooth_segmented(w, h); st (optional): the stream of the two launches instead of the context's own
int fb_smooth_device(Context* c, float* maps, int n_maps, int w, int h, float s0_ems_prob, float no_change_prob, const int* n_dev, PoseBlock* cumP,
                     int cumN, int cumNdp, float* world_scale, float* dst, hipStream_t st) {
    if (n_maps <= 0) return 0;
    if (!st) st = c->stream;
    if (!fb_smooth_segmented(w, h)) {
        if (dst || st != c->stream) return (int)hipErrorInvalidValue;
        // larger than any segmented launch: one lane per line walks the recurrence step by step (the reference's own structure,
        // fb_smooth.h:26-69; no size limit).  The projective maps of the cost kernels then need their own launch.
        if (cumP) hipLaunchKernelGGL(k_cum_poses, dim3(1), dim3(64), 0, c->stream, cumP, cumN, cumNdp, world_scale);
        return fb_smooth_strict_device(c, maps, n_maps, w, h, s0_ems_prob, no_change_prob, n_dev);
    }
    // Steps per lane.  A lane also chains the S - 1 segment matrices of its line up to its own segment (fb_incoming): S - 1 Moebius steps
    // next to the 2 x FB_SEG of its segment.  In the latency regime (one or two waves per SIMD: 640x480, 1241x376) short segments
    // win -- the dependent chain is what takes the time.  Where the pass fills the chip many times over it is bound by VALU issue (0.91
    // at 1080p) and the chaining is half of all instructions with 20-step segments (1920 wide: 95 + 40 steps per lane): 40-step
    // segments then do the same work in 37 % fewer instructions.
    const int forced = debug_switches().fb_segment;
    const bool many_waves = (size_t)w * h * n_maps >= ((size_t)8 << 20);  // >= 8 waves per SIMD at 20 steps per lane
    const bool rows40 = w > 20 * FB_MAX_ROW_SEGS || (forced ? forced == 40 : many_waves);
    const bool cols40 = h > 20 * FB_MAX_COL_SEGS || (forced ? forced == 40 : many_waves);
    float* out = dst ? dst : maps;
    if (!rows40) fb_rows_launch<20>(st, maps, out, n_maps, w, h, s0_ems_prob, no_change_prob, n_dev, cumP, cumN, cumNdp, world_scale);
    else fb_rows_launch<40>(st, maps, out, n_maps, w, h, s0_ems_prob, no_change_prob, n_dev, cumP, cumN, cumNdp, world_scale);
    if (!cols40) fb_cols_launch<20>(st, out, n_maps, w, h, s0_ems_prob, no_change_prob, n_dev);
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
    hipLaunchKernelGGL(k_reduce_density, dim3(n_launch + (scale_out ? 1 : 0)), dim3(256), 0, c->stream, partial, nblk, npx, c->cams.as<CamState>(), P, n_launch, scale_out, scale_ready);
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

}  // namespace vk
