// voldor_amd/csrc/vk_strict.hip -- "strict math" variants of the stages whose fast kernels re-associate arithmetic.
//
// Strict mode (config key --strict_math 1, or vk_set_strict_math) exists to PIN parity: every stage evaluates the
// reference's expressions in the reference's order (the order the CPU oracle restates, file:line below), transcendentals come
// from vk_strict_math.h on both sides, so a whole window of the HIP pipeline can be compared with the oracle bit for bit
// (tests/test_gpu_strict.py) instead of "within the estimator's sampling noise".  Speed is secondary here (a window takes a few
// times longer than in the fast mode); the fast kernels are then held to the strict ones by measured distances.
//   * fb_smooth: one lane per line, the step-by-step recurrence of gpu-kernels/fb_smooth.h:26-69 (the fast kernels compose
//     projective maps and use v_rcp_f32, deviation D7)
//   * mode finding: mean-shift (meanshift.cu:34-150) and robust Gaussian (fit_robust_gaussian.cu:101-286) with every sum taken
//     in the order of the reference's shared-memory tree (reduce_vector_sum.h:12-61: blocks of 512 rows, thread t starts from
//     x[t] + x[t+256], strides 128..1, block sums form the next level), the 6x6 algebra of aux_funs.cpp:101-141 as a serial
//     fp64 LU on one lane, and the scalings of geometry.cpp:186-263 with cv::Mat's `/=` = multiply by (float)(1./s)
// The per-pixel depth kernels have their strict variants as template instantiations in vk_depth.hip.
#include "vk_common.hpp"
#include "vk_device.hpp"
#include "vk_p3p.hpp"
#include "vk_strict_math.h"
#include "vk_ref_cv.h"
#include "vk_lu.hpp"
#include "vk_internal.hpp"
#include <cstddef>

namespace vk {

// ---- fb_smooth, gpu-kernels/fb_smooth.h:26-69 -----------------------------------------------------------------------------
// pass 0: rows (line = row, stride 1), pass 1: columns (line = column, stride w).  F = forward messages of the line (scratch,
// same layout as the maps).  The backward recurrence and the posterior are fused: step i of the backward chain needs the raw
// e1[i], which is overwritten by the posterior only after it has been used.
// The chain of a line is serial (the next message needs the previous one: ~15 dependent operations and an IEEE division per step), so what a
// lane can do about time is to keep the MEMORY off that chain: the values of FBS_CH steps are loaded together, one chunk ahead of the
// arithmetic, and the results of a chunk are stored together -- one exposed memory round trip per line instead of one per step
// (293 -> ~60 us per pass on five 640x480 maps; the arithmetic, its order and its rounding are untouched).
constexpr int FBS_CH = 16;
struct __attribute__((packed, aligned(4))) FbsQuad { float x, y, z, w; };  // 16 bytes at 4-byte alignment: one global_load_dwordx4 on gfx950
// chunk [i0, i0 + FBS_CH) of a line into registers / back.  Row pass (PASS 0: a lane walks along x, lanes are w floats apart): 16-byte
// accesses -- a scalar access per step is 64 cache lines per wave instruction and the pass was bound by that, not by its chain; column
// pass (PASS 1): lanes sit on adjacent columns, every access is one coalesced row piece.
template <int PASS>
__device__ __forceinline__ void fbs_load(const float* __restrict__ p, size_t stride, int i0, int n, float (&v)[FBS_CH]) {
    if (PASS == 0) {
#pragma unroll
        for (int k = 0; k < FBS_CH; k += 4) {
            if (i0 >= 0 && i0 + k + 3 < n) { const FbsQuad q = *reinterpret_cast<const FbsQuad*>(p + i0 + k); v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w; }
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) { const int i = i0 + k + j; v[k + j] = (i >= 0 && i < n) ? p[i] : 0.f; }
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < FBS_CH; k++) { const int i = i0 + k; v[k] = (i >= 0 && i < n) ? p[(size_t)i * stride] : 0.f; }
    }
}
template <int PASS>
__device__ __forceinline__ void fbs_store(float* __restrict__ p, size_t stride, int i0, int n, const float (&v)[FBS_CH]) {
    if (PASS == 0) {
#pragma unroll
        for (int k = 0; k < FBS_CH; k += 4) {
            if (i0 + k + 3 < n) *reinterpret_cast<FbsQuad*>(p + i0 + k) = FbsQuad{ v[k], v[k + 1], v[k + 2], v[k + 3] };
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) if (i0 + k + j < n) p[i0 + k + j] = v[k + j];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < FBS_CH; k++) if (i0 + k < n) p[(size_t)(i0 + k) * stride] = v[k];
    }
}
__device__ __forceinline__ float fb_fwd_step(float prev, float e, float e0, float p) {  // FB_MSG_L2R / T2B (fb_smooth.h:27-36, :47-55)
#pragma clang fp contract(off)
    const float s0 = (prev * (1.f - p) + (1.f - prev) * p) * e0;
    const float s1 = (prev * p + (1.f - prev) * (1.f - p)) * e;
    return s1 / (s0 + s1);
}
__device__ __forceinline__ float fb_bwd_step(float prev, float e, float e0, float p) {  // FB_MSG_R2L / B2T (:37-46, :56-64)
#pragma clang fp contract(off)
    const float s0 = prev * e * (1.f - p) + (1.f - prev) * p * e0;
    const float s1 = prev * e * p + (1.f - prev) * (1.f - p) * e0;
    return s1 / (s0 + s1);
}
__device__ __forceinline__ float fb_posterior(float f, float b) {  // FB_POSTERIOR (:65-69)
#pragma clang fp contract(off)
    const float q0 = (1.f - f) * (1.f - b), q1 = f * b;
    return q1 / (q0 + q1);
}
// chunk loads / stores at ANY base (the backward chain's chunks are aligned to the top of the line, the last chunk of either may hang over an end)
template <int PASS>
__device__ __forceinline__ void fbs_load_any(const float* __restrict__ p, size_t stride, int base, int n, float (&v)[FBS_CH]) {
    if (base >= 0 || PASS != 0) { fbs_load<PASS>(p, stride, base, n, v); return; }
#pragma unroll
    for (int k = 0; k < FBS_CH; k++) { const int i = base + k; v[k] = (i >= 0 && i < n) ? p[i] : 0.f; }
}
template <int PASS>
__device__ __forceinline__ void fbs_store_range(float* __restrict__ p, size_t stride, int base, int lo, int hi, const float (&v)[FBS_CH]) {  // indices [lo, hi] of the chunk at `base`
    if (PASS == 0) {
#pragma unroll
        for (int k = 0; k < FBS_CH; k += 4) {
            if (base + k >= lo && base + k + 3 <= hi) *reinterpret_cast<FbsQuad*>(p + base + k) = FbsQuad{ v[k], v[k + 1], v[k + 2], v[k + 3] };
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) { const int i = base + k + j; if (i >= lo && i <= hi) p[i] = v[k + j]; }
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < FBS_CH; k++) { const int i = base + k; if (i >= lo && i <= hi) p[(size_t)i * stride] = v[k]; }
    }
}
// The forward chain (fb_smooth.h:27-36) and the backward chain (:37-46) of a line are INDEPENDENT of each other -- both read the raw
// map -- and only the posterior (:65-69) joins them.  The pass is a few dozen waves on an otherwise idle chip and each wave is bound by
// its own instruction issue (~60 instructions per index with two IEEE divisions), so the two chains of a line run on two WAVES of one
// workgroup (two SIMDs): wave 0 walks 64 lines forwards, wave 1 the same 64 lines backwards.  Phase 1: either chain covers its own
// half of the line, messages to scratch (F / B); one workgroup barrier; phase 2: either chain walks on through the OTHER half, where it
// finds the other chain's message at every index and stores the posterior (by then both chains have consumed the raw value there).
// (Both chains interleaved in ONE lane were measured first: slower, 160 -> 186 us -- the wave is issue bound, not latency bound.)
// Every value is computed by the operation sequence of the step-by-step form: same bits.
// One chain over steps [s0, s1) of its line: index = DIR ? n - 1 - step : step.  POST: the other chain's messages `other` are there ->
// posterior to the map; else this chain's messages to `mine`.
template <int PASS, int DIR, bool POST>
__device__ __forceinline__ float fb_chain(const float* e1, float* mine /* POST: where the posterior goes (the map itself, or the out-of-place destination) */, const float* __restrict__ other, size_t stride, int n, int s0, int s1,
                                          float prev, float e0, float p, bool live) {
#pragma clang fp contract(off)
    if (s0 >= s1) return prev;
    float buf[FBS_CH], oth[FBS_CH];
    const int first = DIR ? n - 1 - s0 : s0, last = DIR ? n - s1 : s1 - 1;  // first and last index of the walk (inclusive)
    int base = DIR ? first - (FBS_CH - 1) : first;                          // lowest index of the chunk the walk starts in
    fbs_load_any<PASS>(e1, stride, base, n, buf);
    if (POST) fbs_load_any<PASS>(other, stride, base, n, oth);
    for (;;) {
        const int nbase = DIR ? base - FBS_CH : base + FBS_CH;
        float nxt[FBS_CH], noth[FBS_CH], out[FBS_CH];
        fbs_load_any<PASS>(e1, stride, nbase, n, nxt);  // one chunk ahead of the arithmetic
        if (POST) fbs_load_any<PASS>(other, stride, nbase, n, noth);
#pragma unroll
        for (int kk = 0; kk < FBS_CH; kk++) {
            const int k = DIR ? FBS_CH - 1 - kk : kk, i = base + k;
            out[k] = 0.f;
            const bool in = DIR ? (i >= last && i <= first) : (i >= first && i <= last);  // (uniform: every line of a pass has n steps)
            if (in) {
                prev = DIR ? fb_bwd_step(prev, buf[k], e0, p) : fb_fwd_step(prev, buf[k], e0, p);
                out[k] = POST ? (DIR ? fb_posterior(oth[k], prev) : fb_posterior(prev, oth[k])) : prev;
            }
        }
        const int lo = max(base, DIR ? last : first), hi = min(base + FBS_CH - 1, DIR ? first : last);
        if (live) fbs_store_range<PASS>(mine, stride, base, lo, hi, out);  // (an idle lane of the last workgroup walks along on line 0 and stores nothing)
        if (DIR ? base <= last : base + FBS_CH - 1 >= last) break;
        base = nbase;
#pragma unroll
        for (int k = 0; k < FBS_CH; k++) { buf[k] = nxt[k]; if (POST) oth[k] = noth[k]; }
    }
    return prev;
}
template <int PASS>
__global__ __launch_bounds__(128) static void k_fb_strict(const float* maps, float* maps_out /* == maps: in place */, float* __restrict__ fwd, float* __restrict__ bwd, int w, int h, float e0, float p,
                                                          const int* __restrict__ n_dev) {
#pragma clang fp contract(off)
    if (n_dev && (int)blockIdx.y >= *n_dev) return;
    const int dir = threadIdx.x >> 6, l = blockIdx.x * 64 + (threadIdx.x & 63);
    const int n = PASS == 0 ? w : h, lines = PASS == 0 ? h : w;
    const size_t stride = PASS == 0 ? 1 : (size_t)w;
    const bool live = l < lines;
    const size_t base = (size_t)blockIdx.y * w * h + (PASS == 0 ? (size_t)(live ? l : 0) * w : (size_t)(live ? l : 0));
    const float* e1 = maps + base;
    float* eo = maps_out + base;  // (a chain stores a posterior only where BOTH chains have consumed the raw value: in place or not, the same values)
    float* F = fwd + base;
    float* B = bwd + base;
    const int H = n / 2;  // the forward chain's half is [0, H), the backward chain's [H, n)
    // every lane of the workgroup reaches the ONE barrier (no early return: an idle lane follows line 0 with its stores masked)
    float prev = dir == 0 ? e1[0] : e1[(size_t)(n - 1) * stride];  // the chains start from the raw end values (:28, :38)
    if (dir == 0) prev = fb_chain<PASS, 0, false>(e1, F, nullptr, stride, n, 0, H, prev, e0, p, live);
    else prev = fb_chain<PASS, 1, false>(e1, B, nullptr, stride, n, 0, n - H, prev, e0, p, live);
    __syncthreads();
    if (dir == 0) fb_chain<PASS, 0, true>(e1, eo, B, stride, n, H, n, prev, e0, p, live);
    else fb_chain<PASS, 1, true>(e1, eo, F, stride, n, n - H, n, prev, e0, p, live);
}
int fb_smooth_strict_device(Context* c, float* maps, int n_maps, int w, int h, float s0_ems_prob, float no_change_prob, const int* n_dev, float* dst, hipStream_t st_in) {
    if (n_maps <= 0) return 0;
    hipStream_t st = st_in ? st_in : c->stream;
    const size_t plane = (size_t)n_maps * w * h;
    if (int e = c->fb_scratch.reserve(sizeof(float) * 2 * plane)) return e;  // forward and backward messages
    float* F = c->fb_scratch.as<float>(); float* B = F + plane;
    float* out = dst ? dst : maps;
    hipLaunchKernelGGL(k_fb_strict<0>, dim3((h + 63) / 64, n_maps), dim3(128), 0, st, maps, out, F, B, w, h, s0_ems_prob, no_change_prob, n_dev);
    hipLaunchKernelGGL(k_fb_strict<1>, dim3((w + 63) / 64, n_maps), dim3(128), 0, st, out, out, F, B, w, h, s0_ems_prob, no_change_prob, n_dev);
    VK_CHECK_LAST();
    return 0;
}

// ---- sums in the reference's tree order ----------------------------------------------------------------------------------
constexpr int ST_THREADS = 256;  // = the reference's reduction block (reduce_vector_sum.h: 256 threads x 2 rows)
constexpr int ST_MAXBLK = 64;    // level-1 blocks: up to 32768 elements (the pipeline draws 8192 hypotheses)
template <int NV> struct TreeBuf { float s[NV][ST_THREADS]; float lvl[NV][ST_MAXBLK]; float out[NV]; };
// out[v] = sum over i < n of elem(i)[v], summed like the reference: per level, block b covers rows [512 b, 512 b + 512), thread t
// starts from x[t] + x[t+256] (rows beyond n contribute nothing), then a binary tree over strides 128..1; the block sums are the
// rows of the next level.  Called by all ST_THREADS threads; results in tb.out after the call (synchronised).
template <int NV, typename ElemFn>
__device__ __forceinline__ void tree_sum(int n, ElemFn elem, TreeBuf<NV>& tb) {
#pragma clang fp contract(off)
    const int t = threadIdx.x;
    const int nb = (n + 2 * ST_THREADS - 1) / (2 * ST_THREADS);
    if (n == 1) {  // the reference's loop `while (n > 1)` does not run: the single row is the result
        if (t == 0) { float v[NV]; elem(0, v); for (int k = 0; k < NV; k++) tb.out[k] = v[k]; }
        __syncthreads();
        return;
    }
    for (int b = 0; b < nb; b++) {
        const int idx = b * 2 * ST_THREADS + t;
        float v0[NV], v1[NV];
        const bool h0 = idx < n, h1 = idx + ST_THREADS < n;
        if (h0) elem(idx, v0);
        if (h1) elem(idx + ST_THREADS, v1);
#pragma unroll
        for (int k = 0; k < NV; k++) tb.s[k][t] = h0 ? (h1 ? v0[k] + v1[k] : v0[k]) : 0.f;
        __syncthreads();
        for (int stride = ST_THREADS / 2; stride >= 1; stride >>= 1) {
            if (t < stride) {
#pragma unroll
                for (int k = 0; k < NV; k++) tb.s[k][t] += tb.s[k][t + stride];
            }
            __syncthreads();
        }
        if (t < NV) tb.lvl[t][b] = tb.s[t][0];
        __syncthreads();
    }
    if (nb == 1) {
        if (t < NV) tb.out[t] = tb.lvl[t][0];
        __syncthreads();
        return;
    }
    // second level: nb (<= 64) rows, one block
#pragma unroll
    for (int k = 0; k < NV; k++) tb.s[k][t] = t < nb ? tb.lvl[k][t] : 0.f;  // t + 256 < nb never holds
    __syncthreads();
    for (int stride = ST_THREADS / 2; stride >= 1; stride >>= 1) {
        if (t < stride) {
#pragma unroll
            for (int k = 0; k < NV; k++) tb.s[k][t] += tb.s[k][t + stride];
        }
        __syncthreads();
    }
    if (t < NV) tb.out[t] = tb.s[t][0];
    __syncthreads();
}

// serial n x n inverse + determinant, double, LU with partial pivoting: aux_funs.cpp:101-118 (cv::determinant / Matx::inv);
// same operation order as the oracle's lu_inverse.  One lane.  Returns det; Ainv valid iff det > 0.
__device__ static double lu_inverse_serial(const double* A, double* Ainv, int n) {
#pragma clang fp contract(off)
    double a[36], b[36];
    for (int i = 0; i < n * n; i++) a[i] = A[i];
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) b[i * n + j] = (i == j) ? 1.0 : 0.0;
    double det = 1;
    for (int i = 0; i < n; i++) {
        int k = i;
        for (int j = i + 1; j < n; j++) if (fabs(a[j * n + i]) > fabs(a[k * n + i])) k = j;
        if (fabs(a[k * n + i]) < 2.220446049250313e-16) return 0;
        if (k != i) {
            for (int j = 0; j < n; j++) {
                double t = a[i * n + j]; a[i * n + j] = a[k * n + j]; a[k * n + j] = t;
                t = b[i * n + j]; b[i * n + j] = b[k * n + j]; b[k * n + j] = t;
            }
            det = -det;
        }
        det *= a[i * n + i];
        const double d = -1 / a[i * n + i];
        for (int j = i + 1; j < n; j++) {
            const double alpha = a[j * n + i] * d;
            for (int c = i + 1; c < n; c++) a[j * n + c] += alpha * a[i * n + c];
            for (int c = 0; c < n; c++) b[j * n + c] += alpha * b[i * n + c];
        }
    }
    if (det > 0) {
        for (int i = n - 1; i >= 0; i--)
            for (int c = 0; c < n; c++) {
                double s = b[i * n + c];
                for (int k = i + 1; k < n; k++) s -= a[i * n + k] * b[k * n + c];
                b[i * n + c] = s / a[i * n + i];
            }
        for (int i = 0; i < n * n; i++) Ainv[i] = b[i];
    }
    return det;
}

// (float)(1. / (double)b): what cv::Mat `/= b` multiplies by (OpenCV core/mat.inl.hpp) -- geometry.cpp:238,249, py_export.cpp:74
__host__ __device__ __forceinline__ float cv_div_scale(float b) { return (float)(1.0 / (double)b); }

// ---- per-camera mode finding in the reference's arithmetic (voldor/geometry.cpp:156-263) -------------------------------------
// ONE workgroup of 256 threads: ordered compaction of the finite hypotheses into `pool` (geometry.cpp:156-165), mean-shift,
// optionally the robust-Gaussian refit, unscaling, the pose into CamState / PoseBlock, the truncation decision.
struct StrictShared {
    TreeBuf<28> tb;
    int wcnt[4];
    int base;
    int flag;
    double full[36], inv[36];
    float cov[21], cinv[21], mean[6];
};
__global__ __launch_bounds__(ST_THREADS) static void k_pose_strict(const float* __restrict__ rvecs, const float* __restrict__ tvecs, int n_poses, ModeParams mp,
                                                                    CamState* cam, PoseBlock* P, int cam_idx, const int* __restrict__ n_points_dev,
                                                                    float* __restrict__ pool) {
#pragma clang fp contract(off)
    __shared__ StrictShared S;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (*n_points_dev < 4) {  // geometry.cpp:84-88
        if (t == 0) { cam->success = 0; maybe_decide(mp, P, cam, cam_idx); }
        return;
    }
    // ---- pool of finite hypotheses, in index order; rvec scaled for the mean-shift metric (:156-165, :191)
    if (t == 0) S.base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n_poses; i0 += ST_THREADS) {
        const int i = i0 + t;
        float v[6] = { 0, 0, 0, 0, 0, 0 };
        bool fin = false;
        if (i < n_poses) {
            // k_solve<.., FROM_MAP> leaves the pool as coordinate planes [3][n_poses] (vk_pose.hip)
            v[0] = rvecs[i]; v[1] = rvecs[(size_t)n_poses + i]; v[2] = rvecs[(size_t)2 * n_poses + i];
            v[3] = tvecs[i]; v[4] = tvecs[(size_t)n_poses + i]; v[5] = tvecs[(size_t)2 * n_poses + i];
            fin = isfinite(v[0] + v[1] + v[2] + v[3] + v[4] + v[5]);
        }
        const unsigned long long m = __ballot(fin);
        if (lane == 0) S.wcnt[wv] = __popcll(m);
        __syncthreads();
        int off = S.base;
        for (int k = 0; k < wv; k++) off += S.wcnt[k];
        if (fin) {
            const int r = off + __popcll(m & ((1ull << lane) - 1ull));
            pool[(size_t)r * 6] = v[0] * mp.rvec_scale; pool[(size_t)r * 6 + 1] = v[1] * mp.rvec_scale; pool[(size_t)r * 6 + 2] = v[2] * mp.rvec_scale;
            pool[(size_t)r * 6 + 3] = v[3]; pool[(size_t)r * 6 + 4] = v[4]; pool[(size_t)r * 6 + 5] = v[5];
        }
        __syncthreads();
        if (t == 0) S.base += S.wcnt[0] + S.wcnt[1] + S.wcnt[2] + S.wcnt[3];
        __syncthreads();
    }
    const int used = S.base;
    if (used == 0) {
        if (t == 0) { cam->success = 0; maybe_decide(mp, P, cam, cam_idx); }
        return;
    }
    __threadfence_block();
    __syncthreads();
    // ---- mean-shift, meanshift.cu:34-150 (host rand() of :76 -> rng3, as everywhere)
    const bool external_init = mp.use_external_init_mean < 0 ? (cam->pose_sample_count != 0) : (mp.use_external_init_mean != 0);
    float io_mean[6], c_mean[6];
    for (int d = 0; d < 3; d++) { io_mean[d] = cam->rvec[d] * mp.rvec_scale; io_mean[3 + d] = cam->t[d]; }
    const float two_var = 2 * mp.kernel_var;
    TreeBuf<28>& tb = S.tb;
    if (external_init) {
        for (int d = 0; d < 6; d++) c_mean[d] = io_mean[d];
    } else {
        float best = 0;
        int best_idx = -1;
        for (int trial = 0; trial < mp.ms_max_init_trials; trial++) {  // :75-95
            const int idx_rand = (int)(rng3(RAND_SEED, (uint32_t)trial, 0x4D53u) % (uint32_t)used);
            float c[6];
            for (int d = 0; d < 6; d++) c[d] = pool[(size_t)idx_rand * 6 + d];
            auto elem = [&](int i, float* v) {
                float l2 = 0;
                for (int d = 0; d < 6; d++) { const float df = pool[(size_t)i * 6 + d] - c[d]; l2 += df * df; }
                v[0] = vsm_expf(-l2 / two_var);
                for (int k = 1; k < 28; k++) v[k] = 0.f;
            };
            tree_sum<28>(used, elem, tb);
            const float wsum = tb.out[0];
            __syncthreads();
            if (wsum > best) { best = wsum; best_idx = idx_rand; }
            if (best > mp.ms_good_init_confidence * (float)used) break;
        }
        if (best_idx < 0) best_idx = 0;
        for (int d = 0; d < 6; d++) c_mean[d] = pool[(size_t)best_idx * 6 + d];
    }
    int ms_iters = 0;
    float conf = 0.f;
    for (int iter = 0; iter < mp.ms_max_iters; iter++) {  // :103-134
        auto elem = [&](int i, float* v) {
            float x[6], l2 = 0;
            for (int d = 0; d < 6; d++) { x[d] = pool[(size_t)i * 6 + d]; const float df = x[d] - c_mean[d]; l2 += df * df; }
            const float wgt = vsm_expf(-l2 / two_var);
            v[0] = wgt;
            for (int d = 0; d < 6; d++) v[1 + d] = x[d] * wgt;
            for (int k = 7; k < 28; k++) v[k] = 0.f;
        };
        tree_sum<28>(used, elem, tb);
        const float wsum = tb.out[0];
        float m[6];
        for (int d = 0; d < 6; d++) m[d] = tb.out[1 + d] / wsum;
        __syncthreads();
        conf = wsum / (float)used;
        ms_iters = iter + 1;
        float disp = 0;
        for (int d = 0; d < 6; d++) disp += (io_mean[d] - m[d]) * (io_mean[d] - m[d]);  // vs. the stale io mean on the 1st pass (SURVEY B-6)
        disp = sqrtf(disp);
        for (int d = 0; d < 6; d++) io_mean[d] = m[d];
        if (disp < mp.ms_epsilon) break;
        for (int d = 0; d < 6; d++) c_mean[d] = io_mean[d];
    }
    float pose_opm[6];
    for (int d = 0; d < 6; d++) pose_opm[d] = io_mean[d];
    float density = conf;
    int gu_iters = cam->last_used_gu_iters;
    float covar_out[36];
    bool write_covar = false;
    if (mp.do_rg) {  // geometry.cpp:201-246, fit_robust_gaussian.cu:101-286
        const float sc = mp.rg_pose_scaling;
        const int N = used;
        if (t < 21) S.cov[t] = 0.f;
        __syncthreads();
        if (t < 6) S.cov[(t * t + t) / 2 + t] = mp.kernel_var * (sc * sc);  // diag(var) *= sc*sc (:203-208)
        if (t < 6) S.mean[t] = pose_opm[t] * sc;
        __syncthreads();
        float weight = 0;
        int iter = 0;
        bool reliable = true;
        gu_iters = 0;  // fit_robust_gaussian.cu:158-159 resets *used_iters on entry
        for (iter = 0; iter < mp.rg_max_iters; iter++) {
            if (t == 0) {  // covar_half_to_full (:17-25), Ledoit-Wolf shrinkage (aux_funs.cpp:124-141), inverse (:101-118)
                for (int d1 = 0; d1 < 6; d1++)
                    for (int d2 = 0; d2 <= d1; d2++) {
                        S.full[d1 * 6 + d2] = (double)S.cov[(d1 * d1 + d1) / 2 + d2];
                        if (d1 != d2) S.full[d2 * 6 + d1] = S.full[d1 * 6 + d2];
                    }
                if (iter > 0 && mp.rg_covar_reg_lambda > 0) {
                    double tr = 0;
                    for (int d = 0; d < 6; d++) tr += S.full[d * 6 + d];
                    const double m = tr / (double)6, lam = (double)mp.rg_covar_reg_lambda;
                    for (int i = 0; i < 6; i++)
                        for (int j = 0; j < 6; j++) S.full[i * 6 + j] = lam * m * (i == j ? 1.0 : 0.0) + (1 - lam) * S.full[i * 6 + j];
                }
                const double det = lu_inverse_serial(S.full, S.inv, 6);
                S.flag = det <= 0 ? 2 : 0;
                if (det > 0)
                    for (int d1 = 0; d1 < 6; d1++)
                        for (int d2 = 0; d2 <= d1; d2++) {
                            S.cov[(d1 * d1 + d1) / 2 + d2] = (float)S.full[d1 * 6 + d2];
                            S.cinv[(d1 * d1 + d1) / 2 + d2] = (float)S.inv[d1 * 6 + d2];
                        }
            }
            __syncthreads();
            if (S.flag == 2) { reliable = false; break; }
            const float prev_density = weight / (float)N;
            float mean[6], cinv[21];
            for (int d = 0; d < 6; d++) mean[d] = S.mean[d];
            for (int k = 0; k < 21; k++) cinv[k] = S.cinv[k];
            auto elem = [&](int i, float* v) {  // e_step (:56-97)
                float x[6], diff[6];
                for (int d = 0; d < 6; d++) { x[d] = pool[(size_t)i * 6 + d] * sc; diff[d] = x[d] - mean[d]; }
                float z = 0;
                for (int d1 = 0; d1 < 6; d1++) {
                    float tmp = 0;
                    for (int d2 = 0; d2 < 6; d2++) {
                        const int hi = d1 >= d2 ? d1 : d2, lo = d1 >= d2 ? d2 : d1;
                        tmp += cinv[(hi * hi + hi) / 2 + lo] * diff[d2];
                    }
                    z += tmp * diff[d1];
                }
                z = sqrtf(z);
                const float wgt = z < mp.rg_trunc_sigma ? 1.f : 0.f;
                v[0] = wgt;
                for (int d = 0; d < 6; d++) v[1 + d] = wgt * x[d];
                for (int d1 = 0; d1 < 6; d1++)
                    for (int d2 = 0; d2 <= d1; d2++) v[7 + (d1 * d1 + d1) / 2 + d2] = wgt * diff[d1] * diff[d2];
            };
            tree_sum<28>(N, elem, tb);
            weight = tb.out[0];
            float nm[6], nc[21];
            for (int d = 0; d < 6; d++) nm[d] = tb.out[1 + d] / weight;
            for (int k = 0; k < 21; k++) nc[k] = tb.out[7 + k] / weight;
            __syncthreads();
            if (!isfinite(weight)) { reliable = false; break; }
            if (fabsf(weight / (float)N - prev_density) < mp.rg_epsilon) { reliable = true; break; }
            if (t < 6) S.mean[t] = nm[t];  // m step (:213-243)
            if (t < 21) S.cov[t] = nc[t];
            __syncthreads();
        }
        __syncthreads();
        if (reliable) {
            density = weight / (float)N; gu_iters = iter;
            const float isc2 = cv_div_scale(sc * sc);  // pose_covar /= sc*sc (cv::Mat, :224)
            for (int i1 = 0; i1 < 6; i1++)
                for (int i2 = 0; i2 < 6; i2++) {
                    const int hi = i1 >= i2 ? i1 : i2, lo = i1 >= i2 ? i2 : i1;
                    float cv = S.cov[(hi * hi + hi) / 2 + lo] * isc2;
                    if (i1 < 3 || i2 < 3) cv /= mp.rvec_scale;  // element-wise at<float>() /= (:226-233)
                    if (i1 < 3 && i2 < 3) cv /= mp.rvec_scale;
                    covar_out[i1 * 6 + i2] = cv;
                }
            for (int d = 0; d < 6; d++) pose_opm[d] = S.mean[d];
        } else {
            for (int k = 0; k < 36; k++) covar_out[k] = 0.f;
            for (int d = 0; d < 6; d++) pose_opm[d] = pose_opm[d] * sc;  // pose_opm *= sc (:210) stays as it went in
        }
        write_covar = true;
        const float isc = cv_div_scale(sc);  // pose_opm /= sc (:238)
        for (int d = 0; d < 6; d++) pose_opm[d] *= isc;
    }
    {
        const float irs = cv_div_scale(mp.rvec_scale);  // :249
        for (int d = 0; d < 3; d++) pose_opm[d] *= irs;
    }
    if (t == 0) {
        bool ok = true;
        for (int d = 0; d < 6; d++) ok = ok && isfinite(pose_opm[d]);  // checkRange :256
        cam->pose_sample_count = used;
        cam->pose_density = density;
        cam->last_used_ms_iters = ms_iters;
        cam->last_used_gu_iters = gu_iters;
        if (write_covar) for (int k = 0; k < 36; k++) cam->covar[k] = covar_out[k];
        cam->success = ok ? 1 : 0;
        if (ok) {
            for (int d = 0; d < 3; d++) { cam->t[d] = pose_opm[3 + d]; P->ts[cam_idx][d] = pose_opm[3 + d]; }
            float R[9];
            angle_axis_to_rotmat(pose_opm, R, true);
            for (int k = 0; k < 9; k++) P->Rs[cam_idx][k] = R[k];
            // the reference keeps the float matrix only; the vector the next mean shift starts from and the window returns is
            // Camera::rvec() = cv::Rodrigues(R) (utils.h:44-53, geometry.cpp:184): vk_ref_cv.h
            vrcv_rvec_of_R32(R, cam->rvec, 1);
        }
        maybe_decide(mp, P, cam, cam_idx);
    }
}


// ---- the same per-camera mode finding on a PARALLEL launch structure (round 4) ------------------------------------------------------
// k_pose_strict above walks the reference's sum tree block by block on 256 threads (16 blocks x 9 barriers per sum): 1.67 ms per camera,
// two thirds of a strict window.  The tree itself is parallel: reduce_vector_sum.h:12-61 starts thread t of block b from
// x[512 b + t] + x[512 b + t + 256] and then adds s[t] += s[t + stride] for strides 128 .. 1 -- an XOR BUTTERFLY over the nine index bits
// of a row inside its block, taken from the highest bit to the lowest (float addition commutes, so the lane that ends up with the sum
// does not matter), followed by the same tree over the <= 16 block sums.  So:
//   * 512 threads; wave w owns blocks 2w and 2w+1; lane l holds, per block, the eight rows 64 j + bitrev6(l), j = 0..7, in registers
//   * strides 256, 128, 64 pair rows INSIDE a lane (j <-> j+4, j+2, j+1); strides 32 .. 1 pair lanes -- with the rows dealt in
//     bit-reversed order the lane distances come out as 1, 2, .. 32, which is the order wave_reduce_transpose (vk_device.hpp) takes
//     them in: all 28 values of a block cross the wave in one transposing reduction (DPP moves, no LDS)
//   * the 16 block sums meet in LDS (double buffered: ONE workgroup barrier per sum) and lanes k < NV of every wave walk the second-level
//     tree of value k -- the reference's zero padding included: a block beyond the pool is 0.f, strides 128 .. 16 add 0.f (which turns a
//     -0.f into +0.f, nothing else), a partial block starts from (h0 ? (h1 ? v0 + v1 : v0) : 0.f)
//   * the 6x6 inverse is the row-per-lane LU of vk_lu.hpp on wave 0 (element updates independent within a pivot step: the bits of the
//     serial LU), not a serial fp64 chain on one lane
// Every sum is the reference's tree term by term: the same bits as k_pose_strict (vk_debug_switch "strict_plain" selects that one;
// tests/test_gpu_strict.py holds the two against each other and both against the reference's own kernels).
constexpr int SP_THREADS = 512, SP_WAVES = SP_THREADS / 64, SP_SLOTS = 16;  // rows per lane: 2 blocks x 8
constexpr int SP_MAX_POSES = SP_THREADS * SP_SLOTS;                        // 8192 = cfg.n_poses_to_sample default
struct StrictParShared {
    float lvl[2][16][32];  // [parity][block][value]
    float raw[32];         // the single row of a one-row pool (the reference's loop `while (n > 1)` does not run)
    int cnt[SP_SLOTS * SP_WAVES + 1];
    int flag;
    float cov[21], cinv[21], mean[6];
    float cent[28][6];  // centres of a batch of initial-mode trials
    // cooperative form, four waves per 512-row block (round 6): the partial sums of strides 128 and 64 cross waves here; the totals come back through `tot`
    float xch[4][28][64];
    float tot[32];
    int dead;
};
constexpr int COOP_WAVES = 4, COOP_THREADS = 64 * COOP_WAVES;  // a cooperative workgroup: one 512-row block of the pool on four waves, two rows per lane
// out[k] = sum over rows i < n of row(i)[k], k < NV, in the reference's tree order; rowfn(slot, v) = the NV values of this lane's row in
// slot `slot` (block 2 wv + slot / 8, row 64 (slot % 8) + bitrev6(lane) of it; called for every slot, rows beyond n are discarded).
// All threads call it; the result is in every thread.
// lane_total (optional): the total of value `lane` in lanes < NV of every wave (what the broadcast below reads) -- lets the caller finish
// per-value work (the refit's 27 divisions by the weight) in one lane per value instead of in every thread.
// COOPERATIVE form (k_pose_strict_par<true>): the pool's 16 blocks on 16 workgroups (round 6: of four waves each, two rows per lane -- tree_sum_par; rounds 4-5: single
// waves, eight rows per lane) (16 compute units instead of one: a refit
// iteration's pass over 8192 rows is ~130 instructions per row and was bound by the issue rate of ONE compute unit).  Workgroup b owns
// block b: the same in-lane pair sums and the same transposing wave reduction, then the block sums meet in GLOBAL memory (agent-scope
// atomic stores / loads), one grid barrier per sum (double buffered like the LDS form), and every workgroup walks the second-level tree
// itself -- every workgroup carries the whole control flow redundantly on identical numbers, so they agree on every branch.
struct CoopGlobal {
    // a block sum and the number of the sum it belongs to in ONE 64-bit word (value bits | epoch << 32): the store of the word is the arrival, the
    // reader that finds the epoch has the value -- no counter, no fence, no second round trip (tagged[.][.][.] = 0 never matches: epochs start at 1)
    unsigned long long tagged[2][16][32];
    unsigned long long tagged_raw[32];
    int used; unsigned err; unsigned max_polls;
    unsigned fallbacks;  // cameras the single-workgroup kernel had to take over (never reset by a launch: vk_debug_counter reads and clears it)
};
// The workgroups of the cooperative form meet once per sum, in the tagged block sums themselves (CoopGlobal): lane k (< NV) of every workgroup reads
// block sums k of all blocks until each carries this sum's epoch.  A slot is reused two sums later; a workgroup can be at most ONE sum ahead of the
// slowest PARTICIPANT (it needs everybody's sums to get on), so nobody overwrites a word that is still being waited for -- participants are the
// workgroups that own a block of the pool: the others leave right after reading the pool size (k_pose_strict_par), they would wait on words nobody
// waits for in return.  Round 4, first form: an arrival counter + release / acquire fences + the loads (three dependent round trips per
// sum); this form: one.
// FORWARD PROGRESS (round 5).  The launcher only takes this form when the whole grid fits the chip next to itself (occupancy query, pose_mode_strict_device):
// 16 workgroups of 256 threads and ~35 KB of LDS (the occupancy query is made with exactly that shape), so every one of them is dispatched as soon as a compute unit has room -- other kernels on the chip end,
// these are the only ones that wait.  The spin is still BOUNDED (2^20 polls, ~1 s; vk_debug_switch "strict_coop_max_polls" lowers it to force the path
// in tests): a workgroup that gives up raises G->err, every other one sees the flag in its own poll loop and leaves too, NOTHING of the camera record is
// written, and the single-workgroup kernel launched behind it (k_pose_strict_par<false>, gated on the flag: it returns at once when the flag is down)
// computes the camera from the same pool -- the same bits, never a failed pose because of a meeting.
__device__ __forceinline__ unsigned long long coop_pack(float v, unsigned epoch) { return (unsigned long long)__builtin_bit_cast(unsigned, v) | ((unsigned long long)epoch << 32); }
// the words p[t * stride], t < nb (<= 16), all read together until every one carries `epoch`; out[t] = its value (0 beyond nb).  false: gave up
// (own bound reached, or somebody else's flag seen) -- uniform over the wave.
__device__ __forceinline__ bool coop_wait(const unsigned long long* p, int stride, int nb, unsigned epoch, CoopGlobal* G, float (&out)[16]) {
    unsigned spins = 0;
    const unsigned bound = G->max_polls;
    for (;;) {
        unsigned long long w[16];
#pragma unroll
        for (int t = 0; t < 16; t++) w[t] = t < nb ? __hip_atomic_load(p + (size_t)t * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)epoch << 32);
        const unsigned err = __hip_atomic_load(&G->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool all = true;
#pragma unroll
        for (int t = 0; t < 16; t++) all = all && (unsigned)(w[t] >> 32) == epoch;
#pragma unroll
        for (int t = 0; t < 16; t++) out[t] = t < nb ? __builtin_bit_cast(float, (unsigned)w[t]) : 0.f;
        if (all) return true;
        if (err != 0u || ++spins > bound) {  // 1: a meeting was given up; 2: the writer has finished (nothing left to meet for)
            unsigned expect = 0u;
            (void)__hip_atomic_compare_exchange_strong(&G->err, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
template <int NV, bool COOP, typename RowFn>
__device__ __forceinline__ void tree_sum_par(int n, RowFn rowfn, StrictParShared& S, CoopGlobal* G, unsigned& sync_target, int& parity, float (&out)[NV],
                                             float* lane_total = nullptr, bool* dead = nullptr /* COOP: a meeting was given up (uniform); the caller leaves */) {
#pragma clang fp contract(off)
    constexpr int P = NV <= 8 ? 8 : 32;
    constexpr int NBLK_OWN = COOP ? 1 : 2;  // blocks per wave
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, q = (int)(__brev((unsigned)lane) >> 26);
    const int nb = (n + 511) / 512;
    if constexpr (COOP) {
        // FOUR WAVES PER BLOCK (round 6; rounds 4-5: one wave per block, eight rows per lane -- a pass of the refit was ~130 instructions x 8 rows on ONE wave, the
        // issue of which was half of a sum's time).  Wave v holds rows q + 64 v and q + 64 (v + 4) of the block: its pair sum s_v is the tree's stride-256 level;
        // stride 128 adds s_0 + s_2 and s_1 + s_3, stride 64 adds those two -- the same three additions on the same operands as the one-wave form's
        // (s0 + s2) + (s1 + s3), the operands crossing waves through LDS; wave 0 then carries on alone (strides 32 .. 1 in the wave, the block sum into the
        // tagged word, the meeting, the second level) and hands the totals back through LDS.
        const int blk = (int)blockIdx.x, base = blk * 512 + q;
        float sv[NV];
        {
            float v0[NV], v1[NV];
            rowfn(0, v0);
            rowfn(1, v1);
            const bool h0 = base + 64 * wv < n, h1 = base + 64 * wv + 256 < n;
            if (n == 1 && blk == 0 && lane == 0 && wv == 0) {
#pragma unroll
                for (int k = 0; k < NV; k++) __hip_atomic_store(&G->tagged_raw[k], coop_pack(v0[k], sync_target + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < NV; k++) sv[k] = h0 ? (h1 ? v0[k] + v1[k] : v0[k]) : 0.f;
        }
        if (wv >= 2) {
#pragma unroll
            for (int k = 0; k < NV; k++) S.xch[wv][k][lane] = sv[k];
        }
        __syncthreads();
        if (wv < 2) {  // stride 128: s_0 + s_2 (wave 0), s_1 + s_3 (wave 1)
#pragma unroll
            for (int k = 0; k < NV; k++) sv[k] = sv[k] + S.xch[wv + 2][k][lane];
        }
        if (wv == 1) {
#pragma unroll
            for (int k = 0; k < NV; k++) S.xch[1][k][lane] = sv[k];
        }
        __syncthreads();
        ++sync_target;  // the epoch of this sum
        bool gave_up = false;
        if (wv == 0) {
            float acc[P];
#pragma unroll
            for (int k = 0; k < NV; k++) acc[k] = sv[k] + S.xch[1][k][lane];  // stride 64
#pragma unroll
            for (int k = NV; k < P; k++) acc[k] = 0.f;
            const float mine = wave_reduce_transpose<P>(acc);  // strides 32 .. 1 of the rows = lane distances 1 .. 32
            const int slot = wave_slot<P>(lane);
            if (lane < P && slot < NV) __hip_atomic_store(&G->tagged[parity][blk][slot], coop_pack(mine, sync_target), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float tot = 0.f;
            if (lane < NV) {
                float got[16];
                const bool met = n == 1 ? coop_wait(&G->tagged_raw[lane], 0, 1, sync_target, G, got) : coop_wait(&G->tagged[parity][0][lane], 32, nb, sync_target, G, got);
                if (!met) gave_up = true;
                if (n == 1) tot = got[0];
                else if (nb == 1) tot = got[0];
                else {
                    float b[16];
#pragma unroll
                    for (int t = 0; t < 16; t++) b[t] = (t < nb ? got[t] : 0.f) + 0.f;  // strides 128 .. 16 of the second level add zeros
#pragma unroll
                    for (int st = 8; st >= 1; st >>= 1)
#pragma unroll
                        for (int t = 0; t < st; t++) b[t] = b[t] + b[t + st];
                    tot = b[0];
                }
                S.tot[lane] = tot;
            }
            gave_up = __ballot(gave_up) != 0ull;
            if (lane == 0) S.dead = gave_up ? 1 : 0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NV; k++) out[k] = S.tot[k];
        if (lane_total) *lane_total = lane < NV ? S.tot[lane] : 0.f;
        *dead = S.dead != 0;
        parity ^= 1;
        __syncthreads();  // (S.tot / S.dead / S.xch are free for the next sum)
    } else {
#pragma unroll
    for (int h = 0; h < NBLK_OWN; h++) {
        const int blk = COOP ? (int)blockIdx.x : 2 * wv + h, base = blk * 512 + q;
        float acc[P];
        // stride 256: rows (j, j + 4); stride 128: (j, j + 2); stride 64: (0, 1)
        auto pair = [&](int j, float (&s)[NV]) {
            float v0[NV], v1[NV];
            rowfn(h * 8 + j, v0);
            rowfn(h * 8 + j + 4, v1);
            const bool h0 = base + 64 * j < n, h1 = base + 64 * j + 256 < n;
            if (n == 1 && blk == 0 && lane == 0 && j == 0) {
#pragma unroll
                for (int k = 0; k < NV; k++) { if (COOP) __hip_atomic_store(&G->tagged_raw[k], coop_pack(v0[k], sync_target + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else S.raw[k] = v0[k]; }
            }
#pragma unroll
            for (int k = 0; k < NV; k++) s[k] = h0 ? (h1 ? v0[k] + v1[k] : v0[k]) : 0.f;
        };
        {
            float s0[NV], s2[NV];
            pair(0, s0); pair(2, s2);
#pragma unroll
            for (int k = 0; k < NV; k++) acc[k] = s0[k] + s2[k];
        }
        {
            float s1[NV], s3[NV];
            pair(1, s1); pair(3, s3);
#pragma unroll
            for (int k = 0; k < NV; k++) acc[k] = acc[k] + (s1[k] + s3[k]);
        }
#pragma unroll
        for (int k = NV; k < P; k++) acc[k] = 0.f;
        const float mine = wave_reduce_transpose<P>(acc);  // strides 32 .. 1 of the rows = lane distances 1 .. 32
        const int slot = wave_slot<P>(lane);
        if (lane < P && slot < NV) {
            if (COOP) __hip_atomic_store(&G->tagged[parity][blk][slot], coop_pack(mine, sync_target + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else S.lvl[parity][blk][slot] = mine;
        }
    }
    if (COOP) ++sync_target; else __syncthreads();  // COOP: the epoch of this sum; the meeting is in coop_wait below
    float tot = 0.f;
    if (lane < NV) {
        float got[16];
        if (COOP) {  // this is where the workgroups meet
            const bool met = n == 1 ? coop_wait(&G->tagged_raw[lane], 0, 1, sync_target, G, got) : coop_wait(&G->tagged[parity][0][lane], 32, nb, sync_target, G, got);
            if (!met) *dead = true;
        }
        auto at = [&](int t) { return COOP ? got[t] : S.lvl[parity][t][lane]; };
        if (n == 1) tot = COOP ? got[0] : S.raw[lane];
        else if (nb == 1) tot = at(0);
        else {
            float b[16];
#pragma unroll
            for (int t = 0; t < 16; t++) b[t] = (t < nb ? at(t) : 0.f) + 0.f;  // strides 128 .. 16 of the second level add zeros
#pragma unroll
            for (int st = 8; st >= 1; st >>= 1)
#pragma unroll
                for (int t = 0; t < st; t++) b[t] = b[t] + b[t + st];
            tot = b[0];
        }
    }
#pragma unroll
    for (int k = 0; k < NV; k++) out[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), k));
    if (lane_total) *lane_total = tot;
    parity ^= 1;
    if (COOP) *dead = __ballot(*dead) != 0ull;  // (one wave per workgroup: uniform from here on)
    }
}

// ordered compaction of the finite hypotheses into `pool` (geometry.cpp:156-165): counts per (slice of 512, wave), one prefix, ordered
// writes.  Called by all SP_THREADS threads of one workgroup; returns the pool size (0: nothing finite).
__device__ __forceinline__ int strict_compact_pool(const float* __restrict__ rvecs, const float* __restrict__ tvecs, int n_poses, float rvec_scale,
                                                   float* __restrict__ pool, StrictParShared& S) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    auto load6 = [&](int i, float (&v)[6]) {
        v[0] = rvecs[i]; v[1] = rvecs[(size_t)n_poses + i]; v[2] = rvecs[(size_t)2 * n_poses + i];
        v[3] = tvecs[i]; v[4] = tvecs[(size_t)n_poses + i]; v[5] = tvecs[(size_t)2 * n_poses + i];
    };
    unsigned finbits = 0;
#pragma unroll
    for (int k = 0; k < SP_SLOTS; k++) {
        const int i = k * SP_THREADS + t;
        bool fin = false;
        if (i < n_poses) { float v[6]; load6(i, v); fin = isfinite(v[0] + v[1] + v[2] + v[3] + v[4] + v[5]); }
        const unsigned long long m = __ballot(fin);
        if (lane == 0) S.cnt[k * SP_WAVES + wv] = __popcll(m);
        finbits |= fin ? (1u << k) : 0u;
    }
    __syncthreads();
    if (wv == 0) {  // exclusive prefix of the 128 counts
        const int a = S.cnt[lane], b = S.cnt[64 + lane];
        int ia = a, ib = b;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int ta = __shfl_up(ia, o, 64), tb = __shfl_up(ib, o, 64); if (lane >= o) { ia += ta; ib += tb; } }
        const int tota = __shfl(ia, 63, 64), totb = __shfl(ib, 63, 64);
        S.cnt[lane] = ia - a; S.cnt[64 + lane] = tota + ib - b;
        if (lane == 0) S.cnt[SP_SLOTS * SP_WAVES] = tota + totb;
    }
    __syncthreads();
    const int used = S.cnt[SP_SLOTS * SP_WAVES];
#pragma unroll
    for (int k = 0; k < SP_SLOTS; k++) {
        const int i = k * SP_THREADS + t;
        const bool fin = (finbits >> k) & 1u;
        const unsigned long long m = __ballot(fin);
        if (fin) {
            float v[6];
            load6(i, v);
            const int r = S.cnt[k * SP_WAVES + wv] + __popcll(m & ((1ull << lane) - 1ull));
            pool[(size_t)r * 6] = v[0] * rvec_scale; pool[(size_t)r * 6 + 1] = v[1] * rvec_scale; pool[(size_t)r * 6 + 2] = v[2] * rvec_scale;  // :191
            pool[(size_t)r * 6 + 3] = v[3]; pool[(size_t)r * 6 + 4] = v[4]; pool[(size_t)r * 6 + 5] = v[5];
        }
    }
    return used;
}
// first launch of the cooperative form: the pool, its size, and the tagged block sums back to zero
__global__ __launch_bounds__(SP_THREADS) static void k_pose_strict_compact(const float* __restrict__ rvecs, const float* __restrict__ tvecs, int n_poses, float rvec_scale,
                                                                            const int* __restrict__ n_points_dev, float* __restrict__ pool, CoopGlobal* G, unsigned max_polls) {
    __shared__ StrictParShared S;
    int used = 0;
    if (*n_points_dev >= 4) used = strict_compact_pool(rvecs, tvecs, n_poses, rvec_scale, pool, S);
    if (threadIdx.x == 0) { G->used = used; G->err = 0u; G->max_polls = max_polls; }
    // epochs restart at 1 with every launch of the mode kernel: no word of an earlier launch may carry one
    unsigned long long* tg = &G->tagged[0][0][0];
    for (int i = threadIdx.x; i < 2 * 16 * 32; i += SP_THREADS) tg[i] = 0ull;
    if (threadIdx.x < 32) G->tagged_raw[threadIdx.x] = 0ull;
}

template <bool COOP>
__global__ __launch_bounds__(COOP ? COOP_THREADS : SP_THREADS) static void k_pose_strict_par(const float* __restrict__ rvecs, const float* __restrict__ tvecs, int n_poses, ModeParams mp,
                                                                                   CamState* cam, PoseBlock* P, int cam_idx, const int* __restrict__ n_points_dev,
                                                                                   float* __restrict__ pool, CoopGlobal* G, const unsigned* __restrict__ gate /* single-workgroup form launched BEHIND the cooperative one: runs only if that one gave up (*gate == 1) */) {
#pragma clang fp contract(off)
    __shared__ StrictParShared S;
    constexpr int NSLOT = COOP ? 2 : SP_SLOTS;  // rows per lane (cooperative form: rows q + 64 wv and q + 64 (wv + 4) of the workgroup's block)
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (!COOP && gate) {
        if (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) return;
        if (t == 0) atomicAdd(const_cast<unsigned*>(gate) + 2, 1u);  // CoopGlobal::fallbacks (err, max_polls, fallbacks are consecutive words)
    }
    bool dead = false;
    const bool writer = t == 0 && (!COOP || blockIdx.x == 0);  // the one thread that writes the camera record (every cooperative workgroup computes the same numbers)
    unsigned sync_target = 0u;
    if (*n_points_dev < 4) {  // geometry.cpp:84-88
        if (writer) { cam->success = 0; maybe_decide(mp, P, cam, cam_idx); }
        return;
    }
    int used;
    if constexpr (COOP) used = G->used;  // k_pose_strict_compact ran before
    else {
        used = strict_compact_pool(rvecs, tvecs, n_poses, mp.rvec_scale, pool, S);
        __threadfence_block();
        __syncthreads();
    }
    if (used == 0) {
        if (writer) { cam->success = 0; maybe_decide(mp, P, cam, cam_idx); }
        return;
    }
    // a workgroup without a block of the pool takes no part in any sum (it would contribute zeros and wait on words nobody waits for in return)
    if (COOP && blockIdx.x != 0 && (int)blockIdx.x >= (used + 511) / 512) return;
    // ---- this lane's rows into registers
    float X[NSLOT][6];
    {
        const int q = (int)(__brev((unsigned)lane) >> 26);
#pragma unroll
        for (int sl = 0; sl < NSLOT; sl++) {
            const int i = COOP ? (int)blockIdx.x * 512 + 64 * (wv + 4 * sl) + q : (2 * wv + (sl >> 3)) * 512 + 64 * (sl & 7) + q;
#pragma unroll
            for (int d = 0; d < 6; d++) X[sl][d] = i < used ? pool[(size_t)i * 6 + d] : 0.f;
        }
    }
    int parity = 0;
    // ---- mean-shift, meanshift.cu:34-150 (host rand() of :76 -> rng3, as everywhere)
    const bool external_init = mp.use_external_init_mean < 0 ? (cam->pose_sample_count != 0) : (mp.use_external_init_mean != 0);
    float io_mean[6], c_mean[6];
    for (int d = 0; d < 3; d++) { io_mean[d] = cam->rvec[d] * mp.rvec_scale; io_mean[3 + d] = cam->t[d]; }
    const float two_var = 2 * mp.kernel_var;
    if (external_init) {
        for (int d = 0; d < 6; d++) c_mean[d] = io_mean[d];
    } else {
        // :75-95.  The density of a trial does not depend on the other trials, only the decision WHICH trials run does (better-than /
        // good-enough, in order): up to 28 trials are summed in ONE tree (28 values side by side: one barrier instead of 28) and the
        // reference's sequential rule is replayed on the numbers.  Same sums, same picks.
        float best = 0;
        int best_idx = -1;
        bool done = false;
        for (int t0 = 0; t0 < mp.ms_max_init_trials && !done; t0 += 28) {
            const int nt = min(28, mp.ms_max_init_trials - t0);
            __syncthreads();
            if (t < 28 * 6) {
                const int k = t / 6, d = t % 6;
                if (k < nt) S.cent[k][d] = pool[(size_t)(rng3(RAND_SEED, (uint32_t)(t0 + k), 0x4D53u) % (uint32_t)used) * 6 + d];
            }
            __syncthreads();
            auto row = [&](int sl, float (&v)[28]) {
#pragma unroll 1
                for (int k = 0; k < 28; k++) {
                    float l2 = 0;
#pragma unroll
                    for (int d = 0; d < 6; d++) { const float df = X[sl][d] - S.cent[k < nt ? k : 0][d]; l2 += df * df; }
                    v[k] = k < nt ? vsm_expf(-l2 / two_var) : 0.f;
                }
            };
            float o28[28];
            tree_sum_par<28, COOP>(used, row, S, G, sync_target, parity, o28, nullptr, &dead);
            if (COOP && dead) return;
            for (int k = 0; k < nt; k++) {
                const float wsum = o28[k];
                if (wsum > best) { best = wsum; best_idx = (int)(rng3(RAND_SEED, (uint32_t)(t0 + k), 0x4D53u) % (uint32_t)used); }
                if (best > mp.ms_good_init_confidence * (float)used) { done = true; break; }
            }
        }
        if (best_idx < 0) best_idx = 0;
        for (int d = 0; d < 6; d++) c_mean[d] = pool[(size_t)best_idx * 6 + d];
    }
    int ms_iters = 0;
    float conf = 0.f;
    for (int iter = 0; iter < mp.ms_max_iters; iter++) {  // :103-134
        auto row = [&](int sl, float (&v)[7]) {
            float l2 = 0;
#pragma unroll
            for (int d = 0; d < 6; d++) { const float df = X[sl][d] - c_mean[d]; l2 += df * df; }
            const float wgt = vsm_expf(-l2 / two_var);
            v[0] = wgt;
#pragma unroll
            for (int d = 0; d < 6; d++) v[1 + d] = X[sl][d] * wgt;
        };
        float o7[7];
        tree_sum_par<7, COOP>(used, row, S, G, sync_target, parity, o7, nullptr, &dead);
        if (COOP && dead) return;
        const float wsum = o7[0];
        float m[6];
        for (int d = 0; d < 6; d++) m[d] = o7[1 + d] / wsum;
        conf = wsum / (float)used;
        ms_iters = iter + 1;
        float disp = 0;
        for (int d = 0; d < 6; d++) disp += (io_mean[d] - m[d]) * (io_mean[d] - m[d]);  // vs. the stale io mean on the 1st pass (SURVEY B-6)
        disp = sqrtf(disp);
        for (int d = 0; d < 6; d++) io_mean[d] = m[d];
        if (disp < mp.ms_epsilon) break;
        for (int d = 0; d < 6; d++) c_mean[d] = io_mean[d];
    }
    float pose_opm[6];
    for (int d = 0; d < 6; d++) pose_opm[d] = io_mean[d];
    float density = conf;
    int gu_iters = cam->last_used_gu_iters;
    float covar_out[36];
    bool write_covar = false;
    if (mp.do_rg) {  // geometry.cpp:201-246, fit_robust_gaussian.cu:101-286
        const float sc = mp.rg_pose_scaling;
        const int N = used;
        if (t < 21) S.cov[t] = 0.f;
        __syncthreads();
        if (t < 6) S.cov[(t * t + t) / 2 + t] = mp.kernel_var * (sc * sc);  // diag(var) *= sc*sc (:203-208)
        if (t < 6) S.mean[t] = pose_opm[t] * sc;
        __syncthreads();
        float weight = 0;
        int iter = 0;
        bool reliable = true;
        gu_iters = 0;  // fit_robust_gaussian.cu:158-159 resets *used_iters on entry
        for (iter = 0; iter < mp.rg_max_iters; iter++) {
            if (wv == 0) {  // covar_half_to_full (:17-25), Ledoit-Wolf shrinkage (aux_funs.cpp:124-141), inverse (:101-118): rows in lanes
                const bool ok = rg_prepare_wave_rc(S.cov, S.cinv, 6, iter > 0 && mp.rg_covar_reg_lambda > 0, mp.rg_covar_reg_lambda);
                if (lane == 0) S.flag = ok ? 0 : 2;
            }
            __syncthreads();
            if (S.flag == 2) { reliable = false; break; }
            const float prev_density = weight / (float)N;
            float mean[6], cinv[21];
            for (int d = 0; d < 6; d++) mean[d] = S.mean[d];
            for (int k = 0; k < 21; k++) cinv[k] = S.cinv[k];
            auto row = [&](int sl, float (&v)[28]) {  // e_step (:56-97)
                float x[6], diff[6];
#pragma unroll
                for (int d = 0; d < 6; d++) { x[d] = X[sl][d] * sc; diff[d] = x[d] - mean[d]; }
                float z = 0;
#pragma unroll
                for (int d1 = 0; d1 < 6; d1++) {
                    float tmp = 0;
#pragma unroll
                    for (int d2 = 0; d2 < 6; d2++) {
                        const int hi = d1 >= d2 ? d1 : d2, lo = d1 >= d2 ? d2 : d1;
                        tmp += cinv[(hi * hi + hi) / 2 + lo] * diff[d2];
                    }
                    z += tmp * diff[d1];
                }
                z = sqrtf(z);
                const float wgt = z < mp.rg_trunc_sigma ? 1.f : 0.f;
                v[0] = wgt;
#pragma unroll
                for (int d = 0; d < 6; d++) v[1 + d] = wgt * x[d];
#pragma unroll
                for (int d1 = 0; d1 < 6; d1++)
#pragma unroll
                    for (int d2 = 0; d2 <= d1; d2++) v[7 + (d1 * d1 + d1) / 2 + d2] = wgt * diff[d1] * diff[d2];
            };
            float o1[1], mine;
            // only the weight is needed by every thread; thread k (< 28) of wave 0 finishes value k itself
            {
                float o28[28];
                tree_sum_par<28, COOP>(N, row, S, G, sync_target, parity, o28, &mine, &dead);
                if (COOP && dead) return;
                o1[0] = o28[0];
            }
            weight = o1[0];
            if (!isfinite(weight)) { reliable = false; break; }
            if (fabsf(weight / (float)N - prev_density) < mp.rg_epsilon) { reliable = true; break; }
            __syncthreads();  // every thread has read S.mean / S.cinv of this iteration
            if (t >= 1 && t < 7) S.mean[t - 1] = mine / weight;  // m step (:213-243)
            else if (t >= 7 && t < 28) S.cov[t - 7] = mine / weight;
            __syncthreads();
        }
        __syncthreads();
        if (reliable) {
            density = weight / (float)N; gu_iters = iter;
            const float isc2 = cv_div_scale(sc * sc);  // pose_covar /= sc*sc (cv::Mat, :224)
            for (int i1 = 0; i1 < 6; i1++)
                for (int i2 = 0; i2 < 6; i2++) {
                    const int hi = i1 >= i2 ? i1 : i2, lo = i1 >= i2 ? i2 : i1;
                    float cv = S.cov[(hi * hi + hi) / 2 + lo] * isc2;
                    if (i1 < 3 || i2 < 3) cv /= mp.rvec_scale;  // element-wise at<float>() /= (:226-233)
                    if (i1 < 3 && i2 < 3) cv /= mp.rvec_scale;
                    covar_out[i1 * 6 + i2] = cv;
                }
            for (int d = 0; d < 6; d++) pose_opm[d] = S.mean[d];
        } else {
            for (int k = 0; k < 36; k++) covar_out[k] = 0.f;
            for (int d = 0; d < 6; d++) pose_opm[d] = pose_opm[d] * sc;  // pose_opm *= sc (:210) stays as it went in
        }
        write_covar = true;
        const float isc = cv_div_scale(sc);  // pose_opm /= sc (:238)
        for (int d = 0; d < 6; d++) pose_opm[d] *= isc;
    }
    {
        const float irs = cv_div_scale(mp.rvec_scale);  // :249
        for (int d = 0; d < 3; d++) pose_opm[d] *= irs;
    }
    if (writer) {
        bool ok = true;
        for (int d = 0; d < 6; d++) ok = ok && isfinite(pose_opm[d]);  // checkRange :256
        if (COOP) {  // the record is written by the cooperative form XOR by the single-workgroup kernel behind it: claim it (0 -> 2), or leave it alone if somebody gave up (1)
            unsigned expect = 0u;
            if (!__hip_atomic_compare_exchange_strong(&G->err, &expect, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        }
        cam->pose_sample_count = used;
        cam->pose_density = density;
        cam->last_used_ms_iters = ms_iters;
        cam->last_used_gu_iters = gu_iters;
        if (write_covar) for (int k = 0; k < 36; k++) cam->covar[k] = covar_out[k];
        cam->success = ok ? 1 : 0;
        if (ok) {
            for (int d = 0; d < 3; d++) { cam->t[d] = pose_opm[3 + d]; P->ts[cam_idx][d] = pose_opm[3 + d]; }
            float R[9];
            angle_axis_to_rotmat(pose_opm, R, true);
            for (int k = 0; k < 9; k++) P->Rs[cam_idx][k] = R[k];
            // the reference keeps the float matrix only; the vector the next mean shift starts from and the window returns is
            // Camera::rvec() = cv::Rodrigues(R) (utils.h:44-53, geometry.cpp:184): vk_ref_cv.h
            vrcv_rvec_of_R32(R, cam->rvec, 1);
        }
        maybe_decide(mp, P, cam, cam_idx);
    }
}

// ---- B-inner, strict: meanshift_gpu / fit_robust_gaussian on a host-supplied sample matrix space[N][dims] ------------------------
// Same arithmetic as k_pose_strict (the reference's tree-order sums, serial fp64 LU), generic in the dimension like the reference's
// entry points (meanshift.cu:34-150: dims <= 16; fit_robust_gaussian.cu:101-286: dims <= 6).  io layout as k_meanshift_only /
// k_robust_gaussian_only in vk_pose.hip.
__global__ __launch_bounds__(ST_THREADS) static void k_meanshift_strict(const float* __restrict__ space, int N, ModeParams mp, float* __restrict__ io,
                                                                         int* __restrict__ ioi) {
#pragma clang fp contract(off)
    __shared__ TreeBuf<17> tb;
    const int dims = mp.dims;
    float io_mean[16], c_mean[16];
    for (int d = 0; d < 16; d++) { io_mean[d] = d < dims ? io[d] : 0.f; c_mean[d] = io_mean[d]; }
    const float two_var = 2 * mp.kernel_var;
    if (!mp.use_external_init_mean) {  // meanshift.cu:72-95
        float best = 0;
        int best_idx = -1;
        for (int trial = 0; trial < mp.ms_max_init_trials; trial++) {
            const int idx_rand = (int)(rng3(RAND_SEED, (uint32_t)trial, 0x4D53u) % (uint32_t)N);
            auto elem = [&](int i, float* v) {
                float l2 = 0;
                for (int d = 0; d < dims; d++) { const float df = space[(size_t)i * dims + d] - space[(size_t)idx_rand * dims + d]; l2 += df * df; }
                v[0] = vsm_expf(-l2 / two_var);
                for (int k = 1; k < 17; k++) v[k] = 0.f;
            };
            tree_sum<17>(N, elem, tb);
            const float wsum = tb.out[0];
            __syncthreads();
            if (wsum > best) { best = wsum; best_idx = idx_rand; }
            if (best > mp.ms_good_init_confidence * (float)N) break;
        }
        if (best_idx < 0) best_idx = 0;
        for (int d = 0; d < dims; d++) c_mean[d] = space[(size_t)best_idx * dims + d];
    }
    int iters = 0;
    float conf = 0.f;
    for (int iter = 0; iter < mp.ms_max_iters; iter++) {  // :103-134
        auto elem = [&](int i, float* v) {
            float l2 = 0;
            for (int d = 0; d < dims; d++) { const float df = space[(size_t)i * dims + d] - c_mean[d]; l2 += df * df; }
            const float wgt = vsm_expf(-l2 / two_var);
            v[0] = wgt;
            for (int d = 0; d < 16; d++) v[1 + d] = d < dims ? space[(size_t)i * dims + d] * wgt : 0.f;
        };
        tree_sum<17>(N, elem, tb);
        const float wsum = tb.out[0];
        float m[16];
        for (int d = 0; d < 16; d++) m[d] = d < dims ? tb.out[1 + d] / wsum : 0.f;
        __syncthreads();
        conf = wsum / (float)N;
        iters = iter + 1;
        float disp = 0;
        for (int d = 0; d < dims; d++) disp += (io_mean[d] - m[d]) * (io_mean[d] - m[d]);
        disp = sqrtf(disp);
        for (int d = 0; d < dims; d++) io_mean[d] = m[d];
        if (disp < mp.ms_epsilon) break;
        for (int d = 0; d < dims; d++) c_mean[d] = io_mean[d];
    }
    if (threadIdx.x == 0) {
        for (int d = 0; d < dims; d++) io[d] = io_mean[d];
        io[16] = conf; ioi[0] = iters; ioi[1] = 0;
    }
}
// io: [0..5] mean in/out, [6..41] covar full (dims x dims) in/out, [42] density out ; ioi: [0] iters, [1] 0 reliable / 1 not
__global__ __launch_bounds__(ST_THREADS) static void k_rg_strict(const float* __restrict__ space, int N, ModeParams mp, float* __restrict__ io,
                                                                  int* __restrict__ ioi) {
#pragma clang fp contract(off)
    __shared__ TreeBuf<28> tb;
    __shared__ double s_full[36], s_inv[36];
    __shared__ float s_cov[21], s_cinv[21], s_mean[6];
    __shared__ int s_flag;
    const int dims = mp.dims, t = threadIdx.x, dc = (dims * dims + dims) / 2;
    if (t < dims) s_mean[t] = io[t];
    if (t == 0)
        for (int d1 = 0; d1 < dims; d1++)
            for (int d2 = 0; d2 <= d1; d2++) s_cov[(d1 * d1 + d1) / 2 + d2] = io[6 + d1 * dims + d2];
    __syncthreads();
    float weight = 0;
    int iter = 0;
    bool reliable = true;
    for (iter = 0; iter < mp.rg_max_iters; iter++) {
        if (t == 0) {
            for (int d1 = 0; d1 < dims; d1++)
                for (int d2 = 0; d2 <= d1; d2++) {
                    s_full[d1 * dims + d2] = (double)s_cov[(d1 * d1 + d1) / 2 + d2];
                    if (d1 != d2) s_full[d2 * dims + d1] = s_full[d1 * dims + d2];
                }
            if (iter > 0 && mp.rg_covar_reg_lambda > 0) {
                double tr = 0;
                for (int d = 0; d < dims; d++) tr += s_full[d * dims + d];
                const double m = tr / (double)dims, lam = (double)mp.rg_covar_reg_lambda;
                for (int i = 0; i < dims; i++)
                    for (int j = 0; j < dims; j++) s_full[i * dims + j] = lam * m * (i == j ? 1.0 : 0.0) + (1 - lam) * s_full[i * dims + j];
            }
            const double det = lu_inverse_serial(s_full, s_inv, dims);
            s_flag = det <= 0 ? 2 : 0;
            if (det > 0)
                for (int d1 = 0; d1 < dims; d1++)
                    for (int d2 = 0; d2 <= d1; d2++) {
                        s_cov[(d1 * d1 + d1) / 2 + d2] = (float)s_full[d1 * dims + d2];
                        s_cinv[(d1 * d1 + d1) / 2 + d2] = (float)s_inv[d1 * dims + d2];
                    }
        }
        __syncthreads();
        if (s_flag == 2) { reliable = false; break; }
        const float prev_density = weight / (float)N;
        auto elem = [&](int i, float* v) {  // e_step (:56-97)
            float x[6], diff[6];
            for (int d = 0; d < 6; d++) { x[d] = d < dims ? space[(size_t)i * dims + d] : 0.f; diff[d] = d < dims ? x[d] - s_mean[d] : 0.f; }
            float z = 0;
            for (int d1 = 0; d1 < dims; d1++) {
                float tmp = 0;
                for (int d2 = 0; d2 < dims; d2++) {
                    const int hi = d1 >= d2 ? d1 : d2, lo = d1 >= d2 ? d2 : d1;
                    tmp += s_cinv[(hi * hi + hi) / 2 + lo] * diff[d2];
                }
                z += tmp * diff[d1];
            }
            z = sqrtf(z);
            const float wgt = z < mp.rg_trunc_sigma ? 1.f : 0.f;
            for (int k = 0; k < 28; k++) v[k] = 0.f;
            v[0] = wgt;
            for (int d = 0; d < dims; d++) v[1 + d] = wgt * x[d];
            for (int d1 = 0; d1 < dims; d1++)
                for (int d2 = 0; d2 <= d1; d2++) v[7 + (d1 * d1 + d1) / 2 + d2] = wgt * diff[d1] * diff[d2];
        };
        tree_sum<28>(N, elem, tb);
        weight = tb.out[0];
        float nm[6], nc[21];
        for (int d = 0; d < 6; d++) nm[d] = tb.out[1 + d] / weight;
        for (int k = 0; k < 21; k++) nc[k] = tb.out[7 + k] / weight;
        __syncthreads();
        if (!isfinite(weight)) { reliable = false; break; }
        if (fabsf(weight / (float)N - prev_density) < mp.rg_epsilon) { reliable = true; break; }
        if (t < dims) s_mean[t] = nm[t];
        if (t < dc) s_cov[t] = nc[t];
        __syncthreads();
    }
    __syncthreads();
    if (t == 0) {
        ioi[1] = reliable ? 0 : 1;
        if (reliable) {
            ioi[0] = iter;
            io[42] = weight / (float)N;
            for (int d = 0; d < dims; d++) io[d] = s_mean[d];
            for (int d1 = 0; d1 < dims; d1++)
                for (int d2 = 0; d2 <= d1; d2++) {
                    io[6 + d1 * dims + d2] = s_cov[(d1 * d1 + d1) / 2 + d2];
                    io[6 + d2 * dims + d1] = s_cov[(d1 * d1 + d1) / 2 + d2];
                }
        }
    }
}
int meanshift_strict_device(Context* c, const float* space_dev, int N, const ModeParams& mp, float* io_dev, int* ioi_dev) {
    if (N > 2 * ST_THREADS * ST_MAXBLK) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_meanshift_strict, dim3(1), dim3(ST_THREADS), 0, c->stream, space_dev, N, mp, io_dev, ioi_dev);
    VK_CHECK_LAST();
    return 0;
}
int robust_gaussian_strict_device(Context* c, const float* space_dev, int N, const ModeParams& mp, float* io_dev, int* ioi_dev) {
    if (N > 2 * ST_THREADS * ST_MAXBLK) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_rg_strict, dim3(1), dim3(ST_THREADS), 0, c->stream, space_dev, N, mp, io_dev, ioi_dev);
    VK_CHECK_LAST();
    return 0;
}

// The cooperative form spins on its peers: it is only taken when the runtime confirms that the whole grid (at most 16 single-wave workgroups) can be
// resident at once on this device, i.e. the spin can only ever wait for workgroups that are running or about to be dispatched (asked once per process).
static bool coop_fits() {
    static const int fits = [] {
        int per_cu = 0, dev = 0, cus = 0;
        // (a query that fails must not leave its error behind: the launcher's VK_CHECK_LAST would report it as a failed launch -- seen with two HIP runtimes in one process)
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pose_strict_par<true>, COOP_THREADS, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
        return (long long)per_cu * cus >= SP_MAX_POSES / 512 ? 1 : 0;
    }();
    return fits != 0;
}
// cameras of this context whose cooperative mode kernel gave up a meeting and were computed by the single-workgroup kernel instead (read and cleared)
int strict_coop_fallbacks(Context* c) {
    if (!c->sp_coop.p) return 0;
    unsigned v = 0;
    unsigned* d = &c->sp_coop.as<CoopGlobal>()->fallbacks;
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(&v, d, sizeof v, hipMemcpyDeviceToHost) != hipSuccess || hipMemset(d, 0, sizeof v) != hipSuccess) return -1;
    return (int)v;
}
int pose_mode_strict_device(Context* c, int n_poses, const ModeParams& mp, CamState* cam_dev, PoseBlock* P, int cam_idx) {
    if (n_poses > 2 * ST_THREADS * ST_MAXBLK) {
        fprintf(stderr, "voldor_hip: strict mode supports up to %d pose hypotheses\n", 2 * ST_THREADS * ST_MAXBLK);
        return (int)hipErrorInvalidValue;
    }
    if (int e = c->pool.reserve(sizeof(float) * 6 * (size_t)n_poses)) return e;
    if (!debug_switches().strict_plain && n_poses <= SP_MAX_POSES) {  // the parallel tree (same bits)
        if (debug_switches().strict_pose_coop && coop_fits()) {  // one single-wave workgroup per 512-row block, block sums through global memory
            if (!c->sp_coop.p) {
                if (int e = c->sp_coop.reserve(sizeof(CoopGlobal))) return e;
                VK_CHECK(hipMemsetAsync(c->sp_coop.p, 0, sizeof(CoopGlobal), c->stream));
            }
            CoopGlobal* G = c->sp_coop.as<CoopGlobal>();
            static_assert(offsetof(CoopGlobal, fallbacks) == offsetof(CoopGlobal, err) + 2 * sizeof(unsigned), "k_pose_strict_par<false> counts through the gate pointer");
            const int nblk = (n_poses + 511) / 512;
            const int cap = debug_switches().strict_coop_max_polls;
            hipLaunchKernelGGL(k_pose_strict_compact, dim3(1), dim3(SP_THREADS), 0, c->stream, c->rvecs.as<float>(), c->tvecs.as<float>(), n_poses, mp.rvec_scale,
                               c->n_points.as<int>(), c->pool.as<float>(), G, cap > 0 ? (unsigned)cap : (1u << 20));
            hipLaunchKernelGGL(k_pose_strict_par<true>, dim3(nblk), dim3(COOP_THREADS), 0, c->stream, c->rvecs.as<float>(), c->tvecs.as<float>(), n_poses, mp, cam_dev, P, cam_idx,
                               c->n_points.as<int>(), c->pool.as<float>(), G, (const unsigned*)nullptr);
            // behind it, gated on the give-up flag: the same camera on ONE workgroup (same bits); returns at once when the meetings all took place
            hipLaunchKernelGGL(k_pose_strict_par<false>, dim3(1), dim3(SP_THREADS), 0, c->stream, c->rvecs.as<float>(), c->tvecs.as<float>(), n_poses, mp, cam_dev, P, cam_idx,
                               c->n_points.as<int>(), c->pool.as<float>(), (CoopGlobal*)nullptr, (const unsigned*)&G->err);
        } else
            hipLaunchKernelGGL(k_pose_strict_par<false>, dim3(1), dim3(SP_THREADS), 0, c->stream, c->rvecs.as<float>(), c->tvecs.as<float>(), n_poses, mp, cam_dev, P, cam_idx,
                               c->n_points.as<int>(), c->pool.as<float>(), (CoopGlobal*)nullptr, (const unsigned*)nullptr);
        VK_CHECK_LAST();
        return 0;
    }
    hipLaunchKernelGGL(k_pose_strict, dim3(1), dim3(ST_THREADS), 0, c->stream, c->rvecs.as<float>(), c->tvecs.as<float>(), n_poses, mp, cam_dev, P, cam_idx,
                       c->n_points.as<int>(), c->pool.as<float>());
    VK_CHECK_LAST();
    return 0;
}

}  // namespace vk
