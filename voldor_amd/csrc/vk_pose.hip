// voldor_amd/csrc/vk_pose.hip -- pose half of the EM loop on gfx950, device resident.
// Replaces gpu-kernels/collect_p3p_instances.cu:57-145, the host compaction loop of
// voldor/geometry.cpp:68-88, solve_batch_{lambdatwist,ap3p}.cu, meanshift.cu:34-150 (+ its
// host-driven reductions, reduce_vector_sum.h) and fit_robust_gaussian.cu:101-286 (+ the host
// 6x6 algebra of aux_funs.cpp:101-141).
//
// Reference flow per camera per EM iteration: D2H of 6 MB of NaN-sparse maps, a 307k-iteration
// host scan, H2D of the compacted list, 5 cudaMalloc/cudaFree, ~3 launches + 2 blocking D2H per
// mean-shift iteration.  Here: collect (NaN-marked correspondence maps, no compaction in the pipeline) -> four
// lanes per pose hypothesis draw 4 valid pixels by rejection and solve P3P, one candidate root per lane -> ONE
// single-workgroup kernel that runs the whole mean-shift (and, on the last EM iteration, a second one for the robust
// Gaussian) iteration loop with DPP/LDS reductions and writes the new pose straight into the device PoseBlock.
// Only a CamState record per camera is ever read back, once per EM iteration.
#include <atomic>
#include "vk_common.hpp"
#include "vk_device.hpp"
#include "vk_p3p.hpp"
#include "vk_ref_svd.h"
#include "vk_lu.hpp"
#include "vk_ref_cuda.h"
#include "vk_internal.hpp"
#include "vk_cum_poses.hpp"
#include "vk_fb.hpp"

namespace vk {


// phase clocks of the pose kernels (profiling builds only: scripts/phase_clocks.sh compiles a second library with -DVK_PHASE_CLOCKS)
#ifdef VK_PHASE_CLOCKS
__device__ unsigned long long g_phase[64];
#define PH_DECL unsigned long long ph_t = __builtin_amdgcn_s_memtime()
#define PH_MARK(slot) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_phase[slot], n_ - ph_t); ph_t = n_; } } while (0)
#define PH_ADD(slot, v) do { if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&g_phase[slot], (unsigned long long)(v)); } while (0)
#else
#define PH_DECL do {} while (0)
#define PH_MARK(slot) do {} while (0)
#define PH_ADD(slot, v) do {} while (0)
#endif

// ---- collect_p3p_instances.cu:70-145, 1-D pixel indexing so that block order == row-major order
// BLOCK_COMPACT (the window pipeline with the reference's index draw): instead of NaN-marked maps the workgroup writes its VALID
// correspondences only, in pixel order, to the head of its own 256-entry segment of the same buffers -- entry r of block b sits at
// b * 256 + r.  The rank-select draw of k_solve then reads the point it bisected for directly (block, rank in block): no validity
// words, no select inside a word, one dependent memory round trip less per hypothesis, and the collect writes valid entries only.
template <bool BLOCK_COMPACT>
__global__ __launch_bounds__(256) static void k_collect(const float2* __restrict__ flows, const float* __restrict__ rig,
                                                         const float* __restrict__ depth, const PoseBlock* __restrict__ P,
                                                         float* __restrict__ p2_map, float* __restrict__ p3_map,
                                                         int* __restrict__ blk_counts, unsigned long long* __restrict__ valid_mask, int N, int w, int h, int active_idx,
                                                         float rig_thresh, float rig_sum_thresh, float min_depth,
                                                         float max_depth, int max_trace, int ref_tex /* --reference_tex 1: CUDA's linear filter over the N stacked flow layers (vk_ref_cuda.h) */,
                                                         ReduceArgs ra /* partial != NULL: the workgroups after the first n_px_blocks close the previous E-step (reduce_density_block) */) {
    PH_DECL;
    const int npx = w * h;
    const int n_px_blocks = ra.partial ? ra.n_px_blocks : (int)gridDim.x;
    if ((int)blockIdx.x >= n_px_blocks) { reduce_density_block((int)blockIdx.x - n_px_blocks, ra); return; }
    const int tile = xcd_band_tile(blockIdx.x, n_px_blocks);  // XCD k works on the k-th band of rows (vk_device.hpp)
    const int pi = tile * 256 + threadIdx.x;
    const float qnan = __builtin_nanf("");
    bool valid = false;
    float px = qnan, py = qnan;
    P3 o = { qnan, qnan, qnan };
    if (pi < npx) {
        const int x = pi % w, y = pi / w;
        float d = depth[pi];
        const float rig_a = rig[(size_t)active_idx * npx + pi];  // the first factor of the trace product, in flight with the depth
        bool ok = !(d < min_depth || (max_depth > 0.f && d > max_depth));
        if (ok && rig_sum_thresh > (float)(N + 1)) {  // inert unless thr > N+1 (sic, :88-90)
            float rs = 0.f;
            for (int i = 0; i < N; i++) rs += rig[(size_t)i * npx + pi];
            if (rs < rig_sum_thresh) ok = false;
        }
        int n_trace = 0;
        if (ok) {
            float prod = 1.f;
            const int lo = max_trace > 0 ? max(0, active_idx - max_trace + 1) : 0;
            for (int i = active_idx; i >= lo; i--) {
                prod *= i == active_idx ? rig_a : rig[(size_t)i * npx + pi];
                if (prod > rig_thresh) n_trace++;
                else break;
            }
            ok = n_trace > 0;
        }
        if (ok) {
            o = backproject(P, (float)x, (float)y, d);
            bool out = false;
            for (int i = 0; i <= active_idx; i++) {
                if (i >= active_idx - n_trace + 1) {
                    if (i == active_idx - n_trace + 1) project(P, o, px, py);
                    if (px > 0.f && px < (float)w && py > 0.f && py < (float)h) {  // strict (:120)
                        float2 f2;
                        if (ref_tex) vrc_tex_fetch2(reinterpret_cast<const float*>(flows), px, py, i, w, h, N, &f2.x, &f2.y);
                        else f2 = bilinear2(flows + (size_t)i * npx, w, h, px, py);
                        px += f2.x; py += f2.y;
                    } else { out = true; break; }
                }
                if (i < active_idx) o = transform(P->Rs[i], P->ts[i], o);
            }
            valid = !out && o.z > min_depth && (max_depth <= 0.f || o.z < max_depth);
            // geometry.cpp:73 keeps only entries whose sum is finite
            valid = valid && isfinite(px + py + o.x + o.y + o.z);
        }
        if (!BLOCK_COMPACT) {
            p2_map[(size_t)pi * 2] = valid ? px : qnan; p2_map[(size_t)pi * 2 + 1] = valid ? py : qnan;
            p3_map[(size_t)pi * 3] = valid ? o.x : qnan; p3_map[(size_t)pi * 3 + 1] = valid ? o.y : qnan;
            p3_map[(size_t)pi * 3 + 2] = valid ? o.z : qnan;
        }
    }
    __shared__ int s_cnt[4];
    unsigned long long m = __ballot(valid);
    if ((threadIdx.x & 63) == 0) {
        s_cnt[threadIdx.x >> 6] = __popcll(m);
        if (!BLOCK_COMPACT) valid_mask[(size_t)tile * 4 + (threadIdx.x >> 6)] = m;  // one bit per pixel, row-major: what the rank-select draw of k_solve scans over NaN-marked maps
    }
    __syncthreads();
    if (threadIdx.x == 0) blk_counts[tile] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (BLOCK_COMPACT && valid) {
        int r = __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
        for (int k = 0; k < (int)(threadIdx.x >> 6); k++) r += s_cnt[k];
        const size_t slot = (size_t)tile * 256 + r;
        p2_map[slot * 2] = px; p2_map[slot * 2 + 1] = py;
        p3_map[slot * 3] = o.x; p3_map[slot * 3 + 1] = o.y; p3_map[slot * 3 + 2] = o.z;
    }
    PH_MARK(0); PH_ADD(1, 1);
}

// exclusive scan of the per-block counts (single workgroup), total -> *n_points
__global__ __launch_bounds__(1024) static void k_scan_counts(const int* __restrict__ counts, int* __restrict__ offsets, int n,
                                                              int* __restrict__ n_points, CamState* cam) {
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 1024) {
        int i = base + threadIdx.x;
        int v = i < n ? counts[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        int wave_off = 0;
        for (int k = 0; k < wv; k++) wave_off += s_wave[k];
        int carry = s_carry;
        if (i < n) offsets[i] = carry + wave_off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + wave_off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) { *n_points = s_carry; if (cam) cam->n_points = s_carry; }
}

// ordered compaction of the finite entries (voldor/geometry.cpp:68-80) with wave ballots
__global__ __launch_bounds__(256) static void k_compact(const float* __restrict__ p2_map, const float* __restrict__ p3_map,
                                                         const int* __restrict__ offsets, float* __restrict__ pts2,
                                                         float* __restrict__ pts3, int npx) {
    const int pi = blockIdx.x * 256 + threadIdx.x;
    float a = 0, b = 0, cx = 0, cy = 0, cz = 0;
    bool valid = false;
    if (pi < npx) {
        a = p2_map[(size_t)pi * 2]; b = p2_map[(size_t)pi * 2 + 1];
        cx = p3_map[(size_t)pi * 3]; cy = p3_map[(size_t)pi * 3 + 1]; cz = p3_map[(size_t)pi * 3 + 2];
        valid = isfinite(a + b + cx + cy + cz);
    }
    __shared__ int s_cnt[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long m = __ballot(valid);
    if (lane == 0) s_cnt[wv] = __popcll(m);
    __syncthreads();
    int off = offsets[blockIdx.x];
    for (int k = 0; k < wv; k++) off += s_cnt[k];
    if (valid) {
        int r = off + __popcll(m & ((1ull << lane) - 1ull));
        pts2[(size_t)r * 2] = a; pts2[(size_t)r * 2 + 1] = b;
        pts3[(size_t)r * 3] = cx; pts3[(size_t)r * 3 + 1] = cy; pts3[(size_t)r * 3 + 2] = cz;
    }
}

// ---- batched pose hypotheses: one lane each (solve_batch_lambdatwist.cu:11-42, solve_batch_ap3p.cu:331-378)
// Three ways to draw the 4 correspondences of a hypothesis, all uniform over the valid set:
//  FROM_MAP=false  index into the compacted list, (int)(u*N_pts) like the reference kernel -- used by
//                  the host-pointer API solve_batch_p3p_*_gpu, which receives that list;
//  FROM_MAP=true   rejection sampling over the NaN-marked correspondence maps -- used by the
//                  device-resident window pipeline.  Same distribution, no compaction pass, and a
//                  one-pixel change of the valid set only changes the hypotheses that hit that
//                  pixel instead of re-drawing all of them (with list indices a single insertion
//                  shifts every later index; DESIGN.md "sampling stability").
//  FROM_MAP=true, rank select: the reference's own draw WITHOUT materialising its list.  Entry i of the compacted list
//                  (geometry.cpp:68-80) is the i-th valid pixel in row-major order: the per-workgroup counts of k_collect
//                  (row-major blocks of 256 pixels) are prefix-summed in LDS, the block that holds rank i is found by
//                  bisection and the pixel by a scan of that block.  Taken (a) when the valid density is below
//                  1/DRAW_RANK_INV_DENSITY, where rejection within DRAW_MAX_TRIES probes starts to lose hypotheses
//                  (a hypothesis survives with probability (1-(1-rho)^256)^4: 73 % at rho = 1 %), so that the pool has the
//                  reference's size at ANY density >= 4 points; (b) always in reference-draw mode (--reference_draw 1).
// draw k (0..3) of hypothesis idx: curand_uniform number k of the stream curand_init(RAND_SEED, idx, 0) (solve_batch_lambdatwist.cu:16-19,
// :44-48 -- re-seeded per call, so the state table is the same for every launch) with --reference_rng 1, else the counter generator (D1)
__device__ __forceinline__ float draw_uniform(const vrc_xorwow* __restrict__ xw, int idx, int k) {
    if (xw) {
        vrc_xorwow s = xw[idx];
        uint32_t x = 0u;
        for (int j = 0; j <= k; j++) x = vrc_xorwow_next(&s);
        return vrc_uniform(x);
    }
    return u01(rng3(RAND_SEED, (uint32_t)idx, (uint32_t)k));
}
constexpr int DRAW_MAX_TRIES = 256;
constexpr int DRAW_RANK_INV_DENSITY = 20;  // rank select below 5 % valid pixels (rejection then needs > 90 tries for 1 % of the points)
template <int SOLVER, bool FROM_MAP>  // SOLVER: 0 lambdatwist<float>, 1 ap3p, 2 lambdatwist<double>
__device__ __forceinline__ static void solve_body(const float* __restrict__ pts2, const float* __restrict__ pts3,
                                                      float* __restrict__ rvecs, float* __restrict__ tvecs,
                                                      int* __restrict__ n_pts_dev, const int* __restrict__ blk_counts, int nblk,
                                                      CamState* cam, int npx, float fx, float fy, float cx, float cy, int n_poses,
                                                      int draw /* 0 auto, 1 rank select (reference draw), -1 rejection only */, int strict /* bit 0: strict math, bit 1: reference SVD, bit 2: block-compacted correspondences (with draw 1) */,
                                                      const int* __restrict__ blk_offsets /* exclusive prefix of blk_counts in global memory when it does not fit the LDS, else null */,
                                                      const unsigned long long* __restrict__ valid_mask /* k_collect's bit per pixel (FROM_MAP) */,
                                                      const vrc_xorwow* __restrict__ xw /* --reference_rng 1: states after curand_init(RAND_SEED, idx, 0), else null */) {
    // LambdaTwist: four lanes per hypothesis, one candidate root each (the candidates are independent once the
    // shared cubic / eigen-decomposition is done; a lane per hypothesis walks them one after the other and the wave
    // waits for its slowest lane).  AP3P keeps one lane per hypothesis.
    PH_DECL;
    constexpr int LPH = (SOLVER == 1) ? 1 : 4;
    extern __shared__ int s_pref[];  // FROM_MAP: inclusive prefix of blk_counts (rank select only)
    const int ln = threadIdx.x & 63;
    const int gtid = blockIdx.x * 64 + ln, idx = gtid / LPH, sub = gtid % LPH;
    // FROM_MAP: the first batch of rejection probes does not depend on the number of correspondences, so its reads are in flight
    // together with those of the per-block counts (one memory round trip instead of two at the head of the kernel)
    constexpr int DRAW_BATCH = 4;
    int cand0[4][DRAW_BATCH];
    float probe0[4][DRAW_BATCH];
    if (FROM_MAP && draw <= 0) {  // (draw > 0: the rank select never probes)
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int j = 0; j < DRAW_BATCH; j++) {
                const uint32_t r = rng3(RAND_SEED, (uint32_t)idx, (uint32_t)(k * DRAW_MAX_TRIES + j));
                cand0[k][j] = (int)(((unsigned long long)r * (unsigned long long)npx) >> 32);
                probe0[k][j] = pts2[(size_t)cand0[k][j] * 2];
            }
    }
    int n_pts;
    // Inclusive prefix of k_collect's per-block counts into the LDS (what the rank-select draw bisects) and their total.  Per pass lane l
    // takes 32 CONSECUTIVE blocks (one 128-byte line: eight 16-byte loads in flight), sums them up in registers, the wave scans the 64
    // chunk totals once (6 shuffle steps per 2048 blocks) and every lane stores its running sums.  LDS layout padded by one word per 32
    // (pref_at): lane l's j-th store lands in bank (33 l + j) mod 64 -- no conflicts.  640x480: one pass; 1080p (8100 blocks): four.
    auto pref_at = [](int i) { return i + (i >> 5); };
    auto prefix_to_lds = [&]() -> int {
        int carry = 0;
        for (int i0 = 0; i0 < nblk; i0 += 64 * 32) {
            const int b0 = i0 + ln * 32;
            int v[32];
            if (b0 + 32 <= nblk) {
#pragma unroll
                for (int q = 0; q < 8; q++) { const int4 t = *reinterpret_cast<const int4*>(blk_counts + b0 + 4 * q); v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
            } else {
#pragma unroll
                for (int j = 0; j < 32; j++) v[j] = b0 + j < nblk ? blk_counts[b0 + j] : 0;
            }
#pragma unroll
            for (int j = 1; j < 32; j++) v[j] += v[j - 1];
            int incl = v[31];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (ln >= o) incl += t; }
            const int base = carry + incl - v[31];
#pragma unroll
            for (int j = 0; j < 32; j++) if (b0 + j < nblk) s_pref[pref_at(b0 + j)] = base + v[j];
            carry += __shfl(incl, 63, 64);
        }
        return carry;
    };
    if (FROM_MAP) {
        if (draw > 0 && !blk_offsets) {  // the reference's draw (default): the prefix is needed anyway, its last entry is the count
            n_pts = prefix_to_lds();
            __syncthreads();
        } else {
            // number of valid correspondences = sum of k_collect's per-workgroup counts; every wave adds them up itself
            // (a few coalesced loads) instead of a separate single-workgroup launch between collect and solve
            int part = 0;
            for (int i0 = 0; i0 < nblk; i0 += 64 * 32) {  // 32 independent loads in flight per lane: one round trip up to 2048 blocks (1080p: 8100)
                int v[32];
#pragma unroll
                for (int u = 0; u < 32; u++) { const int i = i0 + u * 64 + ln; v[u] = i < nblk ? blk_counts[i] : 0; }
#pragma unroll
                for (int u = 0; u < 32; u++) part += v[u];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
            n_pts = part;
        }
        if (gtid == 0) { *n_pts_dev = n_pts; if (cam) cam->n_points = n_pts; }
    } else
        n_pts = *n_pts_dev;
    PH_MARK(8);
    const bool rank_draw = FROM_MAP && n_pts >= 4 && (draw > 0 || (draw == 0 && (long long)n_pts * DRAW_RANK_INV_DENSITY < (long long)npx));
    if (rank_draw && !blk_offsets && draw <= 0) {  // D3b's low-density fallback: uniform over the workgroup (n_pts is)
        (void)prefix_to_lds();
        __syncthreads();
    }
    if (idx >= n_poses) return;
    const float qnan = __builtin_nanf("");
    float R[9], t[3];
    bool ok = false;
    float errf = 0.f; double errd = 0.0;
    if (n_pts >= (FROM_MAP ? 4 : 1)) {
        float yu[4], yv[4], xp[4][3];
        bool drawn = true;
        int sel[4];
        if (FROM_MAP && rank_draw) {
            // LambdaTwist (four lanes per hypothesis): lane `sub` finds point `sub`, the group exchanges the four pixels -- one search
            // and one mask read per lane instead of four.  AP3P (one lane per hypothesis): four searches side by side.
            constexpr int NS = LPH == 4 ? 1 : 4;
            int rk[NS], lo[NS], hi[NS], want[NS], found[NS];
#pragma unroll
            for (int k = 0; k < NS; k++) {
                // (int)(curand_uniform * N_pts), clamped (D3): solve_batch_lambdatwist.cu:16-19
                rk[k] = min((int)(draw_uniform(xw, idx, LPH == 4 ? sub : k) * (float)n_pts), n_pts - 1);
                lo[k] = 0; hi[k] = nblk - 1;  // first block whose inclusive prefix exceeds the rank
            }
            if (blk_offsets) {  // images beyond 15360 blocks (3.9 MP): the scanned counts stay in global memory (k_scan_counts)
                for (bool more = true; more;) {
                    more = false;
#pragma unroll
                    for (int k = 0; k < NS; k++)
                        if (lo[k] < hi[k]) { const int mid = (lo[k] + hi[k]) >> 1; if (blk_offsets[mid + 1] > rk[k]) hi[k] = mid; else lo[k] = mid + 1; more = true; }
                }
#pragma unroll
                for (int k = 0; k < NS; k++) want[k] = rk[k] - blk_offsets[lo[k]];
            } else {
                for (bool more = true; more;) {
                    more = false;
#pragma unroll
                    for (int k = 0; k < NS; k++)
                        if (lo[k] < hi[k]) { const int mid = (lo[k] + hi[k]) >> 1; if (s_pref[pref_at(mid)] > rk[k]) hi[k] = mid; else lo[k] = mid + 1; more = true; }
                }
#pragma unroll
                for (int k = 0; k < NS; k++) want[k] = rk[k] - (lo[k] > 0 ? s_pref[pref_at(lo[k] - 1)] : 0);
            }
            if (strict & 4) {  // block-compacted correspondences (k_collect<true>): entry `want` of block lo is the point itself
#pragma unroll
                for (int k = 0; k < NS; k++) found[k] = lo[k] * 256 + want[k];
            } else {
            // the want-th valid pixel of block lo: its four 64-bit validity words, popcounts, then a 6-step select inside the word
            unsigned long long wds[NS][4];
#pragma unroll
            for (int k = 0; k < NS; k++) {
                const ulonglong2* mw = reinterpret_cast<const ulonglong2*>(valid_mask + (size_t)lo[k] * 4);
                const ulonglong2 a = mw[0], b = mw[1];
                wds[k][0] = a.x; wds[k][1] = a.y; wds[k][2] = b.x; wds[k][3] = b.y;
            }
#pragma unroll
            for (int k = 0; k < NS; k++) {
                int wn = want[k];
                const int c0 = __popcll(wds[k][0]), c1 = __popcll(wds[k][1]), c2 = __popcll(wds[k][2]);
                unsigned long long wd = wds[k][0]; int q = 0;
                if (wn >= c0) { wn -= c0; wd = wds[k][1]; q = 1; if (wn >= c1) { wn -= c1; wd = wds[k][2]; q = 2; if (wn >= c2) { wn -= c2; wd = wds[k][3]; q = 3; } } }
                found[k] = -1;
                if (wn < __popcll(wd)) {
                    int pos = 0;
#pragma unroll
                    for (int sft = 32; sft >= 1; sft >>= 1) {
                        const int c = __popcll(wd & (((1ull << sft) - 1ull) << pos));
                        if (wn >= c) { wn -= c; pos += sft; }
                    }
                    found[k] = lo[k] * 256 + q * 64 + pos;
                }
            }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                sel[k] = LPH == 4 ? __shfl(found[0], (((int)threadIdx.x & 63) & ~3) + k, 64) : found[NS == 4 ? k : 0];
                if (sel[k] < 0) { drawn = false; sel[k] = 0; }  // cannot happen: the counts come from the same map
            }
        } else if (FROM_MAP) {
            // Rejection draw over the NaN-marked correspondence map: point k takes the first valid pixel of its
            // own candidate sequence j = 0,1,2,...  A probe is a random read (one L2/HBM round trip), so the
            // probes are issued DRAW_BATCH tries x 4 points at a time instead of one dependent read per try;
            // the selected pixels are the same.
#pragma unroll
            for (int k = 0; k < 4; k++) sel[k] = -1;
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int j = 0; j < DRAW_BATCH; j++)
                    if (sel[k] < 0 && isfinite(probe0[k][j])) sel[k] = cand0[k][j];
            for (int j0 = DRAW_BATCH; j0 < DRAW_MAX_TRIES; j0 += DRAW_BATCH) {
                if (sel[0] >= 0 && sel[1] >= 0 && sel[2] >= 0 && sel[3] >= 0) break;
                int cand[4][DRAW_BATCH];
                float probe[4][DRAW_BATCH];
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int j = 0; j < DRAW_BATCH; j++) {
                        const uint32_t r = rng3(RAND_SEED, (uint32_t)idx, (uint32_t)(k * DRAW_MAX_TRIES + j0 + j));
                        cand[k][j] = (int)(((unsigned long long)r * (unsigned long long)npx) >> 32);
                        probe[k][j] = pts2[(size_t)cand[k][j] * 2];
                    }
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int j = 0; j < DRAW_BATCH; j++)
                        if (sel[k] < 0 && isfinite(probe[k][j])) sel[k] = cand[k][j];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) if (sel[k] < 0) { drawn = false; sel[k] = 0; }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // the reference re-seeds per call, so the pattern depends only on (idx, n_pts)
                // (solve_batch_lambdatwist.cu:16-19,80-81). u in (0,1]: clamp the one-past-the-end
                // index the reference can produce.
                int i = (int)(draw_uniform(xw, idx, k) * (float)n_pts);
                sel[k] = min(i, n_pts - 1);
            }
        }
        PH_MARK(9);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = sel[k];
            yu[k] = pts2[(size_t)i * 2]; yv[k] = pts2[(size_t)i * 2 + 1];
            xp[k][0] = pts3[(size_t)i * 3]; xp[k][1] = pts3[(size_t)i * 3 + 1]; xp[k][2] = pts3[(size_t)i * 3 + 2];
        }
#ifdef VK_PHASE_CLOCKS
        if (yu[0] + xp[3][2] == 123456.f) return;  // forces the loads to complete before the mark
#endif
        PH_MARK(10);
        if (drawn) {
            if (SOLVER == 0) ok = lambdatwist_p4p<float>(yu, yv, xp, fx, fy, cx, cy, R, t, sub, &errf);
            else if (SOLVER == 2) ok = lambdatwist_p4p<double>(yu, yv, xp, fx, fy, cx, cy, R, t, sub, &errd);
            else ok = ap3p_p4p(yu, yv, xp, fx, fy, cx, cy, R, t, (strict & 1) != 0);
        }
    }
    PH_MARK(11);
    float aa[3] = { qnan, qnan, qnan };
    if (ok) {
        // rodrigues.h:82-114.  Default: the exact polar factor (D8); --reference_svd 1: U V^T of the reference's approximate SVD, bit for bit
        if (strict & 2) vrs_project_rotation(R); else nearest_rotation(R);
        rotmat_to_angle_axis(R, aa, (strict & 1) != 0);
    }
    PH_MARK(12);
    bool writer = true;
    if (LPH == 4) {
        // fold the four candidates in root order: the first valid one, then any later one with a strictly smaller
        // 4th-point error (lambdatwist_p4p.h:43-58); every lane of the group evaluates the same fold
        const int base = (threadIdx.x & 63) & ~3;
        int win = -1;
        double werr = 0.0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int okc = __shfl((int)ok, base + c, 64);
            const double ec = (SOLVER == 2) ? __shfl(errd, base + c, 64) : (double)__shfl(errf, base + c, 64);
            if (okc && (win < 0 || werr > ec)) { win = c; werr = ec; }
        }
        writer = (win < 0) ? (sub == 0) : (sub == win);  // no candidate: lane 0 writes the NaN marker
    }
    if (writer) {
        if (FROM_MAP) {  // device pipeline: coordinate planes [3][n_poses], what the single-workgroup mode kernels read coalesced
            rvecs[idx] = aa[0]; rvecs[(size_t)n_poses + idx] = aa[1]; rvecs[(size_t)2 * n_poses + idx] = aa[2];
            tvecs[idx] = ok ? t[0] : qnan; tvecs[(size_t)n_poses + idx] = ok ? t[1] : qnan; tvecs[(size_t)2 * n_poses + idx] = ok ? t[2] : qnan;
        } else {  // host-pointer API: [n_poses][3] like the reference's output arrays
            rvecs[(size_t)idx * 3] = aa[0]; rvecs[(size_t)idx * 3 + 1] = aa[1]; rvecs[(size_t)idx * 3 + 2] = aa[2];
            tvecs[(size_t)idx * 3] = ok ? t[0] : qnan; tvecs[(size_t)idx * 3 + 1] = ok ? t[1] : qnan; tvecs[(size_t)idx * 3 + 2] = ok ? t[2] : qnan;
        }
    }
    PH_MARK(13); PH_ADD(14, 1);
}
template <int SOLVER, bool FROM_MAP>
__global__ __launch_bounds__(64) static void k_solve(const float* __restrict__ pts2, const float* __restrict__ pts3, float* __restrict__ rvecs, float* __restrict__ tvecs,
                                                      int* __restrict__ n_pts_dev, const int* __restrict__ blk_counts, int nblk, CamState* cam, int npx, float fx, float fy,
                                                      float cx, float cy, int n_poses, int draw, int strict, const int* __restrict__ blk_offsets,
                                                      const unsigned long long* __restrict__ valid_mask, const vrc_xorwow* __restrict__ xw) {
    solve_body<SOLVER, FROM_MAP>(pts2, pts3, rvecs, tvecs, n_pts_dev, blk_counts, nblk, cam, npx, fx, fy, cx, cy, n_poses, draw, strict, blk_offsets, valid_mask, xw);
}

// ---- single-workgroup mode finding ---------------------------------------------------------------
constexpr int MS_THREADS = 1024;
struct BlockRed {  // all-reduce of NV floats over the 1024-thread workgroup, fixed order
    float wave[16][32];
    float total[32];
};
template <int NV>
__device__ __forceinline__ void block_allreduce(float* v, BlockRed& br) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        float s = wave_sum(v[k]);
        if (lane == 0) br.wave[wv][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j++) s += br.wave[j][threadIdx.x];
        br.total[threadIdx.x] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = br.total[k];
    __syncthreads();
}


// mean-shift on `space[N][dims]` (generic dims<=16): returns via LDS-broadcast state.
// mean_io: in = current mean (io semantics of h_io_mean), out = mode.
__device__ static void meanshift_block(const float* __restrict__ space, int N, int dims, const ModeParams& mp,
                                       float* mean_io /*LDS [16]*/, float* c_mean /*LDS [16]*/, BlockRed& br,
                                       float* o_conf, int* o_iters) {
    __shared__ int s_stop;
    const int tid = threadIdx.x;
    const float inv2v = 1.f / (2.f * mp.kernel_var);
    if (mp.use_external_init_mean) {
        if (tid < dims) c_mean[tid] = mean_io[tid];
        __syncthreads();
    } else {  // best of <= max_init_trials random samples (meanshift.cu:72-95), host rand() -> rng3
        float best = 0.f; int best_idx = 0; bool have = false;
        for (int trial = 0; trial < mp.ms_max_init_trials; trial++) {
            int idx_rand = (int)(rng3(RAND_SEED, (uint32_t)trial, 0x4D53u) % (uint32_t)N);
            float acc[1] = { 0.f };
            for (int i = tid; i < N; i += MS_THREADS) {
                float l2 = 0.f;
                for (int d = 0; d < dims; d++) { float df = space[(size_t)i * dims + d] - space[(size_t)idx_rand * dims + d]; l2 += df * df; }
                acc[0] += __expf(-l2 * inv2v);
            }
            block_allreduce<1>(acc, br);
            if (acc[0] > best) { best = acc[0]; best_idx = idx_rand; have = true; }
            if (best > mp.ms_good_init_confidence * (float)N) break;
        }
        (void)have;
        if (tid < dims) c_mean[tid] = space[(size_t)best_idx * dims + tid];
        __syncthreads();
    }
    int iters = 0;
    float conf = 0.f;
    for (int iter = 0; iter < mp.ms_max_iters; iter++) {  // meanshift.cu:103-134
        float acc[17];
#pragma unroll
        for (int k = 0; k < 17; k++) acc[k] = 0.f;
        if (dims == 6) {
            float m0 = c_mean[0], m1 = c_mean[1], m2 = c_mean[2], m3 = c_mean[3], m4 = c_mean[4], m5 = c_mean[5];
            for (int i = tid; i < N; i += MS_THREADS) {
                const float* s = space + (size_t)i * 6;
                float x0 = s[0], x1 = s[1], x2 = s[2], x3 = s[3], x4 = s[4], x5 = s[5];
                float l2 = (x0 - m0) * (x0 - m0) + (x1 - m1) * (x1 - m1) + (x2 - m2) * (x2 - m2) + (x3 - m3) * (x3 - m3) +
                           (x4 - m4) * (x4 - m4) + (x5 - m5) * (x5 - m5);
                float wgt = __expf(-l2 * inv2v);
                acc[0] += wgt; acc[1] += wgt * x0; acc[2] += wgt * x1; acc[3] += wgt * x2; acc[4] += wgt * x3;
                acc[5] += wgt * x4; acc[6] += wgt * x5;
            }
            block_allreduce<7>(acc, br);
        } else {
            for (int i = tid; i < N; i += MS_THREADS) {
                float l2 = 0.f;
                for (int d = 0; d < dims; d++) { float df = space[(size_t)i * dims + d] - c_mean[d]; l2 += df * df; }
                float wgt = __expf(-l2 * inv2v);
                acc[0] += wgt;
#pragma unroll
                for (int d = 0; d < 16; d++) if (d < dims) acc[1 + d] += wgt * space[(size_t)i * dims + d];
            }
            block_allreduce<17>(acc, br);
        }
        conf = acc[0] / (float)N;
        iters = iter + 1;
        if (tid == 0) {
            float disp = 0.f;
            for (int d = 0; d < dims; d++) {
                float m = acc[1 + d] / acc[0];
                disp += (mean_io[d] - m) * (mean_io[d] - m);  // vs. the stale io mean on the 1st pass (SURVEY B-6)
                mean_io[d] = m;
                c_mean[d] = m;
            }
            s_stop = sqrtf(disp) < mp.ms_epsilon;
        }
        __syncthreads();
        if (s_stop) break;
    }
    *o_conf = conf;
    *o_iters = iters;
}

// The same prepare step with the matrices in LDS (A, B: 36 doubles each) and one lane per element:
// a handful of registers instead of ~100, for kernels whose register budget is pinned elsewhere.
// Element updates are independent within a step -> still bit-identical to the serial LU.
// `warm`: lds72[36..71] still holds the inverse this function produced for the previous iteration's covariance.  The
// gate tightens by a few percent per iteration, so that inverse X is a good start for the Newton-Schulz iteration
// X <- X + X (I - A X): every lane forms one element of each 6x6 product (6 multiply-adds, reads in one LDS batch), the
// residual shrinks quadratically, ~3 updates reach 1e-8.  ~2k cycles against ~9k for the LU, whose pivot steps and
// divisions are a serial fp64 chain.  The result is the same inverse to ~1e-8 relative (it is rounded to float
// afterwards); if the residual of the start value is not small (first iterations) the LU runs instead.  ||I - A X|| < 1
// with X positive definite implies det A > 0, the condition the LU path checks explicitly.
__device__ __forceinline__ bool rg_prepare_lds(float* covar_half, float* cinv_half, float lambda, bool regularise, double* lds72,
                                               bool warm = false) {
    // One lane per element (r,c); the lane keeps its own A and B element in registers, LDS is the exchange medium.
    // This runs on ONE wave while the other 15 wait, so what counts is the number of dependent LDS round trips:
    // every step issues its reads as one batch (pivot column; then both candidate rows), 3 trips per pivot instead
    // of one per element touched.
    constexpr int n = 6;
    const int lane = threadIdx.x & 63;
    const bool act = lane < n * n;
    const int r = act ? lane / n : 0, c = act ? lane % n : 0;
    double* A = lds72; double* B = lds72 + 36;
    double a, b;
    {
        const int hi = r >= c ? r : c, lo = r >= c ? c : r;
        double full = (double)covar_half[(hi * hi + hi) / 2 + lo];
        if (regularise) {
            double tr = 0;
#pragma unroll
            for (int d = 0; d < n; d++) tr += (double)covar_half[(d * d + d) / 2 + d];
            const double m = tr / (double)n, lam = (double)lambda;
            full = lam * m * (r == c ? 1.0 : 0.0) + (1 - lam) * full;
        }
        a = full; b = (r == c) ? 1.0 : 0.0;
        __builtin_amdgcn_wave_barrier();
        if (act) A[lane] = a;
        if (act && r >= c) covar_half[(r * r + r) / 2 + c] = (float)full;
        __builtin_amdgcn_wave_barrier();
    }
    if (warm) {
        double* E = lds72 + 72;
        double x = act ? B[lane] : 0.0;
        bool done = false, fallback = false;
#pragma unroll 1
        for (int step = 0; step < 8 && !done; step++) {
            double ar[n], xc[n];
#pragma unroll
            for (int k = 0; k < n; k++) { ar[k] = A[r * n + k]; xc[k] = B[k * n + c]; }  // one batch
            double y = 0.0;
#pragma unroll
            for (int k = 0; k < n; k++) y += ar[k] * xc[k];
            const double e = (r == c ? 1.0 : 0.0) - y;
            const double ae = act ? fabs(e) : 0.0;
            if (step == 0 && __ballot(!(ae < 0.3)) != 0ull) { fallback = true; break; }  // also catches NaN
            // the update below squares the residual, E <- E^2 exactly, so max|E| <= 4e-5 now means <= 6 * 1.6e-9 = 1e-8 after it:
            // below the float rounding of the result; no further product is spent on verifying that
            done = __ballot(ae > 4e-5) == 0ull;
            __builtin_amdgcn_wave_barrier();
            if (act) E[lane] = e;
            __builtin_amdgcn_wave_barrier();
            double xr[n], ec[n];
#pragma unroll
            for (int k = 0; k < n; k++) { xr[k] = B[r * n + k]; ec[k] = E[k * n + c]; }
            double u = 0.0;
#pragma unroll
            for (int k = 0; k < n; k++) u += xr[k] * ec[k];
            x += u;
            __builtin_amdgcn_wave_barrier();
            if (act) B[lane] = x;
            __builtin_amdgcn_wave_barrier();
        }
        if (!fallback && done) {
            if (act && r >= c) cinv_half[(r * r + r) / 2 + c] = (float)x;
            return true;
        }
    }
    if (act) B[lane] = b;
    __builtin_amdgcn_wave_barrier();
    double det = 1.0;
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        double col[n];
#pragma unroll
        for (int j = 0; j < n; j++) col[j] = A[j * n + i];  // one batch
        const double mycol = A[r * n + i];
        int k = i;
        double best = -1.0, piv = 0.0;
#pragma unroll
        for (int j = 0; j < n; j++) {
            const double v = fabs(col[j]);
            if (j >= i && (j == i ? true : v > best)) { best = v; k = j; piv = col[j]; }
        }
        if (best < 2.220446049250313e-16) return false;
        // rows i and k swap (k == i: both reads hit row i); everybody needs the new pivot row = old row k
        const double ak = A[k * n + c], bk = B[k * n + c], ai = A[i * n + c], bi = B[i * n + c];
        if (k != i) det = -det;
        det *= piv;
        const double d = -1.0 / piv;
        double old_ii = 0.0;  // A_old[i][i] = column element of the row that moves to position k
#pragma unroll
        for (int j = 0; j < n; j++) if (j == i) old_ii = col[j];
        if (act) {
            if (r == i) { a = ak; b = bk; }
            else if (r > i) {
                double ra = a, rb = b, rc = mycol;
                if (r == k) { ra = ai; rb = bi; rc = old_ii; }  // this lane's row now holds old row i
                const double alpha = rc * d;
                a = (c > i) ? ra + alpha * ak : ra;
                b = rb + alpha * bk;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (act) { A[lane] = a; B[lane] = b; }
        __builtin_amdgcn_wave_barrier();
    }
    if (det <= 0.0) return false;
    // back substitution: row i of the inverse from rows > i; U's row and the reciprocal pivot are fetched once
    double urow[n];
#pragma unroll
    for (int k = 0; k < n; k++) urow[k] = A[r * n + k];
    double rdiag = 1.0;
#pragma unroll
    for (int k = 0; k < n; k++) if (k == r) rdiag = 1.0 / urow[k];
#pragma unroll 1
    for (int i = n - 1; i >= 0; i--) {
        double below[n];
#pragma unroll
        for (int k = 0; k < n; k++) below[k] = B[k * n + c];  // one batch; rows > i are final
        if (act && r == i) {
            double sacc = b;
#pragma unroll
            for (int k = 0; k < n; k++) if (k > i) sacc -= urow[k] * below[k];
            b = sacc * rdiag;
            B[lane] = b;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (act && r >= c) cinv_half[(r * r + r) / 2 + c] = (float)b;
    return true;
}

// robust Gaussian on `space[N][dims]`, dims<=6 (fit_robust_gaussian.cu:131-263).
// mean[6]/covar_half[21] in LDS: in = initial, out = result (only if reliable). returns reliable.
__device__ static bool robust_gaussian_block(const float* __restrict__ space, int N, int dims, float scale, const ModeParams& mp,
                                             float* mean /*LDS[6]*/, float* covar_half /*LDS[21]*/, float* cinv_half /*LDS[21]*/,
                                             BlockRed& br, float* o_density, int* o_iters) {
    __shared__ int s_flag;  // 0 continue, 2 unreliable
    __shared__ double s_lu[108];
    const int tid = threadIdx.x;
    const int dc = (dims * dims + dims) / 2;
    float weight = 0.f;
    int iter = 0;
    bool reliable = true;
    for (iter = 0; iter < mp.rg_max_iters; iter++) {
        if (tid < 64) {
            // 6-D (poses): the LDS variant, the same code refit_block runs; other dimensions: the register variant
            const bool ok = dims == 6 ? rg_prepare_lds(covar_half, cinv_half, mp.rg_covar_reg_lambda, iter > 0 && mp.rg_covar_reg_lambda > 0.f, s_lu, iter > 0)
                                      : rg_prepare_wave(covar_half, cinv_half, dims, iter > 0 && mp.rg_covar_reg_lambda > 0.f, mp.rg_covar_reg_lambda);
            if (tid == 0) s_flag = ok ? 0 : 2;
        }
        __syncthreads();
        if (s_flag == 2) { reliable = false; break; }
        const float prev_density = weight / (float)N;
        float acc[28];
#pragma unroll
        for (int k = 0; k < 28; k++) acc[k] = 0.f;
        for (int i = tid; i < N; i += MS_THREADS) {  // e_step (fit_robust_gaussian.cu:56-97)
            float x[6], diff[6];
#pragma unroll
            for (int d = 0; d < 6; d++) { x[d] = d < dims ? space[(size_t)i * dims + d] * scale : 0.f; diff[d] = d < dims ? x[d] - mean[d] : 0.f; }
            float z = 0.f;
#pragma unroll
            for (int d1 = 0; d1 < 6; d1++) {
                float tmp = 0.f;
#pragma unroll
                for (int d2 = 0; d2 < 6; d2++) {
                    if (d1 < dims && d2 < dims) {
                        int hi = d1 >= d2 ? d1 : d2, lo = d1 >= d2 ? d2 : d1;
                        tmp += cinv_half[(hi * hi + hi) / 2 + lo] * diff[d2];
                    }
                }
                z += tmp * diff[d1];
            }
            z = sqrtf(z);
            const float wgt = z < mp.rg_trunc_sigma ? 1.f : 0.f;
            acc[0] += wgt;
#pragma unroll
            for (int d = 0; d < 6; d++) acc[1 + d] += wgt * x[d];
#pragma unroll
            for (int d1 = 0; d1 < 6; d1++)
#pragma unroll
                for (int d2 = 0; d2 <= d1; d2++) acc[7 + (d1 * d1 + d1) / 2 + d2] += wgt * diff[d1] * diff[d2];
        }
        block_allreduce<28>(acc, br);
        weight = acc[0];
        if (!isfinite(weight)) { reliable = false; break; }
        const float density_change = fabsf(weight / (float)N - prev_density);
        if (density_change < mp.rg_epsilon) { reliable = true; break; }
        if (tid == 0) {
            for (int d = 0; d < dims; d++) mean[d] = acc[1 + d] / weight;
            for (int k = 0; k < dc; k++) covar_half[k] = acc[7 + k] / weight;
        }
        __syncthreads();
    }
    __syncthreads();
    *o_density = weight / (float)N;
    *o_iters = iter;
    return reliable;
}

// ---- the per-camera mode kernel of the device-resident pipeline (voldor/geometry.cpp:156-263)
// One workgroup; every thread keeps its share of the hypotheses (6 floats each) in registers for the whole mean-shift /
// robust-Gaussian iteration, so an iteration is a pass over registers + one shuffle/LDS all-reduce with a single barrier (the
// reference does 3 launches + 2 blocking D2H per iteration, meanshift.cu:103-134).
struct RedBuf { float w[2][16][28]; };
// All-reduce of NV per-thread partial sums over the workgroup, fixed summation order, ONE barrier:
// wave partials -> LDS -> barrier -> in every wave lane k (< NV) sums the NW partials of value k, and the
// totals are broadcast inside the wave with v_readlane.  (Summing all NV*NW partials in every thread
// makes the compiler batch NV*NW LDS loads into registers: 400+ VGPR spills in the refit kernel; a
// second barrier for a shared total costs more than the redundant 16-term sums.)
// allreduce_lanes: in every wave, lane k (< NV) returns the workgroup total of value k (other lanes 0)
template <int NV, int NW = 16>
__device__ __forceinline__ float allreduce_lanes(const float* v, RedBuf& rb, int parity) {
    constexpr int P = NV <= 8 ? 8 : 32;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float pv[P];
#pragma unroll
    for (int k = 0; k < P; k++) pv[k] = k < NV ? v[k] : 0.f;
    const float mine = wave_reduce_transpose<P>(pv);  // wave total of value wave_slot<P>(lane)
    const int slot = wave_slot<P>(lane);
    if (lane < P && slot < NV) rb.w[parity][wv][slot] = mine;
    __syncthreads();
    float tot = 0.f;
    if (lane < NV) {
#pragma unroll
        for (int j = 0; j < NW; j++) tot += rb.w[parity][j][lane];
    }
    return tot;
}
__device__ __forceinline__ float lane_value(float v, int k) {  // wave-uniform k
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}
template <int NV, int NW = 16>
__device__ __forceinline__ void allreduce_regs(float* v, RedBuf& rb, int parity) {
    const float tot = allreduce_lanes<NV, NW>(v, rb, parity);
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = lane_value(tot, k);
}

// geometry.cpp:249-263: unscale, checkRange, write the pose into CamState and the PoseBlock (thread 0)
__device__ __forceinline__ void finalize_pose(const float* mean6 /*scaled space*/, float rvec_scale, int used, float density, int ms_iters,
                                              int gu_iters, CamState* cam, PoseBlock* P, int cam_idx) {
    float pose[6];
    for (int d = 0; d < 3; d++) pose[d] = mean6[d] / rvec_scale;  // :249
    for (int d = 3; d < 6; d++) pose[d] = mean6[d];
    bool ok = true;
    for (int d = 0; d < 6; d++) ok = ok && isfinite(pose[d]);  // checkRange :256
    cam->pose_sample_count = used;
    cam->pose_density = density;
    cam->last_used_ms_iters = ms_iters;
    cam->last_used_gu_iters = gu_iters;
    cam->success = ok ? 1 : 0;
    if (ok) {
        for (int d = 0; d < 3; d++) { cam->rvec[d] = pose[d]; cam->t[d] = pose[3 + d]; P->ts[cam_idx][d] = pose[3 + d]; }
        float R[9];
        angle_axis_to_rotmat(pose, R);
        for (int k = 0; k < 9; k++) P->Rs[cam_idx][k] = R[k];
    }
}

// mean-shift stage. DEFER=false: finalises the pose. DEFER=true (last EM iteration, geometry.cpp:201): the robust-Gaussian refit
// (refit_block) runs on the same registers and finalises.
//
// One compute unit does all of it, and its passes over the pool are VALU-issue bound (16 waves on 4 SIMDs), so the pool is
// held as PAIRS of hypotheses per lane and the arithmetic is written on float2: a v_pk_{add,mul,fma}_f32 takes 1.5x the issue time of
// the scalar form on gfx950 (8.5 against 5.5 clocks per wave: scripts/micro/pk_rate.hip) and does two hypotheses.  Hypothesis i lives in thread i % 1024, pair (i / 1024) / 2,
// half (i / 1024) % 2.  A non-finite hypothesis (geometry.cpp:156-165 drops those) is stored as MS_FAR in every coordinate:
// its squared distance to anything is ~6e36, its kernel weight exp(-6e36 / 2 var) is exactly 0 and 0 * MS_FAR = 0, so it
// contributes exactly nothing to any sum -- no mask in the loops.
typedef float f2 __attribute__((ext_vector_type(2)));
#ifndef VK_PM_THREADS
#define VK_PM_THREADS 512
#endif
constexpr int PM_THREADS = VK_PM_THREADS;
constexpr int PM_POOL = 8192;   // hypotheses the mode kernels keep on chip (cfg.n_poses_to_sample default)
constexpr float MS_FAR = 1e18f;
#ifndef VK_MS_TRIAL_BATCH
#define VK_MS_TRIAL_BATCH 5
#endif
constexpr int MS_TRIAL_BATCH = VK_MS_TRIAL_BATCH;  // initial-mode trials evaluated per pass (meanshift.cu:72-95 runs them one at a time)
// The prepare step of a gate iteration (fit_robust_gaussian.cu:172-205: Ledoit-Wolf shrinkage with fixed lambda, aux_funs.cpp:124-141, then
// the inverse the gate needs) for the refit below, in a form EVERY LANE evaluates for itself from the all-reduced totals: no wave is
// singled out, nothing goes through LDS, no second workgroup barrier.  The gate only needs the quadratic form d^T C^-1 d, and with
// the Cholesky factor C = L L^T that is |y|^2, L y = d: a forward substitution per sample with the strict lower triangle of L and the
// reciprocal of its diagonal -- 21 coefficients like the packed inverse, the same 27 multiply-adds per sample -- and a factorisation
// of ~80 fp64 instructions (6 reciprocal square roots by v_rsq_f64 + one Newton step) instead of the LU / Newton-Schulz inverse
// exchanged through LDS by one wave (~40 % of a gate iteration before).  L is better conditioned than C^-1 (square root of its
// condition number).  A pivot that is not positive: the covariance is not positive definite -- the reference's
// `det <= 0` verdict (unreliable fit).  W (out): L's strict lower triangle, 1 / L_ii on the diagonal.
// cov: packed lower triangle (float).  rg_regularised: the regularised covariance rounded to float (what the reference keeps and reports;
// needed once, when the loop has ended).
__device__ __forceinline__ double rsqrt_f64(double s) {
#pragma clang fp contract(fast)
    double y = __builtin_amdgcn_rsq(s);  // v_rsq_f64: 2^29 ulp = 2^-23 relative; one Newton step squares that (2^-45: the factor ends up in floats)
    return y * (1.5 - (0.5 * s) * y * y);
}
__device__ __forceinline__ void rg_regularised(const float (&cov)[21], bool regularise, float lambda, float (&covr)[21]) {
    double tr = 0.0;
#pragma unroll
    for (int d = 0; d < 6; d++) tr += (double)cov[(d * d + d) / 2 + d];
    const double m = tr / 6.0, lam = (double)lambda;
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c <= r; c++) {
            double full = (double)cov[(r * r + r) / 2 + c];
            if (regularise) full = lam * m * (r == c ? 1.0 : 0.0) + (1 - lam) * full;
            covr[(r * r + r) / 2 + c] = (float)full;
        }
}
__device__ __forceinline__ bool rg_whiten(const float (&cov)[21], bool regularise, float lambda, float (&W)[21]) {
#pragma clang fp contract(fast)  // not part of the solver's exact-rounding contract (file-wide: off)
    double a[21];
    double tr = 0.0;
#pragma unroll
    for (int d = 0; d < 6; d++) tr += (double)cov[(d * d + d) / 2 + d];
    const double lam = regularise ? (double)lambda : 0.0, lm = lam * (tr / 6.0), ol = 1 - lam;  // lambda = 0: full = 0 + 1 * full, exactly
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c <= r; c++) a[(r * r + r) / 2 + c] = (r == c ? lm : 0.0) + ol * (double)cov[(r * r + r) / 2 + c];
    double L[21], inv[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double s = a[(j * j + j) / 2 + j];
#pragma unroll
        for (int k = 0; k < j; k++) s -= L[(j * j + j) / 2 + k] * L[(j * j + j) / 2 + k];
        ok = ok && (s > 0.0);  // every pivot positive <=> positive definite <=> the reference's det > 0 (fit_robust_gaussian.cu:195); false for NaN as well (the reference lets a NaN determinant pass and reports NaNs: here the fit counts as unreliable, which checkRange would have made of it anyway)
        inv[j] = rsqrt_f64(s);
        L[(j * j + j) / 2 + j] = s * inv[j];
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double t = a[(i * i + i) / 2 + j];
#pragma unroll
            for (int k = 0; k < j; k++) t -= L[(i * i + i) / 2 + k] * L[(j * j + j) / 2 + k];
            L[(i * i + i) / 2 + j] = t * inv[j];
        }
    }
    // what the gate pass needs: the strict lower triangle of L and the reciprocal diagonal, for a forward substitution per sample
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
        for (int k = 0; k < i; k++) W[(i * i + i) / 2 + k] = (float)L[(i * i + i) / 2 + k];
        W[(i * i + i) / 2 + i] = (float)inv[i];
    }
    return ok;
}

// Distance partition of the pool for the refit.  The gate of iteration t keeps the samples with (x - mu)^T C^-1 (x - mu) < sigma^2; all of
// them lie in the ball |x - m0| < |mu - m0| + sigma sqrt(trace C) around the mean-shift mode m0 (lambda_max <= trace; the shrinkage
// keeps the trace), and that ball contracts with the gate: ~7000 of 8192 samples at the start, ~3000 when the gate holds ~1300.  A sample
// outside the ball has weight exactly 0 -- leaving it out changes no sum.  So the pool is re-dealt ONCE, by distance from m0: pair-slot p
// of every lane (the 1024 samples a pass handles in one iteration of its unrolled loop) receives the samples of distance rank
// [1024 p, 1024 p + 1024) up to the resolution of the classes below, and a pass stops at the first pair-slot whose lower distance
// bound r2[p] lies outside the ball (measured on 640x480 pools: 45-50 % of the pair-slots over the ~35 iterations of a refit).
//   1. histogram of the squared distances over 1024 logarithmic bins (1/16 octave: the top 13 bits of the float), LDS atomics on integers
//   2. its prefix sum splits the bins into 16 classes of ~512 ranks; class c starts at rank s_cstart[c]
//   3. rank of a sample inside its class in a FIXED order (lane-interleaved rows, then slot): per-thread class counters (own row of
//      an LDS table: the returned old value is the rank among the thread's own slots), then a prefix over the threads per class
//   4. the six coordinates move to position = class start + rank through an LDS staging plane (position q -> lane q % 512, slot q / 512)
// Every step is order-independent or in a fixed order: the dealt pool, hence every sum of the refit, is the same from run to run.
// r2[p]: lower edge of the first bin of the class that holds rank 1024 p (every sample at a rank >= 1024 p is at least that far out).
template <int THREADS>
__device__ __forceinline__ void refit_partition(f2 (&X)[PM_POOL / THREADS / 2][6], const float (&m0)[6], float (&r2)[PM_POOL / THREADS / 2]) {
    static_assert(THREADS == 512 && PM_POOL == 8192, "the class / chunk geometry below is written for 512 threads x 16 samples");
    constexpr int SPT = PM_POOL / THREADS, PAIRS = SPT / 2, NCLS = 16, NBIN = 1024, ROW = NCLS + 1 /* padded: conflict-free across lanes */;
    constexpr int BIN_BASE = 1648;  // (bits >> 19) of 2^-24: bins span 2^-24 .. 2^40, clamped at both ends
    __shared__ unsigned s_hist[NBIN];   // counts, then the class of a bin
    __shared__ unsigned s_cedge[NCLS];  // float bits: lowest bin edge of a class
    __shared__ int s_bclass[PAIRS], s_wtot[THREADS / 64];
    __shared__ unsigned s_cstart[NCLS];  // rank at which a class starts = exclusive bin prefix of its first occupied bin
    __shared__ float4 s_big[PM_POOL];    // 128 KB: first the class counters [THREADS][ROW], then the staging plane(s)
    unsigned* cnt = reinterpret_cast<unsigned*>(&s_big[0]);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    s_hist[tid] = 0u; s_hist[tid + THREADS] = 0u;
    if (tid < NCLS) { s_cedge[tid] = 0x7f800000u; s_cstart[tid] = 0x7fffffffu; }
    if (tid < PAIRS) s_bclass[tid] = 0;  // (every rank lies in some bin, so each entry is overwritten below; class 0 = the bound that never skips)
#pragma unroll
    for (int c = 0; c < NCLS; c++) cnt[tid * ROW + c] = 0u;
    __syncthreads();
    int bin[SPT];
#pragma unroll
    for (int p = 0; p < PAIRS; p++) {
        f2 d2 = { 0.f, 0.f };
#pragma unroll
        for (int d = 0; d < 6; d++) { const f2 df = X[p][d] - f2{ m0[d], m0[d] }; d2 += df * df; }
        bin[2 * p] = min(max((int)(__float_as_uint(d2.x) >> 19) - BIN_BASE, 0), NBIN - 1);  // d2 >= 0 or +inf (a dropped hypothesis): no sign, no NaN
        bin[2 * p + 1] = min(max((int)(__float_as_uint(d2.y) >> 19) - BIN_BASE, 0), NBIN - 1);
    }
#pragma unroll
    for (int k = 0; k < SPT; k++) atomicAdd(&s_hist[bin[k]], 1u);
    __syncthreads();
    {   // exclusive prefix over the bins (two per thread) -> classes, class edges, the class that holds rank 1024 p
        const int v0 = (int)s_hist[2 * tid], v1 = (int)s_hist[2 * tid + 1], sum = v0 + v1;
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) s_wtot[wv] = incl;
        __syncthreads();
        int woff = 0;
#pragma unroll
        for (int j = 0; j < THREADS / 64; j++) woff += j < wv ? s_wtot[j] : 0;
        const int e0 = woff + incl - sum, e1 = e0 + v0;
        const int c0 = min(e0 >> 9, NCLS - 1), c1 = min(e1 >> 9, NCLS - 1);
        s_hist[2 * tid] = (unsigned)c0; s_hist[2 * tid + 1] = (unsigned)c1;  // own entries: nobody else has read or will read the counts
        if (v0) { atomicMin(&s_cedge[c0], tid == 0 ? 0u : (unsigned)(2 * tid + BIN_BASE) << 19); atomicMin(&s_cstart[c0], (unsigned)e0); }  // bin 0 also holds everything below its edge
        if (v1) { atomicMin(&s_cedge[c1], (unsigned)(2 * tid + 1 + BIN_BASE) << 19); atomicMin(&s_cstart[c1], (unsigned)e1); }
#pragma unroll
        for (int p = 1; p < PAIRS; p++) {
            const int pos = p * 2 * THREADS;
            if (v0 && e0 <= pos && pos < e0 + v0) s_bclass[p] = c0;
            if (v1 && e1 <= pos && pos < e1 + v1) s_bclass[p] = c1;
        }
    }
    __syncthreads();
    int cls[SPT], myrank[SPT];
#pragma unroll
    for (int k = 0; k < SPT; k++) cls[k] = (int)s_hist[bin[k]];
#pragma unroll
    for (int k = 0; k < SPT; k++) myrank[k] = (int)atomicAdd(&cnt[tid * ROW + cls[k]], 1u);  // own row: rank among my earlier slots of that class
    __syncthreads();
    {   // per class: prefix of the counters over the threads, on top of the class start.  Thread (c, j) owns rows j, j + 32, .. (lanes one padded row apart)
        const int c = tid >> 5, j = tid & 31;
        int a[THREADS / 32], run = 0;
#pragma unroll
        for (int i = 0; i < THREADS / 32; i++) { a[i] = (int)cnt[(i * 32 + j) * ROW + c]; run += a[i]; }
        int incl = run;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) { const int t = __shfl_up(incl, o, 32); if (j >= o) incl += t; }
        int base = (int)s_cstart[c] + incl - run;  // (an empty class: every a[i] is 0 and nobody reads the row entries)
#pragma unroll
        for (int i = 0; i < THREADS / 32; i++) { cnt[(i * 32 + j) * ROW + c] = (unsigned)base; base += a[i]; }
    }
    __syncthreads();
    int q[SPT];
#pragma unroll
    for (int k = 0; k < SPT; k++) q[k] = (int)cnt[tid * ROW + cls[k]] + myrank[k];
    __syncthreads();  // the counters are dead: their memory becomes the staging planes
    // coordinates 0..3 as one 16-byte plane, then 4..5 as an 8-byte plane: 32 scattered writes per thread instead of 96
#pragma unroll
    for (int k = 0; k < SPT; k++)
        s_big[q[k]] = (k & 1) ? float4{ X[k >> 1][0].y, X[k >> 1][1].y, X[k >> 1][2].y, X[k >> 1][3].y } : float4{ X[k >> 1][0].x, X[k >> 1][1].x, X[k >> 1][2].x, X[k >> 1][3].x };
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SPT; k++) {
        const float4 v = s_big[k * THREADS + tid];
        if (k & 1) { X[k >> 1][0].y = v.x; X[k >> 1][1].y = v.y; X[k >> 1][2].y = v.z; X[k >> 1][3].y = v.w; }
        else { X[k >> 1][0].x = v.x; X[k >> 1][1].x = v.y; X[k >> 1][2].x = v.z; X[k >> 1][3].x = v.w; }
    }
    __syncthreads();
    float2* s_two = reinterpret_cast<float2*>(&s_big[0]);
#pragma unroll
    for (int k = 0; k < SPT; k++) s_two[q[k]] = (k & 1) ? float2{ X[k >> 1][4].y, X[k >> 1][5].y } : float2{ X[k >> 1][4].x, X[k >> 1][5].x };
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SPT; k++) {
        const float2 v = s_two[k * THREADS + tid];
        if (k & 1) { X[k >> 1][4].y = v.x; X[k >> 1][5].y = v.y; } else { X[k >> 1][4].x = v.x; X[k >> 1][5].x = v.y; }
    }
    __syncthreads();
    r2[0] = 0.f;
#pragma unroll
    for (int p = 1; p < PAIRS; p++) r2[p] = __uint_as_float(s_cedge[s_bclass[p]]);
}

// robust-Gaussian refit + finalisation (geometry.cpp:201-263, fit_robust_gaussian.cu:131-263); runs
// on the last EM iteration only, ~30-50 gate/refit iterations per camera, each of them dependent phases on ONE compute unit: the gate +
// moment pass over the pool (VALU-issue bound), a 28-value all-reduce, the next gate's coefficients.  The 8192 scaled hypotheses stay
// in registers for the whole loop as pairs (two per lane, packed fp32 arithmetic as in k_pose_mode); 512 threads, because everything
// that is not the pass is executed by every wave.  The gate weight multiplies instead of branching (a wave practically always holds a
// gated sample).  After the all-reduce every lane holds the 28 totals and derives the new mean, the regularised covariance and the
// whitening factor itself (rg_whiten): ONE workgroup barrier per gate iteration, no serial section on a single wave.
// It continues k_pose_mode<true> in the same kernel: the pool is already in registers (scaled for the mean-shift metric, non-finite
// hypotheses as MS_FAR: their Mahalanobis distance is huge, infinite or NaN, never inside the gate, and 0 * MS_FAR = 0 in the sums).
// ms_mean / ms_conf / ms_iters / used: the mean-shift result, the same in every thread.
template <int THREADS>
__device__ __forceinline__ void refit_block(f2 (&X)[PM_POOL / THREADS / 2][6], int used, const float (&ms_mean)[6], float ms_conf, int ms_iters,
                                            const ModeParams& mp, CamState* cam, PoseBlock* P, int cam_idx, RedBuf& rb) {
#pragma clang fp contract(fast)  // the gate / scatter sums are not part of the solver's exact-rounding contract
    constexpr int RF_PAIRS = PM_POOL / THREADS / 2, RF_NW = THREADS / 64;
    PH_DECL;
    const int tid = threadIdx.x;
    const float sc = mp.rg_pose_scaling;
    // x = [rvec * rvec_scale | t] * rg_pose_scaling (geometry.cpp:191,211)
#pragma unroll
    for (int p = 0; p < RF_PAIRS; p++)
#pragma unroll
        for (int d = 0; d < 6; d++) X[p][d] *= f2{ sc, sc };
    float mean[6], cov[21], W[21];
#pragma unroll
    for (int d = 0; d < 6; d++) mean[d] = ms_mean[d] * sc;
#pragma unroll
    for (int k = 0; k < 21; k++) cov[k] = 0.f;
#pragma unroll
    for (int d = 0; d < 6; d++) cov[(d * d + d) / 2 + d] = mp.kernel_var * (sc * sc);  // :203-206
    const bool regularise = mp.rg_covar_reg_lambda > 0.f;
    // uniform values the whole loop needs go through v_readfirstlane: scalar registers, not 14 of the 256 vector registers
    unsigned r2b[RF_PAIRS];  // bounds as float bits (non-negative floats order like their bits: a scalar compare decides a pair-slot)
    float m0[6];
    {
        float r2[RF_PAIRS];
#pragma unroll
        for (int p = 0; p < RF_PAIRS; p++) r2[p] = 0.f;  // without the partition: nothing is ever outside
        if (mp.rg_partition) refit_partition<THREADS>(X, mean, r2);  // `mean` still is the mean-shift mode here
#pragma unroll
        for (int p = 0; p < RF_PAIRS; p++) r2b[p] = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(r2[p]));
#pragma unroll
        for (int d = 0; d < 6; d++) m0[d] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, mean[d])));
    }
    bool pd = rg_whiten(cov, false, mp.rg_covar_reg_lambda, W);  // iteration 0 is not regularised (fit_robust_gaussian.cu:180)
    bool cov_reg = false;  // `cov` is what the last prepare step regularised (the covariance the reference holds when the loop ends)
    const float sig2 = mp.rg_trunc_sigma * mp.rg_trunc_sigma;
    float weight = 0.f, density = 0.f;
    int iter = 0, parity = 0;
    bool reliable = true;
    PH_MARK(24);
    for (iter = 0; iter < mp.rg_max_iters; iter++) {
        if (!pd) { reliable = false; break; }
        const float prev_density = density;
        // e_step (fit_robust_gaussian.cu:56-97): gate, weight, weighted sample and weighted scatter about
        // the CURRENT mean in one pass -> one 28-value all-reduce per iteration
        f2 acc2[28];
#pragma unroll
        for (int k = 0; k < 28; k++) acc2[k] = f2{ 0.f, 0.f };
        // the ball that holds every sample of this gate (refit_partition), 0.1 % wider than the bound: float rounding of the distances and
        // of z stays far inside that margin.  (The shrinkage keeps the trace.)
        float dm2 = 0.f, tr = 0.f;
#pragma unroll
        for (int d = 0; d < 6; d++) { dm2 = fmaf(mean[d] - m0[d], mean[d] - m0[d], dm2); tr += cov[(d * d + d) / 2 + d]; }
        const float ball = (__builtin_amdgcn_sqrtf(dm2) + mp.rg_trunc_sigma * __builtin_amdgcn_sqrtf(tr)) * 1.001f;  // v_sqrt_f32 (1 ulp): the margin is a thousand times that
        const unsigned ballb = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(ball * ball));  // a NaN (degenerate covariance) is above every bound: skips nothing
#pragma unroll
        for (int p = 0; p < RF_PAIRS; p++) {
            if (r2b[p] > ballb) continue;  // scalar compare and branch
            PH_ADD(34, 1);
            f2 diff[6];
#pragma unroll
            for (int d = 0; d < 6; d++) diff[d] = X[p][d] - f2{ mean[d], mean[d] };
            // z = |y|^2, L y = d by forward substitution (W: strict lower triangle of L, reciprocal diagonal): 21 + 6 multiply-adds
            f2 z = { 0.f, 0.f }, y[6];
#pragma unroll
            for (int d1 = 0; d1 < 6; d1++) {
                f2 t = diff[d1];
#pragma unroll
                for (int d2 = 0; d2 < d1; d2++) t -= y[d2] * W[(d1 * d1 + d1) / 2 + d2];
                y[d1] = t * W[(d1 * d1 + d1) / 2 + d1];
                z += y[d1] * y[d1];
            }
            // sqrt(z) < sigma (fit_robust_gaussian.cu:80) as z < sigma^2 (z is a sum of squares: >= 0 or NaN, and NaN stays outside)
            const f2 wgt = { z.x < sig2 ? 1.f : 0.f, z.y < sig2 ? 1.f : 0.f };
            acc2[0] += wgt;
#pragma unroll
            for (int d = 0; d < 6; d++) acc2[1 + d] += wgt * X[p][d];
            f2 wd[6];
#pragma unroll
            for (int d = 0; d < 6; d++) wd[d] = wgt * diff[d];
#pragma unroll
            for (int d1 = 0; d1 < 6; d1++)
#pragma unroll
                for (int d2 = 0; d2 <= d1; d2++) acc2[7 + (d1 * d1 + d1) / 2 + d2] += wd[d1] * diff[d2];
        }
        float acc[28];
#pragma unroll
        for (int k = 0; k < 28; k++) acc[k] = acc2[k].x + acc2[k].y;
#ifdef VK_PHASE_CLOCKS
        if (acc[0] == -123456.f) return;
#endif
        PH_MARK(26);
        const float tot = allreduce_lanes<28, RF_NW>(acc, rb, parity); parity ^= 1;  // lane k of every wave: total of sum k
        weight = lane_value(tot, 0);
        if (!isfinite(weight)) { reliable = false; break; }
        density = weight / (float)used;
        if (fabsf(density - prev_density) < mp.rg_epsilon) { reliable = true; break; }
        PH_MARK(27);
        // M-step (:234-246): mean and covariance of the gated set (no -1: it is regularised next), one division per lane where the
        // totals sit, broadcast; then -- unless this was the last iteration -- the regularised covariance and its whitening factor
        const float q = tot / weight;
#pragma unroll
        for (int d = 0; d < 6; d++) mean[d] = lane_value(q, 1 + d);
#pragma unroll
        for (int k = 0; k < 21; k++) cov[k] = lane_value(q, 7 + k);
        cov_reg = false;
        if (iter + 1 < mp.rg_max_iters) { pd = rg_whiten(cov, regularise, mp.rg_covar_reg_lambda, W); cov_reg = regularise; }
        PH_MARK(25);
    }
    PH_MARK(27); PH_ADD(28, iter); PH_ADD(29, 1);
    if (tid == 0) {
        float mean6[6];
        density = ms_conf;
        int gu_iters = 0;  // fit_robust_gaussian.cu:158-159 resets *used_iters on entry
        if (reliable) {
            float covr[21];
            rg_regularised(cov, cov_reg, mp.rg_covar_reg_lambda, covr);
            density = weight / (float)used; gu_iters = iter;
            for (int i1 = 0; i1 < 6; i1++)
                for (int i2 = 0; i2 < 6; i2++) {
                    const int hi = i1 >= i2 ? i1 : i2, lo = i1 >= i2 ? i2 : i1;
                    float c = covr[(hi * hi + hi) / 2 + lo] / (sc * sc);  // :224-233
                    if (i1 < 3 || i2 < 3) c /= mp.rvec_scale;
                    if (i1 < 3 && i2 < 3) c /= mp.rvec_scale;
                    cam->covar[i1 * 6 + i2] = c;
                }
            for (int d = 0; d < 6; d++) mean6[d] = mean[d] / sc;
        } else {
            for (int k = 0; k < 36; k++) cam->covar[k] = 0.f;
            for (int d = 0; d < 6; d++) mean6[d] = (ms_mean[d] * sc) / sc;  // pose_opm *= sc; /= sc (:210,:238)
        }
        finalize_pose(mean6, mp.rvec_scale, used, density, ms_iters, gu_iters, cam, P, cam_idx);
        maybe_decide(mp, P, cam, cam_idx);
    }
    PH_MARK(30);
}

// THREADS: the per-iteration all-reduce, the mean update and the convergence test are executed by every wave (~100 instructions next
// to ~30 per pair of hypotheses), so fewer, fatter waves do less redundant work: THREADS * SPT = PM_POOL.
template <bool DEFER, int THREADS>
__device__ __forceinline__ static void pose_mode_body(const float* __restrict__ rvecs, const float* __restrict__ tvecs,
                                                      int n_poses, const ModeParams& mp, CamState* cam, PoseBlock* P, int cam_idx,
                                                      const int* __restrict__ n_points_dev, const float* __restrict__ trials_in) {
#pragma clang fp contract(fast)  // kernel-weighted sums: not part of the solver's exact-rounding contract (file-wide: off)
    constexpr int SPT = PM_POOL / THREADS, MS_PAIRS = SPT / 2, NW = THREADS / 64;
    __shared__ RedBuf rb;
    __shared__ int s_cnt[SPT][NW];
    __shared__ float s_pick[MS_TRIAL_BATCH][6];
    PH_DECL;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // successive pose? (voldor.cpp:177: pose_sample_count != 0), decided on the device
    const bool external_init = mp.use_external_init_mean < 0 ? (cam->pose_sample_count != 0) : (mp.use_external_init_mean != 0);
    // ---- pool -> registers (coordinate planes [3][n_poses] written by k_solve: every load of a wave is one 256-byte run);
    // rvec pre-scaled (:191); number of finite hypotheses (geometry.cpp:156-165)
    f2 X[MS_PAIRS][6];
    unsigned finmask = 0;
#pragma unroll
    for (int k = 0; k < SPT; k++) {  // all loads first, straight into their registers
        const int i = min(k * THREADS + tid, n_poses - 1);  // slots past the pool read a valid address and are dropped below
#pragma unroll
        for (int d = 0; d < 6; d++) {
            const float v = d < 3 ? rvecs[(size_t)d * n_poses + i] : tvecs[(size_t)(d - 3) * n_poses + i];
            if (k & 1) X[k >> 1][d].y = v; else X[k >> 1][d].x = v;
        }
    }
    if (*n_points_dev < 4) {  // geometry.cpp:84-88 (tested with the pool already on its way: the buffers exist either way)
        if (tid == 0) { cam->success = 0; maybe_decide(mp, P, cam, cam_idx); }
        return;
    }
#pragma unroll
    for (int k = 0; k < SPT; k++) {
        float v[6];
#pragma unroll
        for (int d = 0; d < 6; d++) v[d] = (k & 1) ? X[k >> 1][d].y : X[k >> 1][d].x;
        const bool fin = k * THREADS + tid < n_poses && isfinite(v[0] + v[1] + v[2] + v[3] + v[4] + v[5]);
        const unsigned long long b = __ballot(fin);
        if (fin) finmask |= 1u << k;
        if (lane == 0) s_cnt[k][wv] = __popcll(b);
#pragma unroll
        for (int d = 0; d < 6; d++) {
            const float u = fin ? (d < 3 ? v[d] * mp.rvec_scale : v[d]) : MS_FAR;
            if (k & 1) X[k >> 1][d].y = u; else X[k >> 1][d].x = u;
        }
    }
    __syncthreads();
    int used = 0;
#pragma unroll
    for (int k = 0; k < SPT; k++)
#pragma unroll
        for (int j = 0; j < NW; j++) used += s_cnt[k][j];
    if (used == 0) {
        if (tid == 0) { cam->success = 0; maybe_decide(mp, P, cam, cam_idx); }
        return;
    }
    PH_MARK(16);
    // ---- mean-shift (meanshift.cu:34-150)
    float io_mean[6], c_mean[6];
#pragma unroll
    for (int d = 0; d < 3; d++) { io_mean[d] = cam->rvec[d] * mp.rvec_scale; io_mean[3 + d] = cam->t[d]; }
    const float inv2v = 1.f / (2.f * mp.kernel_var);
    const f2 ninv2v = { -inv2v, -inv2v };
    int parity = 0;
    if (external_init) {
#pragma unroll
        for (int d = 0; d < 6; d++) c_mean[d] = io_mean[d];
    } else {  // best of <= max_init_trials random hypotheses (meanshift.cu:72-95), host rand() -> rng3
        float best = 0.f;
        float bestx[6] = { 0, 0, 0, 0, 0, 0 };
        bool have = false;
        if (trials_in) {  // the densities of the trial hypotheses were evaluated by k_mode_trials, one workgroup per trial: replay the rule
            for (int t = 0; t < mp.ms_max_init_trials; t++) {
                const float dens = trials_in[t * 8];
                if (dens > best) {
                    best = dens;
#pragma unroll
                    for (int d = 0; d < 6; d++) bestx[d] = trials_in[t * 8 + 1 + d];
                    have = true;
                }
                if (best > mp.ms_good_init_confidence * (float)used) break;
            }
        } else {
        // rank of each of this thread's finite hypotheses in index order (slot-major, then thread), in registers for the
        // duration of the trials
        int rank[SPT];
        {
            int base = 0;
#pragma unroll
            for (int k = 0; k < SPT; k++) {
                int wbase = base;
#pragma unroll
                for (int j = 0; j < NW; j++) { if (j < wv) wbase += s_cnt[k][j]; base += s_cnt[k][j]; }
                const unsigned long long b = __ballot((finmask >> k) & 1u);
                rank[k] = ((finmask >> k) & 1u) ? wbase + __popcll(b & ((1ull << lane) - 1ull)) : -1;
            }
        }
        // The trials do not depend on one another (trial t looks at the (rng(t) % used)-th finite hypothesis); only the
        // "good enough, stop" test is sequential.  MS_TRIAL_BATCH of them share one pass over the pool and one all-reduce,
        // and the reference's sequential rule is replayed on the totals in trial order: same picks, same early stop.
        bool stop = false;
        for (int t0 = 0; t0 < mp.ms_max_init_trials && !stop; t0 += MS_TRIAL_BATCH) {
            const int nb = min(MS_TRIAL_BATCH, mp.ms_max_init_trials - t0);
#pragma unroll
            for (int b = 0; b < MS_TRIAL_BATCH; b++) {
                if (b < nb) {
                    const int target = (int)(rng3(RAND_SEED, (uint32_t)(t0 + b), 0x4D53u) % (uint32_t)used);
                    int kk = -1;  // which of my slots holds the target-th finite hypothesis (at most one thread has one)
#pragma unroll
                    for (int k = 0; k < SPT; k++) kk = rank[k] == target ? k : kk;
                    if (kk >= 0) {
                        float v[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll
                        for (int k = 0; k < SPT; k++)
#pragma unroll
                            for (int d = 0; d < 6; d++) v[d] = k == kk ? ((k & 1) ? X[k >> 1][d].y : X[k >> 1][d].x) : v[d];
#pragma unroll
                        for (int d = 0; d < 6; d++) s_pick[b][d] = v[d];
                    }
                }
            }
            __syncthreads();
            f2 acc2[MS_TRIAL_BATCH];
#pragma unroll
            for (int b = 0; b < MS_TRIAL_BATCH; b++) {
                acc2[b] = f2{ 0.f, 0.f };
                if (b < nb) {  // uniform
                    f2 c[6];
#pragma unroll
                    for (int d = 0; d < 6; d++) { const float v = s_pick[b][d]; c[d] = f2{ v, v }; }
#pragma unroll
                    for (int p = 0; p < MS_PAIRS; p++) {
                        f2 l2 = { 0.f, 0.f };
#pragma unroll
                        for (int d = 0; d < 6; d++) { const f2 df = X[p][d] - c[d]; l2 += df * df; }
                        const f2 a = l2 * ninv2v;
                        acc2[b] += f2{ __expf(a.x), __expf(a.y) };
                    }
                }
            }
            float acc[MS_TRIAL_BATCH];
#pragma unroll
            for (int b = 0; b < MS_TRIAL_BATCH; b++) acc[b] = acc2[b].x + acc2[b].y;
            allreduce_regs<MS_TRIAL_BATCH, NW>(acc, rb, parity); parity ^= 1;
#pragma unroll
            for (int b = 0; b < MS_TRIAL_BATCH; b++) {
                if (b < nb && !stop) {
                    if (acc[b] > best) {
                        best = acc[b];
#pragma unroll
                        for (int d = 0; d < 6; d++) bestx[d] = s_pick[b][d];
                        have = true;
                    }
                    if (best > mp.ms_good_init_confidence * (float)used) stop = true;
                }
            }
            __syncthreads();  // s_pick is rewritten by the next batch
        }
        }
        if (!have) {  // no trial had positive weight (the reference would index element -1)
#pragma unroll
            for (int d = 0; d < 6; d++) bestx[d] = io_mean[d];
        }
#pragma unroll
        for (int d = 0; d < 6; d++) c_mean[d] = bestx[d];
    }
    PH_MARK(17);
    int ms_iters = 0;
    float conf = 0.f;
    for (int iter = 0; iter < mp.ms_max_iters; iter++) {  // meanshift.cu:103-134
        f2 c[6];
#pragma unroll
        for (int d = 0; d < 6; d++) c[d] = f2{ c_mean[d], c_mean[d] };
        f2 acc2[7];
#pragma unroll
        for (int k = 0; k < 7; k++) acc2[k] = f2{ 0.f, 0.f };
#pragma unroll
        for (int p = 0; p < MS_PAIRS; p++) {
            f2 l2 = { 0.f, 0.f };
#pragma unroll
            for (int d = 0; d < 6; d++) { const f2 df = X[p][d] - c[d]; l2 += df * df; }
            const f2 a = l2 * ninv2v;
            const f2 wgt = { __expf(a.x), __expf(a.y) };
            acc2[0] += wgt;
#pragma unroll
            for (int d = 0; d < 6; d++) acc2[1 + d] += wgt * X[p][d];
        }
        float acc[7];
#pragma unroll
        for (int k = 0; k < 7; k++) acc[k] = acc2[k].x + acc2[k].y;
#ifdef VK_PHASE_CLOCKS
        if (acc[0] == 123456.f) return;
#endif
        PH_MARK(32);
        const float tot = allreduce_lanes<7, NW>(acc, rb, parity); parity ^= 1;  // lane k: total of sum k
        const float wsum = lane_value(tot, 0);
        conf = wsum;  // (divided by the sample count once, after the loop: off the per-iteration chain)
        ms_iters = iter + 1;
        const float quot = tot / wsum;  // lanes 1..6: the new mean, one division per wave instead of six per thread
        float disp = 0.f;
#pragma unroll
        for (int d = 0; d < 6; d++) {
            const float m = lane_value(quot, 1 + d);
            disp += (io_mean[d] - m) * (io_mean[d] - m);  // vs. the stale io mean on the first pass (SURVEY B-6)
            io_mean[d] = m; c_mean[d] = m;
        }
        if (__builtin_amdgcn_sqrtf(disp) < mp.ms_epsilon) break;  // uniform: every thread holds the same totals (v_sqrt_f32: the test is a threshold on a displacement, not a result)
        PH_MARK(33);
    }
    conf = conf / (float)used;
    PH_MARK(33); PH_ADD(19, ms_iters); PH_ADD(20, 1);
    if (DEFER) {  // robust-Gaussian refit on the same registers (geometry.cpp:201-263), which finalises the pose itself
        refit_block<THREADS>(X, used, io_mean, conf, ms_iters, mp, cam, P, cam_idx, rb);
        return;
    }
    if (tid == 0) {
        finalize_pose(io_mean, mp.rvec_scale, used, conf, ms_iters, cam->last_used_gu_iters, cam, P, cam_idx);
        PH_MARK(21);
        maybe_decide(mp, P, cam, cam_idx);
    }
    PH_MARK(22);
}
// fb_smooth blocks riding in the (non-refit) mode kernel's launch (FbRide, vk_common.hpp): workgroup q of the riders takes slots 2 q and 2 q + 1 of its
// range, one per 256-thread half, each half with its own LDS.  Slots of one workgroup belong to ONE stack (FbRide::split): both halves run the same pass
// on maps of one size in segments of one length -- the same instantiation, the same barriers.  A half past the range or in a stack's padding slot keeps the
// barriers company (fb_rows_body / fb_cols_body: a block past the job).
template <int SEG>
__device__ __forceinline__ void mode_fb_ride_seg(const FbRide& R, const FbStack* J, int b, bool in) {
    __shared__ FbMat s_fb[2][4][256];  // [half][sF 2 x 256 | sB 2 x 256]
    const int half = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const int bx = in ? b % J->blocks_x : (1 << 20), by = in ? b / J->blocks_x : 0;
    FbMat* sF = &s_fb[half][0][0]; FbMat* sB = &s_fb[half][2][0];
    if (R.kind == 1) {
        if (J->vec4) fb_rows_body<true, SEG>(J->src, J->dst, R.w, R.h, J->S, R.e0, R.p, bx, by, sF, sB, tid);
        else fb_rows_body<false, SEG>(J->src, J->dst, R.w, R.h, J->S, R.e0, R.p, bx, by, sF, sB, tid);
    } else fb_cols_body<SEG>(J->dst, R.w, R.h, J->S, J->CW, R.e0, R.p, bx, by, sF, sB, tid);
}
__device__ __forceinline__ void mode_fb_ride(const FbRide& R, int q) {
    const int half = threadIdx.x >> 8, slot = 2 * q + half, idx = R.first + slot;
    const int s = (R.first + 2 * q) < R.split ? 0 : 1;  // (first, split even: uniform over the workgroup)
    const FbStack* J = &R.st[s];
    const int b = idx - (s ? R.split : 0);
    const bool in = slot < R.count && b < J->n_blocks;
    if (J->seg == 12) mode_fb_ride_seg<12>(R, J, b, in);
    else if (J->seg == 20) mode_fb_ride_seg<20>(R, J, b, in);
    else mode_fb_ride_seg<40>(R, J, b, in);
}
template <bool DEFER, int THREADS>
__global__ __launch_bounds__(THREADS) static void k_pose_mode(const float* __restrict__ rvecs, const float* __restrict__ tvecs, int n_poses, ModeParams mp, CamState* cam,
                                                                  PoseBlock* P, int cam_idx, const int* __restrict__ n_points_dev, const float* __restrict__ trials_in, FbRide ride) {
    if constexpr (!DEFER && THREADS == 512) {
        if (blockIdx.x > 0) {  // what rides along on the 255 compute units the mode kernel leaves idle (workgroup 0 is the mode kernel itself)
            mode_fb_ride(ride, (int)blockIdx.x - 1);
            return;
        }
    }
    pose_mode_body<DEFER, THREADS>(rvecs, tvecs, n_poses, mp, cam, P, cam_idx, n_points_dev, trials_in);
    if constexpr (!DEFER) {
        // the last mode kernel of a pose half whose fb_smooth rode along also prepares the projective maps of the depth half and the world-scale factor
        // (vk_cum_poses.hpp): every exit of pose_mode_body has written the camera record and the truncation decision; thread 0's stores are pushed
        // out and the workgroup meets before anybody reads them back
        if (ride.cum_N >= 0) {  // (uniform)
            __threadfence();
            __syncthreads();
            cum_poses_block(P, ride.cum_N, ride.cum_Ndp, ride.world_scale);
        }
    }
}
// The initial-mode trials of a camera that has no pose yet (first EM iteration; meanshift.cu:72-95: the kernel density at up to
// max_init_trials random hypotheses, the densest one starts the mean shift) are independent of one another and each is a full pass
// over the pool: inside the single-workgroup mode kernel they cost 20 passes on one compute unit (~33 us).  Here trial b is
// workgroup b: same pool, same pick ((rng(b) % used)-th finite hypothesis in index order), density and hypothesis to
// out[b][0..6]; k_pose_mode replays the sequential "better than the best so far / good enough, stop" rule on them.
template <int THREADS>
__global__ __launch_bounds__(THREADS) static void k_mode_trials(const float* __restrict__ rvecs, const float* __restrict__ tvecs, int n_poses,
                                                                 ModeParams mp, const int* __restrict__ n_points_dev, float* __restrict__ out) {
#pragma clang fp contract(fast)
    constexpr int SPT = PM_POOL / THREADS, MS_PAIRS = SPT / 2, NW = THREADS / 64;
    __shared__ RedBuf rb;
    __shared__ int s_cnt[SPT][NW];
    __shared__ float s_pick[6];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, trial = blockIdx.x;
    f2 X[MS_PAIRS][6];
    unsigned finmask = 0;
#pragma unroll
    for (int k = 0; k < SPT; k++) {
        const int i = min(k * THREADS + tid, n_poses - 1);
#pragma unroll
        for (int d = 0; d < 6; d++) {
            const float v = d < 3 ? rvecs[(size_t)d * n_poses + i] : tvecs[(size_t)(d - 3) * n_poses + i];
            if (k & 1) X[k >> 1][d].y = v; else X[k >> 1][d].x = v;
        }
    }
    if (*n_points_dev < 4) return;
#pragma unroll
    for (int k = 0; k < SPT; k++) {
        float v[6];
#pragma unroll
        for (int d = 0; d < 6; d++) v[d] = (k & 1) ? X[k >> 1][d].y : X[k >> 1][d].x;
        const bool fin = k * THREADS + tid < n_poses && isfinite(v[0] + v[1] + v[2] + v[3] + v[4] + v[5]);
        const unsigned long long b = __ballot(fin);
        if (fin) finmask |= 1u << k;
        if (lane == 0) s_cnt[k][wv] = __popcll(b);
#pragma unroll
        for (int d = 0; d < 6; d++) {
            const float u = fin ? (d < 3 ? v[d] * mp.rvec_scale : v[d]) : MS_FAR;
            if (k & 1) X[k >> 1][d].y = u; else X[k >> 1][d].x = u;
        }
    }
    __syncthreads();
    int used = 0;
#pragma unroll
    for (int k = 0; k < SPT; k++)
#pragma unroll
        for (int j = 0; j < NW; j++) used += s_cnt[k][j];
    if (used == 0) return;
    const int target = (int)(rng3(RAND_SEED, (uint32_t)trial, 0x4D53u) % (uint32_t)used);
    {
        int base = 0;
#pragma unroll
        for (int k = 0; k < SPT; k++) {
            int wbase = base;
#pragma unroll
            for (int j = 0; j < NW; j++) { if (j < wv) wbase += s_cnt[k][j]; base += s_cnt[k][j]; }
            const unsigned long long b = __ballot((finmask >> k) & 1u);
            const int rank = ((finmask >> k) & 1u) ? wbase + __popcll(b & ((1ull << lane) - 1ull)) : -1;
            if (rank == target) {
#pragma unroll
                for (int d = 0; d < 6; d++) s_pick[d] = (k & 1) ? X[k >> 1][d].y : X[k >> 1][d].x;
            }
        }
    }
    __syncthreads();
    const float inv2v = 1.f / (2.f * mp.kernel_var);
    const f2 ninv2v = { -inv2v, -inv2v };
    f2 c[6];
#pragma unroll
    for (int d = 0; d < 6; d++) { const float v = s_pick[d]; c[d] = f2{ v, v }; }
    f2 acc2 = { 0.f, 0.f };
#pragma unroll
    for (int p = 0; p < MS_PAIRS; p++) {
        f2 l2 = { 0.f, 0.f };
#pragma unroll
        for (int d = 0; d < 6; d++) { const f2 df = X[p][d] - c[d]; l2 += df * df; }
        const f2 a = l2 * ninv2v;
        acc2 += f2{ __expf(a.x), __expf(a.y) };
    }
    float acc[1] = { acc2.x + acc2.y };
    allreduce_regs<1, NW>(acc, rb, 0);
    if (tid == 0) {
        out[trial * 8] = acc[0];
        for (int d = 0; d < 6; d++) out[trial * 8 + 1 + d] = s_pick[d];
    }
}

// ---- stand-alone kernels behind the host-pointer API (B-inner) -------------------------------
// io (floats): [0..15] mean in/out, [16] confidence/density out ; ioi (ints): [0] iters, [1] status
__global__ __launch_bounds__(MS_THREADS) static void k_meanshift_only(const float* __restrict__ space, int N, ModeParams mp,
                                                                       float* __restrict__ io, int* __restrict__ ioi) {
    __shared__ BlockRed br;
    __shared__ float s_mean[16], s_cmean[16];
    if (threadIdx.x < mp.dims) s_mean[threadIdx.x] = io[threadIdx.x];
    __syncthreads();
    float conf; int iters;
    meanshift_block(space, N, mp.dims, mp, s_mean, s_cmean, br, &conf, &iters);
    __syncthreads();
    if (threadIdx.x < mp.dims) io[threadIdx.x] = s_mean[threadIdx.x];
    if (threadIdx.x == 0) { io[16] = conf; ioi[0] = iters; ioi[1] = 0; }
}
// io: [0..5] mean in/out, [6..41] covar full in/out, [42] density out
__global__ __launch_bounds__(MS_THREADS) static void k_robust_gaussian_only(const float* __restrict__ space, int N, ModeParams mp,
                                                                             float* __restrict__ io, int* __restrict__ ioi) {
    __shared__ BlockRed br;
    __shared__ float s_mean[6], s_cov[21], s_cinv[21];
    const int dims = mp.dims, tid = threadIdx.x;
    if (tid < dims) s_mean[tid] = io[tid];
    if (tid == 0)
        for (int d1 = 0; d1 < dims; d1++)
            for (int d2 = 0; d2 <= d1; d2++) s_cov[(d1 * d1 + d1) / 2 + d2] = io[6 + d1 * dims + d2];
    __syncthreads();
    float dens; int it;
    bool ok = robust_gaussian_block(space, N, dims, 1.f, mp, s_mean, s_cov, s_cinv, br, &dens, &it);
    if (tid == 0) {
        ioi[1] = ok ? 0 : 1;
        if (ok) {
            ioi[0] = it;
            io[42] = dens;
            for (int d = 0; d < dims; d++) io[d] = s_mean[d];
            for (int d1 = 0; d1 < dims; d1++)
                for (int d2 = 0; d2 <= d1; d2++) {
                    io[6 + d1 * dims + d2] = s_cov[(d1 * d1 + d1) / 2 + d2];
                    io[6 + d2 * dims + d1] = s_cov[(d1 * d1 + d1) / 2 + d2];
                }
        }
    }
}

// ---- host launchers ------------------------------------------------------------------------------
int collect_device(Context* c, const ImageSet& S, int N, int w, int h, int active_idx, float rig_thresh, float rig_sum_thresh,
                   float min_depth, float max_depth, int max_trace, CamState* cam_dev, bool compact, bool block_compact, bool ref_tex) {
    const int npx = w * h, nblk = (npx + 255) / 256;
    if (int e = c->p2_map.reserve(sizeof(float) * 2 * (size_t)npx)) return e;
    if (int e = c->p3_map.reserve(sizeof(float) * 3 * (size_t)npx)) return e;
    if (int e = c->blk_counts.reserve(sizeof(int) * (size_t)nblk)) return e;
    if (int e = c->blk_offsets.reserve(sizeof(int) * (size_t)nblk)) return e;
    if (int e = c->valid_mask.reserve(sizeof(unsigned long long) * 4 * (size_t)nblk)) return e;
    if (int e = c->ensure_n_points()) return e;
    if (block_compact && compact) return (int)hipErrorInvalidValue;  // the ordered list of the host-pointer API is built from the NaN-marked maps
    // the density reduction the window pipeline left behind rides in the trace of camera 0 (which reads only depth, the weights of frame 0, the first
    // flow layer and the intrinsics); any other trace runs behind its own launch
    ReduceArgs ra;
    int extra = 0;
    if (c->pending_reduce.partial) {
        if (active_idx == 0) { ra = c->pending_reduce; ra.n_px_blocks = nblk; extra = ra.n_launch + (ra.scale_out ? 1 : 0); c->pending_reduce = ReduceArgs(); c->dbg_reduces_rode++; }
        else if (int e = flush_pending_reduce(c)) return e;
    }
    if (block_compact)
        hipLaunchKernelGGL(k_collect<true>, dim3(nblk + extra), dim3(256), 0, c->stream, S.flows.as<float2>(), S.rig.as<float>(), S.depth.as<float>(),
                           S.pb(), c->p2_map.as<float>(), c->p3_map.as<float>(), c->blk_counts.as<int>(), c->valid_mask.as<unsigned long long>(), N, w, h, active_idx,
                           rig_thresh, rig_sum_thresh, min_depth, max_depth, max_trace, ref_tex ? 1 : 0, ra);
    else
        hipLaunchKernelGGL(k_collect<false>, dim3(nblk + extra), dim3(256), 0, c->stream, S.flows.as<float2>(), S.rig.as<float>(), S.depth.as<float>(),
                           S.pb(), c->p2_map.as<float>(), c->p3_map.as<float>(), c->blk_counts.as<int>(), c->valid_mask.as<unsigned long long>(), N, w, h, active_idx,
                           rig_thresh, rig_sum_thresh, min_depth, max_depth, max_trace, ref_tex ? 1 : 0, ra);
    c->n_map_blocks = nblk;
    c->maps_block_compact = block_compact;
    if (compact) {  // the host-pointer API hands the compacted list to its caller (geometry.cpp:68-80)
        hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, c->stream, c->blk_counts.as<int>(), c->blk_offsets.as<int>(), nblk,
                           c->n_points.as<int>(), cam_dev);
        if (int e = c->pts2.reserve(sizeof(float) * 2 * (size_t)npx)) return e;
        if (int e = c->pts3.reserve(sizeof(float) * 3 * (size_t)npx)) return e;
        hipLaunchKernelGGL(k_compact, dim3(nblk), dim3(256), 0, c->stream, c->p2_map.as<float>(), c->p3_map.as<float>(),
                           c->blk_offsets.as<int>(), c->pts2.as<float>(), c->pts3.as<float>(), npx);
    }
    VK_CHECK_LAST();
    return 0;
}

template <bool FROM_MAP>
static int solve_launch(Context* c, const float* pts2, const float* pts3, int* n_pts_dev, CamState* cam, int npx, float fx, float fy,
                        float cx, float cy, int n_poses, int solver, int draw, bool strict, bool ref_svd, bool ref_rng) {
    const vrc_xorwow* xw = nullptr;
    if (ref_rng) {
        if (FROM_MAP && draw <= 0) { fprintf(stderr, "voldor_hip: --reference_rng 1 needs the reference's index draw (--reference_draw 1)\n"); return (int)hipErrorInvalidValue; }
        if (int e = xorwow_pose_states_device(c, n_poses)) return e;
        xw = c->xw_pose_states.as<vrc_xorwow>();
    }
    if (int e = c->rvecs.reserve(sizeof(float) * 3 * (size_t)n_poses)) return e;
    if (int e = c->tvecs.reserve(sizeof(float) * 3 * (size_t)n_poses)) return e;
    const int lph = (solver == 1) ? 1 : 4;  // lanes per hypothesis (k_solve)
    dim3 g((n_poses * lph + 63) / 64), b(64);
    float* rv = c->rvecs.as<float>(); float* tv = c->tvecs.as<float>();
    const int* bc = c->blk_counts.as<int>(); const int nb = FROM_MAP ? c->n_map_blocks : 0;
    // rank-select draw: the prefix of the block counts lives in the LDS of every workgroup (32 KB at 1080p: two workgroups of one
    // wave per CU still fit); not allocated when the draw cannot be taken (draw < 0).  Beyond 60 KB (3.9 MP) the counts are scanned
    // into global memory by one extra launch and the bisection reads them from there.
    size_t lds = (FROM_MAP && draw >= 0) ? sizeof(int) * ((size_t)nb + nb / 32 + 1) : 0;  // padded: pref_at() in k_solve
    const int* offs = nullptr;
    if (lds > 60 * 1024) {
        if (int e = c->blk_offsets.reserve(sizeof(int) * (size_t)nb)) return e;
        hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, c->stream, bc, c->blk_offsets.as<int>(), nb, n_pts_dev, (CamState*)nullptr);
        offs = c->blk_offsets.as<int>();
        lds = 0;
    }
    if (FROM_MAP && c->maps_block_compact && draw <= 0) {
        fprintf(stderr, "voldor_hip: block-compacted correspondences can only be drawn by rank (draw = 1)\n");
        return (int)hipErrorInvalidValue;
    }
    const int st = (strict ? 1 : 0) | (ref_svd ? 2 : 0) | ((FROM_MAP && c->maps_block_compact) ? 4 : 0);
    const unsigned long long* vm = FROM_MAP ? c->valid_mask.as<unsigned long long>() : nullptr;
    if (solver == 0) hipLaunchKernelGGL((k_solve<0, FROM_MAP>), g, b, lds, c->stream, pts2, pts3, rv, tv, n_pts_dev, bc, nb, cam, npx, fx, fy, cx, cy, n_poses, draw, st, offs, vm, xw);
    else if (solver == 1) hipLaunchKernelGGL((k_solve<1, FROM_MAP>), g, b, lds, c->stream, pts2, pts3, rv, tv, n_pts_dev, bc, nb, cam, npx, fx, fy, cx, cy, n_poses, draw, st, offs, vm, xw);
    else hipLaunchKernelGGL((k_solve<2, FROM_MAP>), g, b, lds, c->stream, pts2, pts3, rv, tv, n_pts_dev, bc, nb, cam, npx, fx, fy, cx, cy, n_poses, draw, st, offs, vm, xw);
    VK_CHECK_LAST();
    return 0;
}
int solve_device(Context* c, const float* pts2, const float* pts3, int* n_pts_dev, float fx, float fy, float cx, float cy,
                 int n_poses, int solver, bool strict, CamState* cam_dev, bool ref_svd, bool ref_rng) {
    return solve_launch<false>(c, pts2, pts3, n_pts_dev, cam_dev, 0, fx, fy, cx, cy, n_poses, solver, 0, strict, ref_svd, ref_rng);
}
int solve_from_maps_device(Context* c, int npx, float fx, float fy, float cx, float cy, int n_poses, int solver, CamState* cam_dev, int draw,
                           bool strict, bool ref_svd, bool ref_rng) {
    return solve_launch<true>(c, c->p2_map.as<float>(), c->p3_map.as<float>(), c->n_points.as<int>(), cam_dev, npx, fx, fy, cx, cy, n_poses,
                              solver, draw, strict, ref_svd, ref_rng);
}

// ---- --reference_rng 1: cuRAND XORWOW state tables (vk_ref_cuda.h) ------------------------------------------------------------------
// states[i] = the state after curand_init(RAND_SEED, first + i, 0) advanced by `epoch` outputs
__global__ __launch_bounds__(256) static void k_xorwow_init(vrc_xorwow* __restrict__ states, const uint32_t* __restrict__ J, int n, uint32_t epoch) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    vrc_xorwow s;
    vrc_xorwow_init(J, (unsigned long long)RAND_SEED, (uint32_t)i, &s);
    for (uint32_t e = 0; e < epoch; e++) (void)vrc_xorwow_next(&s);
    states[i] = s;
}
int xorwow_jumps_device(Context* c) {
    if (c->xw_jumps.p) return 0;
    std::vector<uint32_t> J((size_t)32 * VRC_XW_MAT);
    vrc_build_sequence_jumps(J.data());  // T^(2^67 2^k), k < 32: 98 products of 160 x 160 bit matrices, a few milliseconds once per context
    if (int e = c->xw_jumps.reserve(sizeof(uint32_t) * J.size())) return e;
    VK_CHECK(hipMemcpyAsync(c->xw_jumps.p, J.data(), sizeof(uint32_t) * J.size(), hipMemcpyHostToDevice, c->stream));
    VK_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}
int xorwow_pixel_states_device(Context* c, int npx, uint32_t epoch) {
    if (c->xw_px_states.p && c->xw_px_n == npx && c->xw_px_epoch == epoch) return 0;  // the states continue where the last sample launch left them
    if (int e = xorwow_jumps_device(c)) return e;
    if (int e = c->xw_px_states.reserve(sizeof(vrc_xorwow) * (size_t)npx)) return e;
    hipLaunchKernelGGL(k_xorwow_init, dim3((npx + 255) / 256), dim3(256), 0, c->stream, c->xw_px_states.as<vrc_xorwow>(), c->xw_jumps.as<uint32_t>(), npx, epoch);
    VK_CHECK_LAST();
    c->xw_px_n = npx; c->xw_px_epoch = epoch;
    return 0;
}
int xorwow_pose_states_device(Context* c, int n_poses) {
    if (c->xw_pose_states.p && c->xw_pose_n >= n_poses) return 0;
    if (int e = xorwow_jumps_device(c)) return e;
    if (int e = c->xw_pose_states.reserve(sizeof(vrc_xorwow) * (size_t)n_poses)) return e;
    hipLaunchKernelGGL(k_xorwow_init, dim3((n_poses + 255) / 256), dim3(256), 0, c->stream, c->xw_pose_states.as<vrc_xorwow>(), c->xw_jumps.as<uint32_t>(), n_poses, 0u);
    VK_CHECK_LAST();
    c->xw_pose_n = n_poses;
    return 0;
}

int pose_mode_device(Context* c, int n_poses, const ModeParams& mp_in, CamState* cam_dev, PoseBlock* P, int cam_idx, bool trials_first, const FbRide* ride_in) {
    ModeParams mp = mp_in;
    FbRide ride;
    if (ride_in) ride = *ride_in;
    if (mp.do_rg && (ride.kind != 0 || ride.cum_N >= 0)) return (int)hipErrorInvalidValue;  // (the refit kernel carries nothing: 141 KB of LDS)
    const int n_riders = ride.kind ? (ride.count + 1) / 2 : 0;
    if (ride.kind) c->dbg_fb_blocks_rode += ride.count;
    mp.rg_partition = debug_switches().refit_partition;
    if (n_poses > PM_POOL) {
        fprintf(stderr, "voldor_hip: n_poses_to_sample=%d exceeds the %d hypotheses the mode kernel keeps in registers\n", n_poses,
                PM_POOL);
        return (int)hipErrorInvalidValue;
    }
    // trials_first: the camera has no pose yet (the caller's copy of pose_sample_count is 0), so the mode kernel will look for a start
    // among random hypotheses: those trials run as their own workgroups first.  (If the device record disagrees the mode kernel ignores
    // them -- external start -- or, without them, runs the trials itself.)
    constexpr int MAX_SPLIT_TRIALS = 64;
    const float* trials = nullptr;
    if (trials_first && debug_switches().split_trials && mp.use_external_init_mean <= 0 && mp.ms_max_init_trials > 0 && mp.ms_max_init_trials <= MAX_SPLIT_TRIALS) {
        if (int e = c->ms_io.reserve(sizeof(float) * (64 + 8 * MAX_SPLIT_TRIALS) + sizeof(int) * 4)) return e;
        float* out = c->ms_io.as<float>() + 64;
        hipLaunchKernelGGL((k_mode_trials<PM_THREADS>), dim3(mp.ms_max_init_trials), dim3(PM_THREADS), 0, c->stream, c->rvecs.as<float>(),
                           c->tvecs.as<float>(), n_poses, mp, c->n_points.as<int>(), out);
        trials = out;
    }
    if (mp.do_rg)
        hipLaunchKernelGGL((k_pose_mode<true, PM_THREADS>), dim3(1), dim3(PM_THREADS), 0, c->stream, c->rvecs.as<float>(), c->tvecs.as<float>(), n_poses,
                           mp, cam_dev, P, cam_idx, c->n_points.as<int>(), trials, ride);
    else
        hipLaunchKernelGGL((k_pose_mode<false, PM_THREADS>), dim3(1 + n_riders), dim3(PM_THREADS), 0, c->stream, c->rvecs.as<float>(), c->tvecs.as<float>(), n_poses,
                           mp, cam_dev, P, cam_idx, c->n_points.as<int>(), trials, ride);
    VK_CHECK_LAST();
    return 0;
}

int meanshift_device(Context* c, const float* space_dev, int N, const ModeParams& mp, float* io_dev, int* ioi_dev) {
    hipLaunchKernelGGL(k_meanshift_only, dim3(1), dim3(MS_THREADS), 0, c->stream, space_dev, N, mp, io_dev, ioi_dev);
    VK_CHECK_LAST();
    return 0;
}
int robust_gaussian_device(Context* c, const float* space_dev, int N, const ModeParams& mp, float* io_dev, int* ioi_dev) {
    hipLaunchKernelGGL(k_robust_gaussian_only, dim3(1), dim3(MS_THREADS), 0, c->stream, space_dev, N, mp, io_dev, ioi_dev);
    VK_CHECK_LAST();
    return 0;
}

}  // namespace vk


#ifdef VK_PHASE_CLOCKS
extern "C" __attribute__((visibility("default"))) int vk_phase_read(unsigned long long* out, int n, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vk::g_phase), sizeof(unsigned long long) * (size_t)(n < 64 ? n : 64)) != hipSuccess) return 2;
    if (reset) { unsigned long long z[64] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(vk::g_phase), z, sizeof z) != hipSuccess) return 3; }
    return 0;
}
#endif
