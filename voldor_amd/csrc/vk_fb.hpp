// voldor_amd/csrc/vk_fb.hpp -- the segmented forward-backward smoothing of the fast path (gpu-kernels/fb_smooth.h:17-109) as device functions: the
// bodies of the row and the column pass (vk_depth.hip k_fb_rows / k_fb_cols).
#pragma once
#include "vk_common.hpp"
#include "vk_device.hpp"

namespace vk {

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
// Every function below evaluates with explicit fused multiply-adds and contraction OFF: what a pass computes does not depend on the build flags of the
// translation unit it is compiled into (round 5 compiled it into two: profiles/r05_summary.md).
__device__ __forceinline__ FbCoef fb_coef(float e0, float p) {
#pragma clang fp contract(off)
    FbCoef k;
    k.p = p; k.q = 1.f - p; k.dd = p - k.q; k.e0 = e0; k.e0p = e0 * p; k.e0dd = e0 * k.dd; k.qe0 = k.q * e0; k.pq = p + k.q;
    k.pqe0 = k.pq * e0;
    return k;
}
struct FbMat { float a, b, c, d; };  // acts on (x, 1-x): x' = (a x + b (1-x)) / ((a+c) x + (b+d)(1-x))
__device__ __forceinline__ float fb_apply(const FbMat& M, float x) {
#pragma clang fp contract(off)
    const float y = 1.f - x, n1 = fmaf(M.a, x, M.b * y), n0 = fmaf(M.c, x, M.d * y);
    return n1 * fast_rcp(n1 + n0);
}
// segment matrices: F = A_{n-1} ... A_0 with A_t = diag(e_t, e0) T ; B = C_0 ... C_{n-1} with C_t = T diag(e_t, e0)
template <int FB_SEG>
__device__ __forceinline__ void fb_compose(const FbCoef& K, const float (&e)[FB_SEG], int n, FbMat& F, FbMat& B) {
#pragma clang fp contract(off)
    F = { 1.f, 0.f, 0.f, 1.f }; B = { 1.f, 0.f, 0.f, 1.f };
#pragma unroll
    for (int k = 0; k < FB_SEG; k++) {
        if (k < n) {
            const float e1 = e[k];
            const float fa = fmaf(K.p, F.a, K.q * F.c) * e1, fb = fmaf(K.p, F.b, K.q * F.d) * e1;
            const float fc = fmaf(K.q, F.a, K.p * F.c) * K.e0, fd = fmaf(K.q, F.b, K.p * F.d) * K.e0;
            F = { fa, fb, fc, fd };
            const float ba = fmaf(B.a, K.p, B.b * K.q) * e1, bb = fmaf(B.a, K.q, B.b * K.p) * K.e0;
            const float bc = fmaf(B.c, K.p, B.d * K.q) * e1, bd = fmaf(B.c, K.q, B.d * K.p) * K.e0;
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
#pragma clang fp contract(off)
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
            xb = fmaf(fmaf(e[k], K.p, -K.qe0), xb, K.qe0) * fast_rcp(fmaf(fmaf(K.pq, e[k], -K.pqe0), xb, K.pqe0));
            const float a0 = (1.f - Fm[k]) * (1.f - xb), a1 = Fm[k] * xb;
            e[k] = a1 * fast_rcp(fmaf(Fm[k], xb, a0));
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
#pragma clang fp contract(off)
    FbMat r = { fmaf(M.a, N.a, M.b * N.c), fmaf(M.a, N.b, M.b * N.d), fmaf(M.c, N.a, M.d * N.c), fmaf(M.c, N.b, M.d * N.d) };
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
// pieces (160 bytes per lane).  256 threads = floor(256/S) rows.  (bx, by) = (row block, map); called by ALL threads of the workgroup (barriers): a
// 256-thread workgroup (tid = threadIdx.x), or the two 256-thread halves of a 512-thread one, each with its own block, tid = threadIdx.x & 255 and its
// own sF / sB ([2][256] each).  A block past the job (bx * lpb >= h; by any valid map) takes part in the barriers and does nothing else.
// maps_out == maps: in place.
struct __attribute__((packed, aligned(4))) FbQuad { float x, y, z, w; };  // 16 bytes at 4-byte alignment: one global_load_dwordx4 on gfx950
template <bool VEC4, int FB_SEG>
__device__ __forceinline__ void fb_rows_body(const float* maps, float* maps_out, int w, int h, int S, float e0, float p, int bx, int by, FbMat* sF, FbMat* sB, int tid) {
    const int lpb = 256 / S;
    const int ll = tid / S, seg = tid - ll * S, row = bx * lpb + ll;
    const bool live = ll < lpb && row < h;
    const int c0 = seg * FB_SEG, n = live ? min(FB_SEG, w - c0) : 0;
    const size_t line_off = (size_t)by * w * h + (size_t)(live ? row : 0) * w;
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
    sF[tid] = F; sB[tid] = B;
    __syncthreads();
    float xf, xb;
    fb_incoming_scan(sF, sB, 256, F, B, ll < lpb, tid, 1, seg, S, first, last, xf, xb);
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
// Column pass: thread = (segment, column); CW adjacent columns share a workgroup, so every access is a contiguous 4 CW-byte row piece and
// a workgroup uses CW * S threads (the own launch: CW = 16, up to 64 segments; as extra workgroups of a 256-thread launch: CW = 256 / S columns).
// (bx, by) = (column block, map); called by ALL threads of the workgroup (one barrier; threads beyond CW * S idle); sF / sB: [CW * S] each.
template <int FB_SEG>
__device__ __forceinline__ void fb_cols_body(float* __restrict__ maps, int w, int h, int S, int CW, float e0, float p, int bx, int by, FbMat* sF, FbMat* sB, int tid) {
    const int seg = tid / CW, cl = tid - seg * CW, col = bx * CW + cl;
    const bool live = seg < S && col < w;
    const int r0 = seg * FB_SEG, n = live ? min(FB_SEG, h - r0) : 0;
    float* m = maps + (size_t)by * w * h + (live ? col : 0);
    const FbCoef K = fb_coef(e0, p);
    float e[FB_SEG];
#pragma unroll
    for (int k = 0; k < FB_SEG; k++) e[k] = (k < n) ? m[(size_t)(r0 + k) * w] : 0.5f;
    const float first = m[0], last = m[(size_t)(h - 1) * w];
    FbMat F, B;
    fb_compose<FB_SEG>(K, e, n, F, B);
    if (seg < S) { sF[tid] = F; sB[tid] = B; }
    __syncthreads();
    if (!live) return;
    float xf, xb;
    fb_incoming(sF + cl, sB + cl, CW, seg, S, first, last, xf, xb);  // (the scan of the row pass does not pay here: measured 15.4 -> 16.3 us at 1241x376)
    fb_walk<FB_SEG>(K, e, n, xf, xb);
#pragma unroll
    for (int k = 0; k < FB_SEG; k++) if (k < n) m[(size_t)(r0 + k) * w] = e[k];
}
constexpr int FB_CW = 16;  // columns per workgroup of the column pass's own launch
constexpr int FB_MAX_ROW_SEGS = 256, FB_MAX_COL_SEGS = 1024 / FB_CW;

}  // namespace vk
