// voldor_amd/csrc/vk_device.hpp -- device-side math shared by the per-pixel kernels:
// counter-based RNG, log-logistic residual model, pinhole geometry, ALU bilinear fetch.
// CDNA has no image-sampler path, so the reference's tex2D gathers (gmat.h:175-179) become
// four explicit loads + a lerp; the residual model (residual_model.h:6-68) is re-derived so
// that one rigidness needs 5 transcendental ops instead of 6 powf + expf + 2 sqrtf.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "vk_common.hpp"

namespace vk {

// ---- RNG: stateless, keyed by (seed, stream, counter). Replaces the 48-byte-per-pixel
// cuRAND XORWOW state (optimize_depth.cu:269-291), whose read-modify-write per sample
// launch was the largest HBM stream of the reference M-step.
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ uint32_t rng3(uint32_t seed, uint32_t stream, uint32_t counter) {
    uint32_t h = fmix32(seed ^ 0x9E3779B9u);
    h = fmix32(h ^ stream);
    h = fmix32(h + counter * 0x9E3779B1u + 0x7F4A7C15u);
    return h;
}
// (0,1], same support as curand_uniform
__host__ __device__ __forceinline__ float u01(uint32_t r) {
    return (float)((r >> 8) + 1u) * (1.0f / 16777216.0f);
}

// ---- residual model -------------------------------------------------------------------
// pdf(x; c, s) = c * r^(-c-1) * (1 + r^(-c))^-2 / s with r = max(.5x, eps)^2 / s
// rigidness = p / (p + mu) where p = pdf(err), mu = pdf(lambda * |flow|): c and 1/s cancel:
//   mu/p = [q_mu / (r_mu (1+q_mu)^2)] * [r (1+q)^2 / q],   q = r^-c = exp2(-c log2 r)
// so  rigidness = 1 / (1 + mu/p)  and  -log(rigidness) = log(1 + mu/p).
// Hardware transcendental / reciprocal instructions (1 ulp class): v_log_f32, v_exp_f32, v_rcp_f32,
// v_sqrt_f32.  The per-pixel chains are latency-bound on dependent VALU work, so the IEEE
// div/sqrt expansions (10-15 instructions each) are kept out of the residual model.
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

struct FiskParams { float c, inv_s; };
__device__ __forceinline__ FiskParams fisk_params(float fmag) {
    float g = fminf(fmaxf(fmag * 0.5f, 2.f), 100.f);  // residual_model.h:16, :22
    FiskParams p;
    p.c = 1.0f - 0.0022f * g;                          // FISK_B1 + FISK_B2*g
    // s = 0.01*exp(0.09 g)  ->  1/s = 100 * exp2(-0.09*log2(e)*g)
    p.inv_s = 100.f * fast_exp2(-0.12984255368000671f * g);
    return p;
}
// t = mu/p from the SQUARED error magnitude e2 and squared strictness magnitude m2 (both already
// divided by abs_rf^2).  With l = log2 r, lm = log2 r_mu, q = 2^(-c l), q_mu = 2^(-c lm):
//   mu/p = (q_mu/q) (r/r_mu) ((1+q)/(1+q_mu))^2 = 2^((c+1)(l-lm)) ((1+q)/(1+q_mu))^2
// 2 log2 + 3 exp2 + 1 rcp; all quantities stay finite for e2,m2 >= 0 (r >= eps^2/s > 0).
__device__ __forceinline__ float fisk_ratio_sq(float e2, float m2, FiskParams fp) {
    const float eps2 = 1.1920929e-07f * 1.1920929e-07f;  // ZDE^2 (utils.h:19), x = max(.5x, ZDE)
    float re = fmaxf(0.25f * e2, eps2) * fp.inv_s, rm = fmaxf(0.25f * m2, eps2) * fp.inv_s;
    float l = fast_log2(re), lm = fast_log2(rm);
    float q = fast_exp2(-fp.c * l), qm = fast_exp2(-fp.c * lm);
    float a = (1.f + q) * fast_rcp(1.f + qm);
    return fast_exp2((fp.c + 1.f) * (l - lm)) * a * a;
}
__device__ __forceinline__ float flow_ratio(float dx1, float dy1, float dx2, float dy2, float lambda, float inv_arf) {
    const float ia2 = inv_arf * inv_arf;
    float obs2 = (dx2 * dx2 + dy2 * dy2) * ia2;
    float ex = dx1 - dx2, ey = dy1 - dy2;
    float e2 = (ex * ex + ey * ey) * ia2;
    return fisk_ratio_sq(e2, lambda * lambda * obs2, fisk_params(fast_sqrt(obs2)));
}
__device__ __forceinline__ float rigidness_from_flows(float dx1, float dy1, float dx2, float dy2,
                                                      float lambda, float inv_arf) {
    return fast_rcp(1.f + flow_ratio(dx1, dy1, dx2, dy2, lambda, inv_arf));
}
// -log(rigidness): cost contribution of one frame (residual_model.h:45-49)
__device__ __forceinline__ float neglog_rigidness_from_flows(float dx1, float dy1, float dx2, float dy2,
                                                             float lambda, float inv_arf) {
    return 0.6931471805599453f * fast_log2(1.f + flow_ratio(dx1, dy1, dx2, dy2, lambda, inv_arf));
}
// depth-prior variant on disparities (residual_model.h:51-68)
__device__ __forceinline__ float depth_ratio(float d1, float d2, float basefocal, float omega, float inv_arf) {
    float disp1 = (basefocal / d1) * inv_arf, disp2 = (basefocal / d2) * inv_arf;
    float dd = disp1 - disp2, om = omega * disp2;
    return fisk_ratio_sq(dd * dd, om * om, fisk_params(disp2));
}

// ---- fast path: the same model with the observation-only part split off ---------------------------------------------------
// Everything that depends on the OBSERVED flow alone (c, log2(1/s), the strictness term mu) is computed once per gather
// (and once per pixel for frame 0, whose observation does not depend on the depth hypothesis); a residual then costs
// 2 v_log + 2 v_exp.  log2(1/s) = log2(100) - 0.09 log2(e) g needs no exp at all, and the eps clamp of the strictness term
// moves into the log domain: log2(max(x, eps^2)) = max(log2 x, log2 eps^2).
struct __attribute__((aligned(8))) TexPair { float ax, ay, bx, by; };  // two horizontally adjacent flow texels
struct ObsTerms { float c, c1, ls, lm, rqm; };
__device__ __forceinline__ ObsTerms obs_terms(float ox, float oy, float ia2, float log2_qlam2 /* log2(0.25 lambda^2) */) {
#pragma clang fp contract(off)  // explicit fma only: every kernel must get the same bits (vk_depth.hip "one arithmetic")
    const float obs2 = fmaf(ox, ox, oy * oy) * ia2;
    const float g = __builtin_amdgcn_fmed3f(0.5f * fast_sqrt(obs2), 2.f, 100.f);  // residual_model.h:16
    ObsTerms t;
    t.c = fmaf(-0.0022f, g, 1.0f);
    t.c1 = t.c + 1.f;
    t.ls = fmaf(-0.12984255368000671f, g, 6.643856189774724f);  // log2(100 / exp(0.09 g))
    t.lm = fmaxf(fast_log2(obs2) + log2_qlam2, -45.99999998912693f) + t.ls;  // log2(ZDE^2), ZDE = FLT_EPSILON (utils.h:19)
    t.rqm = fast_rcp(1.f + fast_exp2(-t.c * t.lm));
    return t;
}
// mu/p for an end-point error (ex, ey)
__device__ __forceinline__ float obs_ratio(const ObsTerms& t, float ex, float ey, float qia2 /* 0.25 / abs_rf^2 */) {
#pragma clang fp contract(off)
    const float l = fast_log2(fmaxf(fmaf(ex, ex, ey * ey) * qia2, 1.4210854822304103e-14f)) + t.ls;
    const float a = (1.f + fast_exp2(-t.c * l)) * t.rqm;
    return fast_exp2(t.c1 * (l - t.lm)) * a * a;
}

// -log(rigidness) of one frame for the float inputs strict::rigidness takes (vk_strict_model.hpp): the pre-filter of the strict sample pass
// (vk_depth_impl.hpp "the strict sample pass behind an fp32 filter").  |filt_neglog - (-vsm_logf(strict::rigidness))| <= SF_ABS + SF_REL * value is what the filter
// relies on; tests/test_gpu_strict_filter.py measures the difference over the input range and holds it a factor 10 below.
constexpr float SF_REL = 1e-4f, SF_ABS = 1e-4f;
__device__ __forceinline__ float filt_neglog(const ObsTerms& T, float dx1, float dy1, float ox, float oy, float qia2) {
#pragma clang fp contract(off)
    return 0.6931471805599453f * fast_log2(1.f + obs_ratio(T, dx1 - ox, dy1 - oy, qia2));
}

// Two pixels per lane: the same operation sequence on float pairs.  v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 round each half like the scalar
// instruction, the transcendentals, max and med3 have no packed form and run per half: the bits of obs_terms / obs_ratio for either pixel.
typedef float pf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pf2 pk_fma(pf2 a, pf2 b, pf2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ pf2 pk_all(float v) { return pf2{ v, v }; }
struct ObsTerms2 { pf2 c, c1, ls, lm, rqm; };
__device__ __forceinline__ ObsTerms2 obs_terms2(pf2 ox, pf2 oy, float ia2, float log2_qlam2) {
#pragma clang fp contract(off)
    const pf2 obs2 = pk_fma(ox, ox, oy * oy) * ia2;
    const pf2 hs = 0.5f * pf2{ fast_sqrt(obs2.x), fast_sqrt(obs2.y) };
    const pf2 g = { __builtin_amdgcn_fmed3f(hs.x, 2.f, 100.f), __builtin_amdgcn_fmed3f(hs.y, 2.f, 100.f) };
    ObsTerms2 t;
    t.c = pk_fma(pk_all(-0.0022f), g, pk_all(1.0f));
    t.c1 = t.c + 1.f;
    t.ls = pk_fma(pk_all(-0.12984255368000671f), g, pk_all(6.643856189774724f));
    const pf2 lg = pf2{ fast_log2(obs2.x), fast_log2(obs2.y) } + log2_qlam2;
    t.lm = pf2{ fmaxf(lg.x, -45.99999998912693f), fmaxf(lg.y, -45.99999998912693f) } + t.ls;
    const pf2 e = -t.c * t.lm;
    t.rqm = pf2{ fast_rcp(1.f + fast_exp2(e.x)), fast_rcp(1.f + fast_exp2(e.y)) };
    return t;
}
__device__ __forceinline__ pf2 obs_ratio2(const ObsTerms2& t, pf2 ex, pf2 ey, float qia2) {
#pragma clang fp contract(off)
    const pf2 m = pk_fma(ex, ex, ey * ey) * qia2;
    const pf2 l = pf2{ fast_log2(fmaxf(m.x, 1.4210854822304103e-14f)), fast_log2(fmaxf(m.y, 1.4210854822304103e-14f)) } + t.ls;
    const pf2 q = -t.c * l;
    const pf2 a = (1.f + pf2{ fast_exp2(q.x), fast_exp2(q.y) }) * t.rqm;
    const pf2 u = t.c1 * (l - t.lm);
    return pf2{ fast_exp2(u.x), fast_exp2(u.y) } * a * a;
}

// bilinear flow fetch at a position that is known to lie inside [0,w) x [0,h): two 16-byte texel-pair loads (rows yb, yb+1 at
// column xb).  At the last column / row the pair is shifted inwards and the weight pinned to 1: the same value as clamping.
typedef float vf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 bilinear2_inside(const float2* __restrict__ img, int w, int h, float x, float y) {
#pragma clang fp contract(off)
    const float fx = floorf(x), fy = floorf(y);
    const int x0 = (int)fx, y0 = (int)fy;
    const int xb = min(x0, w - 2), yb = min(y0, h - 2);
    const float a = x0 > xb ? 1.f : x - fx, b = y0 > yb ? 1.f : y - fy;
    const char* base = reinterpret_cast<const char*>(img);
    const unsigned off = (unsigned)(__mul24(yb, w) + xb) * 8u;  // v_mad_i32_i24: full rate (a 32-bit integer multiply issues at a quarter of it); rows and widths are far below 2^23
    const TexPair r0 = *reinterpret_cast<const TexPair*>(base + off);
    const TexPair r1 = *reinterpret_cast<const TexPair*>(base + off + (unsigned)w * 8u);
    const float tx = fmaf(a, r0.bx - r0.ax, r0.ax), ty = fmaf(a, r0.by - r0.ay, r0.ay);
    const float ux = fmaf(a, r1.bx - r1.ax, r1.ax), uy = fmaf(a, r1.by - r1.ay, r1.ay);
    return make_float2(fmaf(b, ux - tx, tx), fmaf(b, uy - ty, ty));
}
// homogeneous pixel of (x, y, d) under one projective map (PoseBlock::cumM / cumT)
struct H3 { float x, y, z; };
__device__ __forceinline__ H3 hom_dir(const float* __restrict__ M, float x, float y) {
#pragma clang fp contract(off)
    return { fmaf(M[0], x, fmaf(M[1], y, M[2])), fmaf(M[3], x, fmaf(M[4], y, M[5])), fmaf(M[6], x, fmaf(M[7], y, M[8])) };
}

// ---- geometry (optimize_depth.cu:54-81) -------------------------------------------------
struct P3 { float x, y, z; };
// Geometry and bilinear weights are evaluated with the reference's operation order, true
// division and NO fma contraction, so pixel positions, in-bounds decisions and gather weights are
// bit-identical to an un-fused fp32 evaluation (the oracle); only the transcendental part of the
// residual model differs between the two.
__device__ __forceinline__ P3 backproject(const PoseBlock* P, float px, float py, float d) {
#pragma clang fp contract(off)
    return { (P->K4i[0] * px + P->K4i[1]) * d, (P->K4i[2] * py + P->K4i[3]) * d, d };
}
__device__ __forceinline__ void project(const PoseBlock* P, P3 o, float& px, float& py) {
#pragma clang fp contract(off)
    px = (P->K4[0] * o.x + P->K4[1] * o.z) / o.z;
    py = (P->K4[2] * o.y + P->K4[3] * o.z) / o.z;
}
__device__ __forceinline__ P3 transform(const float* R, const float* t, P3 o) {
#pragma clang fp contract(off)
    return { o.x * R[0] + o.y * R[1] + o.z * R[2] + t[0],
             o.x * R[3] + o.y * R[4] + o.z * R[5] + t[1],
             o.x * R[6] + o.y * R[7] + o.z * R[8] + t[2] };
}

// ---- bilinear fetch, clamp-to-edge per layer, exact fp32 weights ------------------------
struct BilIdx { int i00, i10, i01, i11; float a, b; };
__device__ __forceinline__ BilIdx bil_index(float x, float y, int w, int h) {
#pragma clang fp contract(off)
    float fx = floorf(x), fy = floorf(y);
    BilIdx r;
    r.a = x - fx; r.b = y - fy;
    int x0 = (int)fx, y0 = (int)fy;
    int x1 = min(max(x0 + 1, 0), w - 1), y1 = min(max(y0 + 1, 0), h - 1);
    x0 = min(max(x0, 0), w - 1); y0 = min(max(y0, 0), h - 1);
    const int r0 = __mul24(y0, w), r1 = __mul24(y1, w);
    r.i00 = r0 + x0; r.i10 = r0 + x1; r.i01 = r1 + x0; r.i11 = r1 + x1;
    return r;
}
// Two horizontally adjacent flow texels in ONE 16-byte access (8-byte aligned; gfx950 global loads
// are alignment-free).  The per-pixel kernels are bound by the number of distinct cache lines their
// uncoalesced gathers touch per wave instruction, not by bytes: a bilinear fetch is 2 line accesses
// instead of 4.  Border handling picks the clamped texels out of the pair, so the values are the same
// as four clamped single-texel fetches.
__device__ __forceinline__ float2 bilinear2(const float2* __restrict__ img, int w, int h, float x, float y) {
#pragma clang fp contract(off)
    float fx = floorf(x), fy = floorf(y);
    const float a = x - fx, b = y - fy;
    int x0 = (int)fx, y0 = (int)fy;
    const int x1 = min(max(x0 + 1, 0), w - 1), y1 = min(max(y0 + 1, 0), h - 1);
    x0 = min(max(x0, 0), w - 1); y0 = min(max(y0, 0), h - 1);
    float2 t00, t10, t01, t11;
    if (w >= 2) {
        const int xb = min(x0, w - 2);
        // unsigned 32-bit byte offsets from the (wave-uniform) layer base: scalar base + vector offset addressing
        const char* base = reinterpret_cast<const char*>(img);
        const TexPair r0 = *reinterpret_cast<const TexPair*>(base + (unsigned)(__mul24(y0, w) + xb) * 8u);
        const TexPair r1 = *reinterpret_cast<const TexPair*>(base + (unsigned)(__mul24(y1, w) + xb) * 8u);
        const bool lo0 = x0 == xb, lo1 = x1 == xb;
        t00 = lo0 ? make_float2(r0.ax, r0.ay) : make_float2(r0.bx, r0.by);
        t10 = lo1 ? make_float2(r0.ax, r0.ay) : make_float2(r0.bx, r0.by);
        t01 = lo0 ? make_float2(r1.ax, r1.ay) : make_float2(r1.bx, r1.by);
        t11 = lo1 ? make_float2(r1.ax, r1.ay) : make_float2(r1.bx, r1.by);
    } else {
        t00 = img[y0 * w + x0]; t10 = img[y0 * w + x1]; t01 = img[y1 * w + x0]; t11 = img[y1 * w + x1];
    }
    float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
    return make_float2(w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x,
                       w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y);
}
__device__ __forceinline__ float bilinear1(const float* __restrict__ img, int w, int h, float x, float y) {
#pragma clang fp contract(off)
    BilIdx k = bil_index(x, y, w, h);
    float w00 = (1.f - k.a) * (1.f - k.b), w10 = k.a * (1.f - k.b), w01 = (1.f - k.a) * k.b, w11 = k.a * k.b;
    return w00 * img[k.i00] + w10 * img[k.i10] + w01 * img[k.i01] + w11 * img[k.i11];
}

// ---- XCD-aware workgroup order ---------------------------------------------------------------
// MI355X dispatches consecutive workgroup ids round-robin over its 8 XCDs, each with a private 4 MB L2.
// The per-pixel kernels gather from the N flow layers around their own pixel: with the default order every
// XCD touches every part of every layer (12 MB at 640x480 N=5, 100+ MB at 1080p) and the gathers miss L2.
// This remap gives XCD k the k-th contiguous eighth of the (row-major) tile list, i.e. a band of image rows,
// so an L2 only ever sees one band (+ halo) of each layer.  Returns the tile this workgroup should process.
__device__ __forceinline__ int xcd_band_tile(int bid, int nb) {
    constexpr int NXCD = 8;
    const int per = nb / NXCD;
    if (bid >= per * NXCD) return bid;  // remainder tiles keep their place
    return (bid % NXCD) * per + bid / NXCD;
}

// ---- wave64 / block reductions ----------------------------------------------------------
// Sum over the 64 lanes, result in every lane.  Row (16-lane) butterflies run on the VALU through
// DPP modifiers (quad_perm / row_half_mirror / row_mirror); the four row sums are then combined
// through v_readlane.  __shfl_xor would go through the LDS crossbar (ds_bpermute): with 28 sums x 16
// waves per mode-finding iteration that alone cost several microseconds per iteration.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror  -> every lane holds its 16-lane row sum
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

// Transposing wave reduction: NV (<= P = 32 or 8) per-lane values -> lane L returns the 64-lane total of value
// wave_slot<P>(L).  Stage with lane distance d halves the number of values a lane holds: the lane with bit d clear
// keeps the lower half, its partner the upper half, each adds what the other one sends (2 selects + 1 DPP add per kept
// value).  ~3.4 instructions per value instead of the 11 of an all-lanes wave_sum per value; with 28 values x 16 waves
// per refit iteration on ONE compute unit that is the difference between 310 and ~110 issue slots per wave.
template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ float dpp_get(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, BANK, false));
}
template <int D>
__device__ __forceinline__ float lane_xor(float v) {  // value of lane (L ^ D)
    if (D == 1) return dpp_mov<0xB1>(v);
    if (D == 2) return dpp_mov<0x4E>(v);
    // row_ror:n hands lane i the value of lane (i - n) mod 16: banks 0,2 (bit 2 clear) read i+4 via ror:12, banks 1,3 read i-4 via ror:4
    if (D == 4) { float t = dpp_get<0x12C, 0x5>(v, v); return dpp_get<0x124, 0xa>(t, v); }
    if (D == 8) return dpp_mov<0x128>(v);                                                    // row_ror:8
    return __shfl_xor(v, D, 64);
}
template <int P> __device__ __forceinline__ int wave_slot(int lane) {
    int idx = 0;
#pragma unroll
    for (int s = 0, half = P / 2; half >= 1; s++, half >>= 1) idx += ((lane >> s) & 1) * half;
    return idx;
}
template <int P, int D, int HALF>
__device__ __forceinline__ void wave_transpose_stage(float (&v)[P], int lane) {
    const bool up = (lane & D) != 0;
#pragma unroll
    for (int j = 0; j < HALF; j++) {
        const float keep = up ? v[j + HALF] : v[j], send = up ? v[j] : v[j + HALF];
        v[j] = keep + lane_xor<D>(send);
    }
}
template <int P>
__device__ __forceinline__ float wave_reduce_transpose(float (&v)[P]) {
    static_assert(P == 32 || P == 8, "padded value count");
    const int lane = threadIdx.x & 63;
    if (P == 32) {
        wave_transpose_stage<P, 1, 16>(v, lane); wave_transpose_stage<P, 2, 8>(v, lane); wave_transpose_stage<P, 4, 4>(v, lane);
        wave_transpose_stage<P, 8, 2>(v, lane); wave_transpose_stage<P, 16, 1>(v, lane);
        return v[0] + lane_xor<32>(v[0]);
    } else {
        wave_transpose_stage<P, 1, 4>(v, lane); wave_transpose_stage<P, 2, 2>(v, lane); wave_transpose_stage<P, 4, 1>(v, lane);
        float t = v[0];
        t += lane_xor<8>(t); t += lane_xor<16>(t); t += lane_xor<32>(t);
        return t;
    }
}

}  // namespace vk
