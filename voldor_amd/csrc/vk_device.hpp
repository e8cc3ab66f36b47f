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
struct FiskParams { float c, inv_s; };
__device__ __forceinline__ FiskParams fisk_params(float fmag) {
    float g = fminf(fmaxf(fmag * 0.5f, 2.f), 100.f);  // residual_model.h:16, :22
    FiskParams p;
    p.c = 1.0f - 0.0022f * g;                          // FISK_B1 + FISK_B2*g
    // s = 0.01*exp(0.09 g)  ->  1/s = 100 * exp2(-0.09*log2(e)*g)
    p.inv_s = 100.f * exp2f(-0.12984255368000671f * g);
    return p;
}
// returns t = mu/p given the error magnitude `e` and the strictness magnitude `m` (both
// already divided by abs_rf).  All quantities stay finite for e,m >= 0.
__device__ __forceinline__ float fisk_ratio(float e, float m, FiskParams fp) {
    const float eps = 1.1920929e-07f;  // ZDE = FLT_EPSILON (utils.h:19)
    float xe = fmaxf(e * 0.5f, eps), xm = fmaxf(m * 0.5f, eps);
    float r = xe * xe * fp.inv_s, rm = xm * xm * fp.inv_s;
    float q = exp2f(-fp.c * __log2f(r)), qm = exp2f(-fp.c * __log2f(rm));
    // mu/p = (qm * r * (1+q)^2) / (q * rm * (1+qm)^2); evaluate as a product of ratios to
    // keep intermediates in range (q can reach ~1e12 at e -> 0).
    float a = (1.f + q) / (1.f + qm);
    return (qm / q) * (r / rm) * a * a;
}
__device__ __forceinline__ float rigidness_from_flows(float dx1, float dy1, float dx2, float dy2,
                                                      float lambda, float inv_arf) {
    float obs = sqrtf(dx2 * dx2 + dy2 * dy2) * inv_arf;
    float ex = dx1 - dx2, ey = dy1 - dy2;
    float diff = sqrtf(ex * ex + ey * ey) * inv_arf;
    float t = fisk_ratio(diff, lambda * obs, fisk_params(obs));
    return 1.f / (1.f + t);
}
// -log(rigidness): cost contribution of one frame (residual_model.h:45-49)
__device__ __forceinline__ float neglog_rigidness_from_flows(float dx1, float dy1, float dx2, float dy2,
                                                             float lambda, float inv_arf) {
    float obs = sqrtf(dx2 * dx2 + dy2 * dy2) * inv_arf;
    float ex = dx1 - dx2, ey = dy1 - dy2;
    float diff = sqrtf(ex * ex + ey * ey) * inv_arf;
    float t = fisk_ratio(diff, lambda * obs, fisk_params(obs));
    return __logf(1.f + t);
}
// depth-prior variant on disparities (residual_model.h:51-68)
__device__ __forceinline__ float depth_ratio(float d1, float d2, float basefocal, float omega, float inv_arf) {
    float disp1 = (basefocal / d1) * inv_arf, disp2 = (basefocal / d2) * inv_arf;
    return fisk_ratio(fabsf(disp1 - disp2), omega * disp2, fisk_params(disp2));
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
    r.i00 = y0 * w + x0; r.i10 = y0 * w + x1; r.i01 = y1 * w + x0; r.i11 = y1 * w + x1;
    return r;
}
__device__ __forceinline__ float2 bilinear2(const float2* __restrict__ img, int w, int h, float x, float y) {
#pragma clang fp contract(off)
    BilIdx k = bil_index(x, y, w, h);
    float2 t00 = img[k.i00], t10 = img[k.i10], t01 = img[k.i01], t11 = img[k.i11];
    float w00 = (1.f - k.a) * (1.f - k.b), w10 = k.a * (1.f - k.b), w01 = (1.f - k.a) * k.b, w11 = k.a * k.b;
    return make_float2(w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x,
                       w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y);
}
__device__ __forceinline__ float bilinear1(const float* __restrict__ img, int w, int h, float x, float y) {
#pragma clang fp contract(off)
    BilIdx k = bil_index(x, y, w, h);
    float w00 = (1.f - k.a) * (1.f - k.b), w10 = k.a * (1.f - k.b), w01 = (1.f - k.a) * k.b, w11 = k.a * k.b;
    return w00 * img[k.i00] + w10 * img[k.i10] + w01 * img[k.i01] + w11 * img[k.i11];
}

// ---- wave64 / block reductions ----------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace vk
