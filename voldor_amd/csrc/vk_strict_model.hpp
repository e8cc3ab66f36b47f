// voldor_amd/csrc/vk_strict_model.hpp -- the residual model in the REFERENCE's operation order on the software
// transcendentals of vk_strict_math.h ("strict math" mode).
//
// The fast kernels use an algebraically re-derived form on v_log_f32 / v_exp_f32 (vk_device.hpp: 5 transcendental
// instructions per rigidness).  Here the formulas of gpu-kernels/residual_model.h:15-68 are evaluated as written there --
// fun_fmag_c, fun_fmag_scale, fisk_dist_pdf, fun_rigidness, fun_depth_rigidness, one rounding per operation, no fma -- with
// vsm_expf / vsm_powf / vsm_logf in place of the library calls.  The oracle in strict mode (oracle/orc_math.h) evaluates the
// same expressions with the same functions, so both produce the same bits; host-compiled, these functions are checked against
// it on the CPU (tests/test_strict_host.py), on the GPU they are checked against their own host build (tests/test_gpu_strict.py).
#pragma once
#include <hip/hip_runtime.h>
#include "vk_strict_math.h"

namespace vk { namespace strict {

#define VK_SHD __host__ __device__ __forceinline__

// residual_model.h:6-13: EST_RF 0.5 is a DOUBLE literal there (fmag * EST_RF widens), the rest are floats
VK_SHD float clamp_fmag(float fmag) {
#pragma clang fp contract(off)
    return fminf(fmaxf((float)((double)fmag * 0.5), 2.f), 100.f);  // :16 / :22
}
VK_SHD float fmag_c(float fmag) {
#pragma clang fp contract(off)
    return 1.0f + -0.0022f * clamp_fmag(fmag);  // FISK_B1 + FISK_B2 * fmag (:17)
}
VK_SHD float fmag_scale(float fmag) {
#pragma clang fp contract(off)
    return 0.01f * vsm_expf(0.09f * clamp_fmag(fmag));  // FISK_A1 * expf(FISK_A2 * fmag) (:23)
}
// vsm_powf(x, y) = (float)exp(y log x) behind its special cases (vk_strict_math.h): two powers of ONE base share the logarithm.  Same operations on
// the same values as two calls of vsm_powf -- the same bits -- without relying on the compiler to find the common subexpression.
VK_SHD void powf2_same_base(float x, float y1, float y2, float& o1, float& o2) {
#pragma clang fp contract(off)
    const double xd = (double)x;
    if (xd != xd || xd < 0.0 || xd == 0.0 || xd == 1.0) { o1 = vsm_powf(x, y1); o2 = vsm_powf(x, y2); return; }  // special bases: the plain calls
    const double l = vsm_log(xd);
    auto one = [&](float y) -> float {
        const double yd = (double)y;
        if (yd == 0.0) return 1.0f;
        if (yd != yd) return (float)vsm_nan();
        if (l == 0.0) return 1.0f;
        return (float)vsm_exp(yd * l);
    };
    o1 = one(y1); o2 = one(y2);
}
// ---- the same FLOAT with fewer software transcendentals (round 6) ---------------------------------------------------------------------------------------
// vsm_powf rounds a double that is within ~1e-13 of the true power to float.  Any other double that close to the true value rounds to the same float unless
// a float rounding boundary lies between the two; the boundaries are the doubles whose 29 low mantissa bits (the ones a float drops) read 0x10000000.  So a
// cheaper double may stand in for the plain one wherever it keeps 2^17 units of its last place (a relative 1.4e-11) away from every boundary and is a normal
// float -- all but one value in 2000; elsewhere the plain function runs.  Same bits, decided per value, on the host and on gfx950 alike.  The oracle keeps
// calling vsm_powf (oracle/orc_math.h): every strict comparison with it checks the shortcut too, tests/test_strict_host.py holds it against the plain
// call over whole binades and measures the distance between the two doubles.  (Built the same way and measured without gain, not kept: r^(-c-1) from r^(-c) / r --
// one exponential less, a division and a guard more -- and fused-multiply-add forms of vsm_exp / vsm_log: scripts/experiments/r06_quick_forms.patch.txt.)
VK_SHD bool rounds_alike(double g) {
    const unsigned long long u = vsm_bits(g);
    const unsigned e = (unsigned)(u >> 52) & 0x7ffu, d = (unsigned)(u & 0x1fffffffull);
    const unsigned dist = d > 0x10000000u ? d - 0x10000000u : 0x10000000u - d;
    return e >= 1023u - 126u && e <= 1023u + 126u && dist > (1u << 17);
}
// (float)vsm_pow(x, -2) for the x = 1 + r^-c >= 1 of fisk_pdf without the logarithm and the exponential: x * x is exact in double (48 bits) and the division
// rounds once, so y is within 2^-53 of x^-2; vsm_exp(-2 vsm_log(x)) is within 1e-13 of it for x <= 1e18 (|t| <= 83: the rounding of t = -2 l dominates).
// Measured: the two differ by <= 65 units of the last place.
VK_SHD float pow_m2(float x) {
#pragma clang fp contract(off)
    if (x >= 1.0f && x <= 1e18f) {  // (false for a NaN)
        const double xd = (double)x;
        const double y = 1.0 / (xd * xd);
        if (rounds_alike(y)) return (float)y;
    }
    return vsm_powf(x, -2.f);
}
VK_SHD float fisk_pdf(float x, float c, float scale) {  // :28-31
#pragma clang fp contract(off)
    x = fmaxf((float)((double)x * 0.5), 1.1920929e-07f);
    const float r = (x * x) / scale;
    float p1, p2;
    powf2_same_base(r, -c - 1.f, -c, p1, p2);
    return (c * p1 * pow_m2(1.f + p2)) / scale;
}
// What fun_rigidness and fun_depth_rigidness share once the magnitude of the observation (flow magnitude / disparity), the magnitude of the difference and the
// strictness (lambda / omega) are known -- so that a wave whose lanes hold frames AND depth priors runs the software transcendentals once (strict_term, vk_depth_impl.hpp).
// rig_core_plain is the model as residual_model.h writes it, call by call.
VK_SHD float rig_core_plain(float mag, float diff, float strictness) {
#pragma clang fp contract(off)
    const float c = fmag_c(mag), s = fmag_scale(mag);
    const float p = fisk_pdf(diff, c, s), mu = fisk_pdf(strictness * mag, c, s);
    return p / (p + mu);
}
// The same operations on the same values in ONE STRAIGHT LINE (round 6).  Called one after the other, the seven software transcendentals of a residual are seven
// dependent chains behind entry tests -- branches the scheduler cannot move instructions across -- and a wave advances at the latency of one fp64 operation per
// instruction (measured: the strict passes do not get faster with fewer waves, and get slower with more, smaller ones).  Here the entry tests of all of them are made
// first (bases positive, finite, not 1; exponents finite, not 0; every product y log x inside vsm_exp's range: what holds for any pixel that is not degenerate), then the
// two logarithms, the four exponentials and the two reciprocal squares stand side by side as vsm_log_core / vsm_exp_core -- the bodies vsm_log / vsm_exp run behind
// their tests -- and the scheduler interleaves them.  Where a test fails the model runs call by call (rig_core_plain's path); a reciprocal square too close to a float
// rounding boundary takes its plain call alone.  Same bits: tests/test_strict_host.py holds rig_core against rig_core_plain (and both against the oracle).
VK_SHD float rig_core(float mag, float diff, float strictness) {
#pragma clang fp contract(off)
    const float g = clamp_fmag(mag);  // in [2, 100] whatever mag is (a NaN clamps to 2): 0.09 g is inside vsm_expf's plain range
    const float c = 1.0f + -0.0022f * g;
    const float s = 0.01f * (float)vsm_exp_core((double)(0.09f * g));
    const float xm = strictness * mag;
    const float x1 = fmaxf((float)((double)diff * 0.5), 1.1920929e-07f), x2 = fmaxf((float)((double)xm * 0.5), 1.1920929e-07f);
    const float r1 = (x1 * x1) / s, r2 = (x2 * x2) / s;
    const double d1 = (double)r1, d2 = (double)r2, ya = (double)(-c - 1.f), yb = (double)(-c);
    bool ok = d1 > 0.0 && d1 < 3.5e38 && d1 != 1.0 && d2 > 0.0 && d2 < 3.5e38 && d2 != 1.0 && ya != 0.0 && yb != 0.0;  // (false for a NaN base; c is finite)
    const double l1 = vsm_log_core(ok ? d1 : 2.0, 0), l2 = vsm_log_core(ok ? d2 : 2.0, 0);
    double t1a = ya * l1, t1b = yb * l1, t2a = ya * l2, t2b = yb * l2;
    ok = ok && t1a >= -745.2 && t1a <= 709.782712893384 && t1b >= -745.2 && t1b <= 709.782712893384 && t2a >= -745.2 && t2a <= 709.782712893384 && t2b >= -745.2 && t2b <= 709.782712893384;
    if (!ok) { t1a = 0.0; t1b = 0.0; t2a = 0.0; t2b = 0.0; }
    const float p1a = (float)vsm_exp_core(t1a), p1b = (float)vsm_exp_core(t1b), p2a = (float)vsm_exp_core(t2a), p2b = (float)vsm_exp_core(t2b);
    const float q1 = 1.f + p1b, q2 = 1.f + p2b;
    const double w1 = 1.0 / ((double)q1 * (double)q1), w2 = 1.0 / ((double)q2 * (double)q2);
    if (!ok) return rig_core_plain(mag, diff, strictness);
    float m1 = (float)w1, m2 = (float)w2;
    if (!(q1 >= 1.0f && q1 <= 1e18f && rounds_alike(w1))) m1 = vsm_powf(q1, -2.f);  // pow_m2's own fallback
    if (!(q2 >= 1.0f && q2 <= 1e18f && rounds_alike(w2))) m2 = vsm_powf(q2, -2.f);
    const float p = (c * p1a * m1) / s, mu = (c * p2a * m2) / s;
    return p / (p + mu);
}
VK_SHD float rigidness(float dx1, float dy1, float dx2, float dy2, float lambda, float abs_rf) {  // fun_rigidness :34-42
#pragma clang fp contract(off)
    const float obs_fmag = sqrtf(dx2 * dx2 + dy2 * dy2) / abs_rf;
    const float ex = dx1 - dx2, ey = dy1 - dy2;
    const float diff_fmag = sqrtf(ex * ex + ey * ey) / abs_rf;
    return rig_core(obs_fmag, diff_fmag, lambda);
}
// fun_rigidness with its observation-only part split off: c, s and mu = fisk_pdf(lambda |obs|) depend on the OBSERVED flow alone, and frame 0
// observes at the pixel itself -- every depth hypothesis of a pixel shares them (ten random samples: k_cost_rand_q_strict).  The same
// operations on the same values in the same order as rigidness() above: the same bits.
struct RigObs { float c, s, mu, dx2, dy2; };
VK_SHD RigObs rigidness_obs(float dx2, float dy2, float lambda, float abs_rf) {
#pragma clang fp contract(off)
    const float obs_fmag = sqrtf(dx2 * dx2 + dy2 * dy2) / abs_rf;
    RigObs o;
    o.c = fmag_c(obs_fmag); o.s = fmag_scale(obs_fmag);
    o.mu = fisk_pdf(lambda * obs_fmag, o.c, o.s);
    o.dx2 = dx2; o.dy2 = dy2;
    return o;
}
VK_SHD float rigidness_with(const RigObs& o, float dx1, float dy1, float abs_rf) {
#pragma clang fp contract(off)
    const float ex = dx1 - o.dx2, ey = dy1 - o.dy2;
    const float diff_fmag = sqrtf(ex * ex + ey * ey) / abs_rf;
    const float p = fisk_pdf(diff_fmag, o.c, o.s);
    return p / (p + o.mu);
}
VK_SHD float depth_rigidness(float d1, float d2, float basefocal, float omega, float abs_rf) {  // fun_depth_rigidness :51-61
#pragma clang fp contract(off)
    const float disp1 = (basefocal / d1) / abs_rf, disp2 = (basefocal / d2) / abs_rf;
    return rig_core(disp2, fabsf(disp1 - disp2), omega);
}
// fun_cost / fun_depth_cost (:45-49, :64-68): io_cost -= weight * logf(rigidness)
VK_SHD float cost_acc(float cost_sum, float weight, float rig) {
#pragma clang fp contract(off)
    return cost_sum - weight * vsm_logf(rig);
}

#undef VK_SHD
}}  // namespace vk::strict
