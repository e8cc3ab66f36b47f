/* voldor_amd/csrc/vk_strict_math.h -- software transcendentals with ONE rounding sequence on every target.
 *
 * Purpose ("strict math" mode).  The reference evaluates its residual model with CUDA's powf / expf /
 * logf (gpu-kernels/residual_model.h:15-68), the CPU oracle with glibc's, the fast HIP path with v_log_f32 / v_exp_f32: three
 * roundings of the same formulas.  The depth search compares costs with `<` (optimize_depth.cu:201-207), so a last-bit
 * difference flips near-ties and whole windows drift apart to the estimator's sampling noise -- which makes "the HIP window
 * reproduces the reference window" untestable.  In strict mode both sides (kernels in vk_strict.hip / vk_depth.hip and the
 * oracle with orc_set_strict_math(1)) call THESE functions instead: only IEEE-754 + - * / sqrt on double, comparisons and integer
 * bit operations, no fused multiply-add, no library call, no table -- so gcc on the host and hipcc for gfx950 produce the same
 * bits, and a whole window can be compared bit for bit (tests/test_gpu_strict.py).
 *
 * Accuracy: every function is evaluated in double to ~1e-15 relative and rounded once to float, i.e. within half an ulp + 1e-8
 * of the correctly rounded float result -- the same class as glibc's float functions.  Plain C so that the oracle (gcc -std=gnu11)
 * can include it; nothing here is derived from the reference tree.
 */
#ifndef VK_STRICT_MATH_H
#define VK_STRICT_MATH_H

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VSM_FN __host__ __device__ static inline
#else
#define VSM_FN static inline
#endif
#if defined(__clang__)
#define VSM_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define VSM_NO_CONTRACT /* gcc: the oracle is compiled with -ffp-contract=off */
#endif

VSM_FN unsigned long long vsm_bits(double x) { unsigned long long u; __builtin_memcpy(&u, &x, 8); return u; }
VSM_FN double vsm_from_bits(unsigned long long u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
VSM_FN double vsm_inf(void) { return vsm_from_bits(0x7ff0000000000000ull); }
VSM_FN double vsm_nan(void) { return vsm_from_bits(0x7ff8000000000000ull); }
VSM_FN double vsm_abs(double x) { return vsm_from_bits(vsm_bits(x) & 0x7fffffffffffffffull); }
/* 2^k for -1022 <= k <= 1023 */
VSM_FN double vsm_pow2i(int k) { return vsm_from_bits((unsigned long long)(k + 1023) << 52); }

#define VSM_LN2_HI 6.93147180369123816490e-01 /* the top 33 bits of ln 2: k * LN2_HI is exact for |k| < 2^20 */
#define VSM_LN2_LO 1.90821492927058770002e-10
#define VSM_INV_LN2 1.44269504088896338700e+00
#define VSM_PI 3.14159265358979311600e+00
#define VSM_PIO2_HI 1.57079632673412561417e+00 /* the top 33 bits of pi/2 */
#define VSM_PIO2_LO 6.07710050650619224932e-11

/* e^x for -745.2 <= x <= 709.78 (not a NaN): vsm_exp behind its entry tests, one straight line (round 6: callers that have made the tests themselves -- the residual
 * model, vk_strict_model.hpp -- put several of these side by side so that the instruction scheduler can interleave the dependent chains) */
VSM_FN double vsm_exp_core(double x) {
    VSM_NO_CONTRACT
    const double kf = x * VSM_INV_LN2;
    const int k = (int)(kf + (kf >= 0.0 ? 0.5 : -0.5)); /* nearest integer (conversion truncates) */
    const double kd = (double)k;
    const double r = (x - kd * VSM_LN2_HI) - kd * VSM_LN2_LO; /* |r| <= 0.347 */
    /* Taylor to r^14 (truncation 1e-19 relative), Horner */
    double p = 1.0 / 87178291200.0;
    p = p * r + 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    /* scale by 2^k in two normal factors (k down to -1075 / up to 1024) */
    const int k1 = k / 2, k2 = k - k1;
    return (p * vsm_pow2i(k1)) * vsm_pow2i(k2);
}
/* e^x */
VSM_FN double vsm_exp(double x) {
    VSM_NO_CONTRACT
    if (x != x) return x;
    if (x > 709.782712893384) return vsm_inf();
    if (x < -745.2) return 0.0;
    return vsm_exp_core(x);
}

/* log x for a normal, finite x > 0, plus e0 ln 2: vsm_log behind its entry tests, one straight line */
VSM_FN double vsm_log_core(double x, int e0) {
    VSM_NO_CONTRACT
    const unsigned long long u = vsm_bits(x);
    int e = e0 + (int)((u >> 52) & 0x7ffull) - 1023;
    double m = vsm_from_bits((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull); /* [1, 2) */
    const int upper = m > 1.4142135623730951;                                          /* -> (0.707, 1.414] */
    m = upper ? m * 0.5 : m;
    e += upper;
    const double s = (m - 1.0) / (m + 1.0), s2 = s * s;                               /* |s| <= 0.1716 */
    /* log m = 2 atanh s = 2 s (1 + s^2/3 + s^4/5 + ... + s^22/23) */
    double p = 1.0 / 23.0;
    p = p * s2 + 1.0 / 21.0;
    p = p * s2 + 1.0 / 19.0;
    p = p * s2 + 1.0 / 17.0;
    p = p * s2 + 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0;
    p = p * s2 + 1.0 / 9.0;
    p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;
    p = p * s2 + 1.0 / 3.0;
    p = p * s2 + 1.0;
    const double ed = (double)e;
    return ed * VSM_LN2_HI + (ed * VSM_LN2_LO + 2.0 * s * p);
}
/* natural logarithm */
VSM_FN double vsm_log(double x) {
    VSM_NO_CONTRACT
    if (x != x) return x;
    if (x < 0.0) return vsm_nan();
    if (x == 0.0) return -vsm_inf();
    if (x == vsm_inf()) return x;
    int e = 0;
    if (x < 2.2250738585072014e-308) { x = x * 18014398509481984.0; e = -54; } /* subnormal: scale by 2^54 */
    return vsm_log_core(x, e);
}

/* x^y for x >= 0 (the residual model and AP3P never raise a negative base; a negative base returns NaN) */
VSM_FN double vsm_pow(double x, double y) {
    VSM_NO_CONTRACT
    if (y == 0.0 || x == 1.0) return 1.0;
    if (x != x || y != y) return vsm_nan();
    if (x < 0.0) return vsm_nan();
    if (x == 0.0) return y > 0.0 ? 0.0 : vsm_inf();
    const double l = vsm_log(x);
    if (l == 0.0) return 1.0;
    const double t = y * l; /* inf * 0 cannot occur: l != 0 and y != 0 */
    return vsm_exp(t);
}

/* atan2, result in (-pi, pi] */
VSM_FN double vsm_atan2(double y, double x) {
    VSM_NO_CONTRACT
    if (x != x || y != y) return vsm_nan();
    const double ax = vsm_abs(x), ay = vsm_abs(y);
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else {
        const double inf = vsm_inf();
        double z;
        if (ax == inf && ay == inf) z = 1.0;
        else z = ay > ax ? ax / ay : ay / ax; /* [0, 1] */
        double a0 = 0.0, sg = 1.0, a1 = 0.0;
        if (z > 0.41421356237309503) { a0 = 0.78539816339744828; z = (z - 1.0) / (z + 1.0); } /* atan z = pi/4 + atan((z-1)/(z+1)) */
        if (z < 0.0) { sg = -1.0; z = -z; }
        if (z > 0.19891236737965800) { a1 = 0.39269908169872414; z = (z - 0.41421356237309503) / (1.0 + z * 0.41421356237309503); }
        const double z2 = z * z; /* |z| <= 0.1990: series to z^23 */
        double p = -1.0 / 23.0;
        p = p * z2 + 1.0 / 21.0;
        p = p * z2 - 1.0 / 19.0;
        p = p * z2 + 1.0 / 17.0;
        p = p * z2 - 1.0 / 15.0;
        p = p * z2 + 1.0 / 13.0;
        p = p * z2 - 1.0 / 11.0;
        p = p * z2 + 1.0 / 9.0;
        p = p * z2 - 1.0 / 7.0;
        p = p * z2 + 1.0 / 5.0;
        p = p * z2 - 1.0 / 3.0;
        p = p * z2 + 1.0;
        a = a0 + sg * (a1 + z * p);
        if (ay > ax) a = 1.5707963267948966 - a;
    }
    if (x < 0.0 || (x == 0.0 && (vsm_bits(x) >> 63))) a = VSM_PI - a;
    if (y < 0.0 || (y == 0.0 && (vsm_bits(y) >> 63))) a = -a;
    return a;
}

/* sin and cos of r with |r| <= pi/4 + reduction slack */
VSM_FN double vsm_sin_kernel(double r) {
    VSM_NO_CONTRACT
    const double r2 = r * r;
    double p = -1.0 / 121645100408832000.0;       /* r^19 */
    p = p * r2 + 1.0 / 355687428096000.0;          /* r^17 */
    p = p * r2 - 1.0 / 1307674368000.0;            /* r^15 */
    p = p * r2 + 1.0 / 6227020800.0;               /* r^13 */
    p = p * r2 - 1.0 / 39916800.0;                 /* r^11 */
    p = p * r2 + 1.0 / 362880.0;                   /* r^9 */
    p = p * r2 - 1.0 / 5040.0;                     /* r^7 */
    p = p * r2 + 1.0 / 120.0;                      /* r^5 */
    p = p * r2 - 1.0 / 6.0;                        /* r^3 */
    p = p * r2 + 1.0;
    return r * p;
}
VSM_FN double vsm_cos_kernel(double r) {
    VSM_NO_CONTRACT
    const double r2 = r * r;
    double p = 1.0 / 2432902008176640000.0;        /* r^20 */
    p = p * r2 - 1.0 / 6402373705728000.0;         /* r^18 */
    p = p * r2 + 1.0 / 20922789888000.0;           /* r^16 */
    p = p * r2 - 1.0 / 87178291200.0;              /* r^14 */
    p = p * r2 + 1.0 / 479001600.0;                /* r^12 */
    p = p * r2 - 1.0 / 3628800.0;                  /* r^10 */
    p = p * r2 + 1.0 / 40320.0;                    /* r^8 */
    p = p * r2 - 1.0 / 720.0;                      /* r^6 */
    p = p * r2 + 1.0 / 24.0;                       /* r^4 */
    p = p * r2 - 0.5;                              /* r^2 */
    p = p * r2 + 1.0;
    return p;
}
/* sin / cos for |x| < 2^20 (rotation angles; accuracy degrades gracefully beyond, NaN for non-finite input) */
VSM_FN void vsm_sincos(double x, double* s, double* c) {
    VSM_NO_CONTRACT
    if (x != x || vsm_abs(x) == vsm_inf()) { *s = vsm_nan(); *c = vsm_nan(); return; }
    if (vsm_abs(x) > 1048576.0) { /* outside the intended domain: fold with a plain remainder so the result stays bounded */
        const double q = x / (2.0 * VSM_PI);
        const double qi = (double)(long long)q;
        x = x - qi * (2.0 * VSM_PI);
    }
    const double kf = x * 0.63661977236758138; /* 2/pi */
    const int k = (int)(kf + (kf >= 0.0 ? 0.5 : -0.5));
    const double kd = (double)k;
    const double r = (x - kd * VSM_PIO2_HI) - kd * VSM_PIO2_LO;
    const double sr = vsm_sin_kernel(r), cr = vsm_cos_kernel(r);
    switch (k & 3) {
        case 0: *s = sr; *c = cr; break;
        case 1: *s = cr; *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}
VSM_FN double vsm_sin(double x) { double s, c; vsm_sincos(x, &s, &c); return s; }
VSM_FN double vsm_cos(double x) { double s, c; vsm_sincos(x, &s, &c); return c; }

/* ---- float entry points: evaluate in double, round once ---- */
VSM_FN float vsm_expf(float x) { return (float)vsm_exp((double)x); }
VSM_FN float vsm_logf(float x) { return (float)vsm_log((double)x); }
VSM_FN float vsm_powf(float x, float y) { return (float)vsm_pow((double)x, (double)y); }
VSM_FN float vsm_atan2f(float y, float x) { return (float)vsm_atan2((double)y, (double)x); }
VSM_FN float vsm_sinf(float x) { return (float)vsm_sin((double)x); }
VSM_FN float vsm_cosf(float x) { return (float)vsm_cos((double)x); }
VSM_FN float vsm_cbrtf(float x) {
    VSM_NO_CONTRACT
    if (x != x || x == 0.0f) return x;
    const double a = vsm_abs((double)x);
    if (a == vsm_inf()) return x;
    const double r = vsm_exp(vsm_log(a) / 3.0);
    return (float)(x < 0.0f ? -r : r);
}

#endif /* VK_STRICT_MATH_H */
