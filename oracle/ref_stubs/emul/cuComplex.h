/* stands in for the CUDA toolkit's cuComplex.h (single-precision part) with the algorithms the toolkit header publishes:
 * cuCabsf scales by the larger component, cuCdivf scales by |re|+|im| of the divisor.  TEST INFRASTRUCTURE ONLY. */
#pragma once
#include "../cuda_stub_common.h"
typedef float2 cuFloatComplex;
static inline cuFloatComplex make_cuFloatComplex(float r, float i) { cuFloatComplex c; c.x = r; c.y = i; return c; }
static inline float cuCrealf(cuFloatComplex x) { return x.x; }
static inline float cuCimagf(cuFloatComplex x) { return x.y; }
static inline cuFloatComplex cuCaddf(cuFloatComplex x, cuFloatComplex y) { return make_cuFloatComplex(x.x + y.x, x.y + y.y); }
static inline cuFloatComplex cuCsubf(cuFloatComplex x, cuFloatComplex y) { return make_cuFloatComplex(x.x - y.x, x.y - y.y); }
static inline cuFloatComplex cuCmulf(cuFloatComplex x, cuFloatComplex y) {
    return make_cuFloatComplex((x.x * y.x) - (x.y * y.y), (x.x * y.y) + (x.y * y.x));
}
static inline cuFloatComplex cuCdivf(cuFloatComplex x, cuFloatComplex y) {
    float s = fabsf(y.x) + fabsf(y.y), oos = 1.0f / s;
    const float ars = x.x * oos, ais = x.y * oos, brs = y.x * oos, bis = y.y * oos;
    s = (brs * brs) + (bis * bis); oos = 1.0f / s;
    return make_cuFloatComplex(((ars * brs) + (ais * bis)) * oos, ((ais * brs) - (ars * bis)) * oos);
}
static inline float cuCabsf(cuFloatComplex x) {
    float a = fabsf(x.x), b = fabsf(x.y), v, w, t;
    if (a > b) { v = a; w = b; } else { v = b; w = a; }
    t = w / v; t = 1.0f + t * t; t = v * sqrtf(t);
    if ((v == 0.0f) || (v > 3.402823466e38f) || (w > 3.402823466e38f)) t = v + w;
    return t;
}
