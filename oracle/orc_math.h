/* oracle/orc_math.h -- ORACLE (test infrastructure only, see orc.h).
 * The transcendental calls of the restatement go through these wrappers: by default glibc (what the reference's own
 * CPU-executed code uses in oracle/_ref); after orc_set_strict_math(1) the software functions of
 * voldor_amd/csrc/vk_strict_math.h, the SAME header the HIP kernels compile in strict mode, so that oracle and product share
 * one rounding sequence and whole windows can be compared bit for bit (tests/test_gpu_strict.py).  sqrt, + - * / are IEEE
 * on both sides and stay as they are. */
#ifndef ORC_MATH_H
#define ORC_MATH_H
#include <math.h>
#include "../voldor_amd/csrc/vk_strict_math.h"
extern int orc_strict_math_flag;
static inline float om_expf(float x) { return orc_strict_math_flag ? vsm_expf(x) : expf(x); }
static inline float om_logf(float x) { return orc_strict_math_flag ? vsm_logf(x) : logf(x); }
static inline float om_powf(float x, float y) { return orc_strict_math_flag ? vsm_powf(x, y) : powf(x, y); }
static inline float om_atan2f(float y, float x) { return orc_strict_math_flag ? vsm_atan2f(y, x) : atan2f(y, x); }
static inline float om_sinf(float x) { return orc_strict_math_flag ? vsm_sinf(x) : sinf(x); }
static inline float om_cosf(float x) { return orc_strict_math_flag ? vsm_cosf(x) : cosf(x); }
static inline float om_cbrtf(float x) { return orc_strict_math_flag ? vsm_cbrtf(x) : cbrtf(x); }
static inline double om_sin(double x) { return orc_strict_math_flag ? vsm_sin(x) : sin(x); }
static inline double om_cos(double x) { return orc_strict_math_flag ? vsm_cos(x) : cos(x); }
#endif
